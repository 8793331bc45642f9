"""ctypes loader for libsdfhip.so (the hand-written HIP engine).

There is NO CPU fallback: if the shared library is missing, or no HIP device is usable, every entry point fails
loudly.  ``import torch`` happens first (when torch is installed) so that the library binds to the same HIP
runtime (libamdhip64.so.7) torch already loaded — device pointers and streams can then be shared.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDFLIB_USE_ENOKI=1 (the name of the reference's CMake option, read once at import): the library whose interpolateValue follows the
# order of the reference's Enoki flavour (libsdfhip_enoki.so); everything else, the ABI included, is the same
USE_ENOKI = os.environ.get("SDFLIB_USE_ENOKI", "0") not in ("", "0", "OFF", "off")
LIB_PATH = os.path.join(_HERE, "libsdfhip_enoki.so" if USE_ENOKI else "libsdfhip.so")
_LIB = None


class SdfHipError(RuntimeError):
    pass


class OctreeInfo(C.Structure):
    _fields_ = [("box_min", C.c_float * 3), ("box_max", C.c_float * 3), ("start_grid_size", C.c_int32), ("max_depth", C.c_uint32),
                ("value_range", C.c_float), ("min_border_value", C.c_float), ("num_words", C.c_uint64), ("num_leaves", C.c_uint64),
                ("num_nodes", C.c_uint64), ("num_samples", C.c_uint64), ("cell_begin", C.c_uint32), ("cell_end", C.c_uint32),
                ("body_words", C.c_uint64), ("body_offset", C.c_uint64), ("seconds_samples", C.c_double), ("seconds_decide", C.c_double),
                ("seconds_total", C.c_double), ("leaves_per_depth", C.c_uint64 * 16), ("fit_rechecks", C.c_uint64), ("num_traversals", C.c_uint64),
                ("post_pass_scheduled", C.c_uint64), ("start_grid_cell_size", C.c_float), ("reserved0", C.c_float), ("num_nearest_fallbacks", C.c_uint64),
                ("near_expansions", C.c_uint64), ("near_triangle_tests", C.c_uint64), ("seconds_near_candidates", C.c_double), ("seconds_near_search", C.c_double)]


class OctreeParams(C.Structure):
    _fields_ = [("box_min", C.c_float * 3), ("box_max", C.c_float * 3), ("depth", C.c_uint32), ("start_depth", C.c_uint32),
                ("rule", C.c_int32), ("rule_params", C.c_float * 2), ("algorithm", C.c_int32), ("layout", C.c_int32),
                ("fit_mode", C.c_int32), ("cell_begin", C.c_uint32), ("cell_end", C.c_uint32)]


class ExactInfo(C.Structure):
    _fields_ = [("box_min", C.c_float * 3), ("box_max", C.c_float * 3), ("start_grid_size", C.c_int32), ("start_depth", C.c_uint32),
                ("max_depth", C.c_uint32), ("bit_encoding_start_depth", C.c_uint32), ("bits_per_index", C.c_uint32),
                ("min_triangles_in_leafs", C.c_uint32), ("max_triangles_in_leafs", C.c_uint32),
                ("max_triangles_encoded_in_leafs", C.c_uint32), ("num_nodes", C.c_uint64), ("num_set_words", C.c_uint64),
                ("num_mask_bytes", C.c_uint64), ("num_triangles", C.c_uint64), ("cull_tests", C.c_uint64), ("seconds_total", C.c_double),
                ("start_grid_cell_size", C.c_float), ("reserved0", C.c_float)]


class MultiStats(C.Structure):
    _fields_ = [("ranks", C.c_int32), ("uses_rccl", C.c_int32), ("bytes_exchanged", C.c_uint64), ("seconds_bvh", C.c_double), ("seconds_shards", C.c_double),
                ("seconds_exchange", C.c_double)]


ACQUIRE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64)


class Exchange(C.Structure):
    _fields_ = [("user", C.c_void_p), ("acquire", ACQUIRE_FN), ("all_reduce_sum", ALL_REDUCE_FN), ("rank", C.c_int32), ("world", C.c_int32)]


# every symbol include/sdfhip.h declares: name -> (restype, argtypes)
_vp, _u32, _u64, _i32, _f32, _int = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_float, C.c_int
SIGNATURES = {
    "sdfhip_last_error": (C.c_char_p, []),
    "sdfhip_interpolation_flavour": (_int, []),
    "sdfhip_version": (C.c_char_p, []),
    "sdfhip_ctx_create": (_int, [_int, _vp, _int, C.POINTER(_vp)]),
    "sdfhip_ctx_destroy": (_int, [_vp]),
    "sdfhip_ctx_synchronize": (_int, [_vp]),
    "sdfhip_ctx_trim": (_int, [_vp, C.c_uint64]), "sdfhip_ctx_cached_bytes": (_int, [_vp, _vp]),
    "sdfhip_octree_compact": (_int, [_vp]), "sdfhip_octree_device_bytes": (_int, [_vp, _vp]),
    "sdfhip_ctx_stream": (_vp, [_vp]),
    "sdfhip_ctx_set_exchange": (_int, [_vp, C.POINTER(Exchange)]),
    "sdfhip_mesh_create": (_int, [_vp, _vp, _u32, _vp, _u32, C.POINTER(_vp)]),
    "sdfhip_mesh_create_ex": (_int, [_vp, _vp, _u32, _vp, _u32, _vp, C.POINTER(_vp)]),
    "sdfhip_mesh_create_opt": (_int, [_vp, _vp, _u32, _vp, _u32, _vp, _u32, C.POINTER(_vp)]),
    "sdfhip_mesh_edge_stats": (_int, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "sdfhip_mesh_destroy": (_int, [_vp]),
    "sdfhip_mesh_triangle_data": (_int, [_vp, _vp]),
    "sdfhip_mesh_build_bvh": (_int, [_vp, C.POINTER(C.c_double)]),
    "sdfhip_mesh_bvh_export": (_int, [_vp, _vp, _vp, _int]),
    "sdfhip_mesh_bvh_import": (_int, [_vp, _vp, _vp, _int]),
    "sdfhip_mesh_nearest": (_int, [_vp, _vp, _u64, _vp, _int]),
    "sdfhip_abi_sizes": (None, [_vp]),
    "sdfhip_test_acosf_mismatches": (_u64, [_u32, _u32, _u64, _int]),
    "sdfhip_test_acosf_device": (_int, [_vp, _u32, _u32, _u32, _vp]),
    "sdfhip_test_set_host_acos": (None, [_int]),
    "sdfhip_test_sort_matches_std": (_int, [_vp, _u64, _int]),
    "sdfhip_test_heap_sort_matches_std": (_int, [_vp, _u64]),
    "sdfhip_test_plan_bvh": (_int, [_vp, _u32, _vp, _u32, _vp, _vp, _vp]),
    "sdfhip_mesh_nearest_stats": (_int, [_vp, _vp, _u64, _vp]),
    "sdfhip_mesh_nearest_stats_preseeded": (_int, [_vp, _vp, _u64, _vp]),
    "sdfhip_mesh_point_values": (_int, [_vp, _vp, _vp, _u64, _vp, _int]),
    "sdfhip_octree_build": (_int, [_vp, _vp, C.POINTER(OctreeParams), C.POINTER(_vp)]),
    "sdfhip_octree_build_shard": (_int, [_vp, _vp, C.POINTER(OctreeParams), C.POINTER(_vp)]),
    "sdfhip_octree_emit_shard": (_int, [_vp, _u64, _vp, _vp, _int]),
    "sdfhip_octree_from_data": (_int, [_vp, _vp, _u64, _int, _vp, _vp, _i32, _u32, _f32, _f32, C.POINTER(_vp)]),
    "sdfhip_octree_set_start_grid_cell_size": (_int, [_vp, _f32]),
    "sdfhip_octree_destroy": (_int, [_vp]),
    "sdfhip_octree_get_info": (_int, [_vp, C.POINTER(OctreeInfo)]),
    "sdfhip_octree_download": (_int, [_vp, _vp, _int]),
    "sdfhip_octree_device_words": (_vp, [_vp]),
    "sdfhip_octree_query": (_int, [_vp, _vp, _u64, _vp, _vp, _int, _int]),
    "sdfhip_octree_query_grid": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _int, _int]),
    "sdfhip_exact_build": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, C.POINTER(_vp)]),
    "sdfhip_exact_from_data": (_int, [_vp, C.POINTER(ExactInfo), _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "sdfhip_exact_set_start_grid_cell_size": (_int, [_vp, _f32]),
    "sdfhip_exact_destroy": (_int, [_vp]),
    "sdfhip_exact_get_info": (_int, [_vp, C.POINTER(ExactInfo)]),
    "sdfhip_exact_download": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "sdfhip_exact_build_shard": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, C.POINTER(_vp)]),
    "sdfhip_exact_shard_cells": (_int, [_vp, _vp]),
    "sdfhip_exact_emit_shard": (_int, [_vp, _u64, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _int]),
    "sdfhip_exact_from_parts": (_int, [_vp, _vp, C.POINTER(ExactInfo), _vp, _vp, _vp, _vp, _int, C.POINTER(_vp)]),
    "sdfhip_exact_triangle_data": (_int, [_vp, _vp]),
    "sdfhip_exact_query": (_int, [_vp, _vp, _u64, _vp, _vp, _vp, _int]),
    "sdfhip_tricubic_fit": (_int, [_vp, _vp, _vp, _u64, _vp, _int]),
    "sdfhip_is_near_minimize": (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _vp]),
    "sdfhip_test_gather_blocks": (_int, [_vp, _vp, _vp, _u64, _vp]),
    "sdfhip_test_valu_peak": (_int, [_vp, _u32, _u32, _vp]),
    "sdfhip_multi_create": (_int, [_vp, _int, C.POINTER(_vp)]),
    "sdfhip_multi_destroy": (_int, [_vp]),
    "sdfhip_multi_size": (_int, [_vp]),
    "sdfhip_multi_ctx": (_vp, [_vp, _int]),
    "sdfhip_multi_transport": (C.c_char_p, [_vp]),
    "sdfhip_multi_get_stats": (_int, [_vp, C.POINTER(MultiStats)]),
    "sdfhip_multi_octree_build": (_int, [_vp, _vp, _u32, _vp, _u32, _vp, C.POINTER(OctreeParams), _vp, _vp]),
    "sdfhip_multi_exact_build": (_int, [_vp, _vp, _u32, _vp, _u32, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp]),
}


def lib():
    """Load libsdfhip.so; raises SdfHipError if it has not been built (no fallback of any kind)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SdfHipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  sdflib_amd has no CPU fallback.")
        try:
            import torch  # noqa: F401  (share torch's HIP runtime when present)
        except Exception:
            pass
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        msg = lib().sdfhip_last_error()
        raise SdfHipError(f"sdfhip error {rc}: {msg.decode() if msg else ''}")
