"""ctypes binding of the ORACLE (oracle/libsdf_oracle.so) — TEST INFRASTRUCTURE ONLY.

May be imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RULE_NONE, RULE_TRAPEZOIDAL, RULE_SIMPSONS, RULE_BY_DISTANCE = 0, 1, 2, 3
LAYOUT_GLOBAL_DFS, LAYOUT_SUBTREES = 0, 1


def build(force=False):
    so = os.path.join(_HERE, "libsdf_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".h", ".cpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        # SDFLIB_USE_ENOKI=1 (the name of the reference's CMake option): the oracle with interpolateValue in the Enoki flavour's order
        enoki = os.environ.get("SDFLIB_USE_ENOKI", "0") not in ("", "0", "OFF", "off")
        so = os.path.join(_HERE, "libsdf_oracle_enoki.so" if enoki else "libsdf_oracle.so")
        if not os.path.exists(so):
            build(force=True)
        L = C.CDLL(so)
        vp, u32, u64, f32, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int32
        sig = {
            "orc_mesh_create": (vp, [vp, u32, vp, u32]), "orc_mesh_create_ex": (vp, [vp, u32, vp, u32, vp]), "orc_mesh_destroy": (None, [vp]),
            "orc_mesh_triangle_data": (None, [vp, vp]), "orc_mesh_build_bvh": (C.c_double, [vp]),
            "orc_bvh_num_nodes": (u64, [vp]), "orc_bvh_export": (None, [vp, vp, vp]),
            "orc_bvh_nearest": (None, [vp, vp, u64, vp, vp]),
            "orc_sqdist_point_triangle": (f32, [vp, u32, vp]), "orc_sqdist_point_triangle_raw": (f32, [vp, vp, vp, vp]),
            "orc_signed_dist_point_triangle": (f32, [vp, u32, vp]),
            "orc_signed_dist_point_triangle_grad": (f32, [vp, u32, vp, vp]),
            "orc_signed_dist_point_triangle_grad_local": (f32, [vp, u32, vp, vp]),
            "orc_point_values": (None, [vp, vp, vp, u64, vp]),
            "orc_fit_matrix": (None, [vp]), "orc_tricubic_fit": (None, [vp, f32, vp]),
            "orc_tricubic_value": (f32, [vp, vp]), "orc_tricubic_gradient": (None, [vp, vp, vp]),
            "orc_tricubic_value_literal": (f32, [vp, vp]), "orc_tricubic_value_enoki": (f32, [vp, vp]), "orc_interpolation_flavour": (C.c_int, []),
            "orc_tricubic_vertex_values": (None, [vp, vp, f32, vp]), "orc_rule_value": (f32, [C.c_int, vp, vp, f32]),
            "orc_stencil": (None, [vp, vp, vp]), "orc_is_near_minimize": (C.c_int, [f32, vp, vp, f32, vp]),
            "orc_octree_build": (vp, [vp, vp, u32, u32, C.c_int, f32, f32, C.c_int, C.c_int]),
            "orc_octree_build_continuity": (vp, [vp, vp, u32, u32, C.c_int, f32, f32]),
            "orc_octree_destroy": (None, [vp]), "orc_octree_size": (u64, [vp]), "orc_octree_data": (None, [vp, vp]),
            "orc_octree_info": (None, [vp, vp, vp, vp, vp, vp, vp]),
            "orc_octree_query": (None, [vp, vp, u64, vp, vp, C.c_int]),
            "orc_octree_query_raw": (None, [vp, u64, vp, i32, f32, vp, u64, vp, vp, C.c_int]),
            "orc_exact_build": (vp, [vp, vp, u32, u32, u32, C.c_int]), "orc_exact_destroy": (None, [vp]),
            "orc_exact_build_mt": (vp, [vp, vp, u32, u32, u32, C.c_int, C.c_int]),
            "orc_exact_sizes": (None, [vp, vp, vp, vp, vp, vp, vp, vp]), "orc_exact_data": (None, [vp, vp, vp, vp, vp]),
            "orc_exact_query": (None, [vp, vp, u64, vp, vp, vp, C.c_int]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name); fn.restype = res; fn.argtypes = args
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Mesh:
    def __init__(self, vertices, triangles, bbox=None):
        self.v = _f32(vertices).reshape(-1, 3)
        self.f = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        bb = _f32(bbox) if bbox is not None else None
        self.h = lib().orc_mesh_create_ex(_p(self.v), len(self.v), _p(self.f), len(self.f), _p(bb))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mesh_destroy(self.h); self.h = None

    def triangle_data(self):
        out = np.empty((len(self.f), 37), dtype=np.float32)
        lib().orc_mesh_triangle_data(self.h, _p(out)); return out

    def build_bvh(self):
        return lib().orc_mesh_build_bvh(self.h)

    def bvh_export(self):
        n = lib().orc_bvh_num_nodes(self.h)
        sph = np.empty((n, 8), dtype=np.float64); lr = np.empty((n, 2), dtype=np.int32)
        lib().orc_bvh_export(self.h, _p(sph), _p(lr)); return sph, lr

    def nearest(self, pts, with_dist=False):
        pts = _f32(pts).reshape(-1, 3)
        ids = np.empty(len(pts), dtype=np.uint32)
        d = np.empty(len(pts), dtype=np.float64) if with_dist else None
        lib().orc_bvh_nearest(self.h, _p(pts), len(pts), _p(ids), _p(d))
        return (ids, d) if with_dist else ids

    def sqdist(self, tri, p):
        return lib().orc_sqdist_point_triangle(self.h, int(tri), _p(_f32(p)))

    def signed(self, tri, p):
        return lib().orc_signed_dist_point_triangle(self.h, int(tri), _p(_f32(p)))

    def signed_grad(self, tri, p, local=False):
        g = np.zeros(3, dtype=np.float32)
        fn = lib().orc_signed_dist_point_triangle_grad_local if local else lib().orc_signed_dist_point_triangle_grad
        d = fn(self.h, int(tri), _p(_f32(p)), _p(g)); return d, g

    def point_values(self, pts, tris):
        pts = _f32(pts).reshape(-1, 3); tris = np.ascontiguousarray(tris, dtype=np.uint32)
        out = np.empty((len(pts), 8), dtype=np.float32)
        lib().orc_point_values(self.h, _p(pts), _p(tris), len(pts), _p(out)); return out


def triangle_distance_test(seed=2222, n=1000000):
    """The reference's TriangleDistanceTest loop (glibc srand/rand sample sequence): (violations, max |raw - data|, max |signed^2 - data|, sums)."""
    f = lib().orc_triangle_distance_test
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint, C.c_uint32] + [C.POINTER(C.c_float)] * 4
    a, b, c, d = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    bad = f(seed, n, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
    return int(bad), a.value, b.value, c.value, d.value


def sqdist_raw(p, a, b, c):
    return lib().orc_sqdist_point_triangle_raw(_p(_f32(p)), _p(_f32(a)), _p(_f32(b)), _p(_f32(c)))


def fit_matrix():
    m = np.empty((64, 64), dtype=np.int32); lib().orc_fit_matrix(_p(m)); return m


def tricubic_fit(values_8x8, node_size):
    out = np.empty(64, dtype=np.float32)
    lib().orc_tricubic_fit(_p(_f32(values_8x8)), np.float32(node_size), _p(out)); return out


def tricubic_value(c, frac):
    return lib().orc_tricubic_value(_p(_f32(c)), _p(_f32(frac)))


def tricubic_gradient(c, frac):
    g = np.empty(3, dtype=np.float32); lib().orc_tricubic_gradient(_p(_f32(c)), _p(_f32(frac)), _p(g)); return g


def tricubic_vertex_values(c, frac, node_size):
    o = np.empty(8, dtype=np.float32)
    lib().orc_tricubic_vertex_values(_p(_f32(c)), _p(_f32(frac)), np.float32(node_size), _p(o)); return o


def rule_value(rule, c, mid_19x8, param1=0.0):
    return lib().orc_rule_value(int(rule), _p(_f32(c)), _p(_f32(mid_19x8)), np.float32(param1))


def stencil():
    cs = np.empty((8, 8), dtype=np.int32); rel = np.empty((19, 3), dtype=np.float32); w = np.empty(19, dtype=np.float32)
    lib().orc_stencil(_p(cs), _p(rel), _p(w)); return cs, rel, w


def is_near_minimize(half, radius8, tri3x3, thr):
    it = C.c_uint32(0)
    r = lib().orc_is_near_minimize(np.float32(half), _p(_f32(radius8)), _p(_f32(tri3x3)), np.float32(thr), C.byref(it))
    return bool(r), it.value


class Octree:
    """Oracle OctreeSdf (NO_CONTINUITY)."""

    def __init__(self, mesh, box6, depth, start_depth, threshold=1e-3, rule=RULE_TRAPEZOIDAL, param1=0.0,
                 vertex_cache=False, layout=LAYOUT_SUBTREES, continuity=False):
        self.mesh = mesh
        if continuity:
            self.h = lib().orc_octree_build_continuity(mesh.h, _p(_f32(box6)), depth, start_depth, rule, np.float32(threshold), np.float32(param1))
        else:
            self.h = lib().orc_octree_build(mesh.h, _p(_f32(box6)), depth, start_depth, rule, np.float32(threshold),
                                            np.float32(param1), int(vertex_cache), int(layout))
        box = np.empty(6, dtype=np.float32); g = C.c_int32(); cell = C.c_float(); vr = C.c_float(); mb = C.c_float(); nq = C.c_uint64()
        lib().orc_octree_info(self.h, _p(box), C.byref(g), C.byref(cell), C.byref(vr), C.byref(mb), C.byref(nq))
        self.box, self.start_grid_size, self.cell_size = box, g.value, cell.value
        self.value_range, self.min_border, self.num_bvh_queries = vr.value, mb.value, nq.value
        self.depth, self.start_depth = depth, start_depth

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_octree_destroy(self.h); self.h = None

    def data(self):
        out = np.empty(lib().orc_octree_size(self.h), dtype=np.uint32)
        lib().orc_octree_data(self.h, _p(out)); return out

    def query(self, pts, grad=False, threads=0):
        pts = _f32(pts).reshape(-1, 3)
        d = np.empty(len(pts), dtype=np.float32); g = np.zeros((len(pts), 3), dtype=np.float32) if grad else None
        lib().orc_octree_query(self.h, _p(pts), len(pts), _p(d), _p(g), threads)
        return (d, g) if grad else d


def octree_query_raw(data, box6, start_grid_size, min_border, pts, grad=False, threads=0):
    data = np.ascontiguousarray(data, dtype=np.uint32); pts = _f32(pts).reshape(-1, 3)
    d = np.empty(len(pts), dtype=np.float32); g = np.zeros((len(pts), 3), dtype=np.float32) if grad else None
    lib().orc_octree_query_raw(_p(data), len(data), _p(_f32(box6)), int(start_grid_size), np.float32(min_border),
                               _p(pts), len(pts), _p(d), _p(g), threads)
    return (d, g) if grad else d


class Exact:
    """Oracle ExactOctreeSdf (single-thread semantics).  threads != 1 (0 = all cores) builds the start cells concurrently in canonical
    mode (vertex_cache False) — the same arrays as the sequential build, only faster (full-size GPU parity tests)."""

    def __init__(self, mesh, box6, depth, start_depth=1, min_triangles=128, vertex_cache=False, threads=1):
        self.mesh = mesh
        self.h = lib().orc_exact_build_mt(mesh.h, _p(_f32(box6)), depth, start_depth, min_triangles, int(vertex_cache), int(threads))
        nn, ns, nm, cull = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        bits, ml, me = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().orc_exact_sizes(self.h, C.byref(nn), C.byref(ns), C.byref(nm), C.byref(bits), C.byref(ml), C.byref(me), C.byref(cull))
        self.num_nodes, self.num_set_words, self.num_mask_bytes = nn.value, ns.value, nm.value
        self.bits_per_index, self.max_tri_in_leafs, self.max_tri_encoded, self.cull_tests = bits.value, ml.value, me.value, cull.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_exact_destroy(self.h); self.h = None

    def data(self):
        nodes = np.empty((self.num_nodes, 2), dtype=np.uint32); has = np.empty(self.num_nodes, dtype=np.uint8)
        sets = np.empty(self.num_set_words, dtype=np.uint32); masks = np.empty(max(self.num_mask_bytes, 1), dtype=np.uint8)
        lib().orc_exact_data(self.h, _p(nodes), _p(has), _p(sets), _p(masks))
        return nodes, has, sets, masks[:self.num_mask_bytes]

    def query(self, pts, grad=False, tri=False, threads=0):
        pts = _f32(pts).reshape(-1, 3)
        d = np.empty(len(pts), dtype=np.float32)
        g = np.zeros((len(pts), 3), dtype=np.float32) if grad else None
        t = np.empty(len(pts), dtype=np.uint32) if tri else None
        lib().orc_exact_query(self.h, _p(pts), len(pts), _p(d), _p(g), _p(t), threads)
        res = [d]
        if grad: res.append(g)
        if tri: res.append(t)
        return res[0] if len(res) == 1 else tuple(res)
