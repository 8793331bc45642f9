"""CPU tests: the C-ABI library loads and exports every symbol include/sdfhip.h and include/sdfhip_test.h declare; it fails loudly
(no CPU fallback) when no HIP device is present.  No compute calls are made here."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "sdfhip.h")).read() + open(os.path.join(ROOT, "include", "sdfhip_test.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sdfhip_[a-z0-9_]+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def built_lib():
    import sdflib_amd
    if not os.path.exists(sdflib_amd.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "sdflib_amd", "csrc"), "-j8"])
    return sdflib_amd.lib()


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    from sdflib_amd._lib import SIGNATURES
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(built_lib, name), f"{name} declared in sdfhip.h but not exported by libsdfhip.so"
        assert name in SIGNATURES, f"{name} has no ctypes signature in sdflib_amd/_lib.py"
    assert sorted(SIGNATURES) == declared


def test_restated_acosf_equals_the_running_libm_on_every_float(built_lib):
    """The mesh preparation takes the corner angles of the vertex pseudonormals on the device with glibc's acosf algorithm restated
    (dev_math.h::acosfGlibc) instead of sending the cosines to the host's libm.  libm's acosf is not correctly rounded, so the claim is
    checked exhaustively: the host compilation of that function against ::acosf on all 2 130 706 434 floats of [-1, 1]."""
    threads = max(1, min(32, len(os.sched_getaffinity(0))))
    assert built_lib.sdfhip_test_acosf_mismatches(0, 1, 1 << 32, threads) == 0


def test_enoki_flavour_library_exports_the_same_abi(built_lib):
    """libsdfhip_enoki.so (interpolateValue in the order of the reference's SDFLIB_USE_ENOKI=ON flavour): same symbols, other flavour id.
    Loading a second copy of the engine is harmless without a device call; the flavour query needs none."""
    import ctypes as C
    import sdflib_amd
    path = os.path.join(os.path.dirname(sdflib_amd.LIB_PATH), "libsdfhip_enoki.so")
    assert os.path.exists(path), "libsdfhip_enoki.so has not been built (make -C sdflib_amd/csrc)"
    L = C.CDLL(path)
    for name in _declared_symbols():
        assert hasattr(L, name), f"{name} not exported by libsdfhip_enoki.so"
    L.sdfhip_interpolation_flavour.restype = C.c_int
    assert L.sdfhip_interpolation_flavour() == 1 and built_lib.sdfhip_interpolation_flavour() == 0
    assert b"gfx950" in open(path, "rb").read()


def test_code_object_targets_gfx950(built_lib):
    import sdflib_amd
    blob = open(sdflib_amd.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_no_device_fails_loudly_without_fallback(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import sdflib_amd as S
    with pytest.raises(S.SdfHipError) as e:
        S.Context(0)
    assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)


def test_product_never_imports_the_oracle():
    """The product path (sdflib_amd/, include/) must not reference oracle/ in any way."""
    bad = []
    for base in ("sdflib_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"import\s+oracle|from\s+oracle|pyoracle|sdf_oracle|orc_[a-z]+\.h|liboracle|libsdf_oracle", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_struct_sizes_match_the_header(built_lib):
    """ctypes mirrors of the info/params structs must have the C sizes (compiled probe)."""
    from sdflib_amd._lib import OctreeInfo, OctreeParams, ExactInfo
    src = '#include "sdfhip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu\\n", sizeof(sdfhip_octree_info), sizeof(sdfhip_octree_params), sizeof(sdfhip_exact_info));return 0;}\n'
    exe = "/tmp/_sdfhip_sizes"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    a, b, c = (int(x) for x in subprocess.check_output([exe]).split())
    assert (ctypes.sizeof(OctreeInfo), ctypes.sizeof(OctreeParams), ctypes.sizeof(ExactInfo)) == (a, b, c)


def test_planner_sort_equals_std_sort():
    """The BVH planner's multi-threaded restatement of libstdc++'s introsort must leave the SAME permutation as std::sort
    (ties decide which triangles fall on which side of a median split, hence the tree, hence nearest-triangle ties)."""
    import ctypes as C
    import numpy as np
    from sdflib_amd._lib import lib
    rng = np.random.default_rng(0)
    for n in (0, 1, 2, 16, 17, 100, 4097, 70001, 400000):
        inputs = [rng.random(n), rng.integers(0, max(1, n // 6 + 1), n).astype(np.float64), np.sort(rng.random(n)),
                  np.sort(rng.integers(0, 50, n).astype(np.float64))[::-1].copy(), np.repeat(rng.random(max(1, n // 3 + 1)), 3)[:n].copy(),
                  np.zeros(n), np.tile([1.0, 0.0], n // 2 + 1)[:n].copy(), np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]).astype(np.float64)]
        for keys in inputs:
            keys = np.ascontiguousarray(keys, dtype=np.float64)
            for threads in (1, 6):
                assert lib().sdfhip_test_sort_matches_std(keys.ctypes.data_as(C.c_void_p), n, threads) == 0


def test_restated_heap_sort_matches_libstdcxx():
    """std::sort falls back to heap sort when a range exhausts introsort's depth limit (it happens in the 1.31 M-triangle tree); the device
    subtree builder restates libstdc++'s __make_heap / __sort_heap move for move (bvh.hip, stdHeapSort: compiled for the host here)."""
    import ctypes as C
    import numpy as np
    from sdflib_amd._lib import lib
    rng = np.random.default_rng(1)
    for n in (0, 1, 2, 3, 4, 5, 16, 17, 31, 32, 33, 100, 1000, 4097, 16384):
        inputs = [rng.random(n), rng.integers(0, max(1, n // 6 + 1), n).astype(np.float64), np.sort(rng.random(n)), np.sort(rng.random(n))[::-1].copy(),
                  np.repeat(rng.random(max(1, n // 3 + 1)), 3)[:n].copy(), np.zeros(n), np.tile([1.0, 0.0], n // 2 + 1)[:n].copy()]
        for keys in inputs:
            keys = np.ascontiguousarray(keys, dtype=np.float64)
            assert lib().sdfhip_test_heap_sort_matches_std(keys.ctypes.data_as(C.c_void_p), n) == 0, n


def test_ctypes_struct_mirrors_have_the_c_sizes():
    import ctypes as C
    from sdflib_amd._lib import lib, OctreeInfo, OctreeParams, ExactInfo
    out = (C.c_uint64 * 3)()
    lib().sdfhip_abi_sizes(out)
    assert [int(x) for x in out] == [C.sizeof(OctreeInfo), C.sizeof(OctreeParams), C.sizeof(ExactInfo)]


def test_every_switch_of_the_library_is_listed_here_or_in_a_named_test():
    """The census the header of this file states: a switch added to the library without a test fails here."""
    import glob, re
    found = set()
    for path in glob.glob(os.path.join(ROOT, "sdflib_amd", "csrc", "*.h*")):
        found |= set(re.findall(r'getenv\("(SDFHIP_[A-Z_0-9]+)"\)', open(path).read()))
    from test_gpu_switches import _CASES
    here = {k for env, _ in _CASES.values() for k in env}
    elsewhere = {"SDFHIP_BVH_BUILD", "SDFHIP_BVH_DEVICE_SUBTREES", "SDFHIP_TIMING", "SDFHIP_BVH_SORT_THREADS", "SDFHIP_BVH_PAR_DEPTH", "SDFHIP_BVH_MIN_PARALLEL",
                 "SDFHIP_BVH_PAR_PARTITION", "SDFHIP_MULTI_CUTS", "SDFHIP_EXACT_LISTS_MB", "SDFHIP_QUERY_CHUNK"}
    assert found <= here | elsewhere, sorted(found - here - elsewhere)
    assert len(found) <= 22, sorted(found)
    tests_text = "".join(open(p).read() for p in glob.glob(os.path.join(ROOT, "tests", "*.py")))
    for k in elsewhere & found:
        assert k in tests_text, k
