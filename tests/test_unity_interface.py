"""libSdfLibUnity.so: the reference's Unity plugin interface (src/tools/SdfLibUnity/SdfExportFunc.h:16-58) bound through ctypes the
way the plugin's C# side binds it (plain pointers, floats by value, vec3 returned by value)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, bits

UNITY = os.path.join(ROOT, "sdflib_amd", "libSdfLibUnity.so")
# name -> (restype, argtypes): the reference's declarations, SdfExportFunc.h:16-58
_vp, _u32, _f = C.c_void_p, C.c_uint32, C.c_float


class Vec3(C.Structure):
    _fields_ = [("x", _f), ("y", _f), ("z", _f)]


REFERENCE_INTERFACE = {
    "saveSdf": (None, [_vp, C.c_char_p]),
    "loadSdf": (_vp, [C.c_char_p]),
    "createExactOctreeSdf": (_vp, [_vp, _u32, _vp, _u32, _f, _f, _f, _f, _f, _f, _u32, _u32, _u32, _u32]),
    "createOctreeSdf": (_vp, [_vp, _u32, _vp, _u32, _f, _f, _f, _f, _f, _f, _u32, _u32, _f, _u32]),
    "getDistance": (_f, [_vp, _f, _f, _f]),
    "getDistanceAndGradient": (_f, [_vp, _f, _f, _f, C.POINTER(Vec3)]),
    "getBBMinPoint": (Vec3, [_vp]),
    "getBBSize": (Vec3, [_vp]),
    "getStartGridSize": (_u32, [_vp]),
    "getOctreeDataSize": (_u32, [_vp]),
    "getOctreeData": (None, [_vp, _vp]),
    "deleteSdf": (None, [_vp]),
}


def _load():
    import sdflib_amd
    if not os.path.exists(UNITY):
        import __graft_entry__ as g
        g.build()
    sdflib_amd.lib()
    L = C.CDLL(UNITY)
    for name, (res, args) in REFERENCE_INTERFACE.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    return L


def test_every_function_of_the_reference_interface_is_exported():
    L = _load()
    for name in REFERENCE_INTERFACE:
        assert hasattr(L, name), f"{name} missing from libSdfLibUnity.so"


@pytest.mark.gpu
def test_unity_interface_matches_oracle(tmp_path, oracle):
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
    L = _load()
    v, f = bumpy_icosphere(3)
    box = box_with_margin(v)
    v = np.ascontiguousarray(v, dtype=np.float32); f = np.ascontiguousarray(f, dtype=np.uint32)
    bb = [float(x) for x in box]
    om = oracle.Mesh(v, f)
    pts = random_points_in_box(box, 300, seed=9)
    pts[:10] *= 3.0          # outside the box too

    # createOctreeSdf always uses the CONTINUITY builder (SdfExportFunc.cpp:84-113)
    h = L.createOctreeSdf(v.ctypes.data, len(v), f.ctypes.data, f.size, *bb, 2, 5, 1e-3, 1)
    assert h
    oc = oracle.Octree(om, box, 5, 2, 1e-3, continuity=True)
    n = L.getOctreeDataSize(h)
    assert n == len(oc.data())
    words = np.empty(n, dtype=np.uint32)
    L.getOctreeData(h, words.ctypes.data)
    assert np.array_equal(words, oc.data())
    assert L.getStartGridSize(h) == 4
    mn, sz = L.getBBMinPoint(h), L.getBBSize(h)
    assert sz.x > 0 and sz.x == sz.y == sz.z          # the octree box is the input box made cubic (OctreeSdf.cpp:43-46)
    assert np.array_equal(np.array([mn.x, mn.y, mn.z], dtype=np.float32), oc.box[:3])
    assert np.array_equal(np.array([sz.x, sz.y, sz.z], dtype=np.float32), oc.box[3:] - oc.box[:3])
    d0, g0 = oc.query(pts, grad=True)
    d = np.array([L.getDistance(h, *map(float, p)) for p in pts], dtype=np.float32)
    assert np.array_equal(bits(d), bits(d0))
    g = Vec3()
    for i in range(0, len(pts), 7):
        di = L.getDistanceAndGradient(h, *map(float, pts[i]), C.byref(g))
        assert np.float32(di).view(np.uint32) == d0[i].view(np.uint32)
        assert np.array_equal(bits(np.array([g.x, g.y, g.z], dtype=np.float32)), bits(g0[i]))
    # saveSdf / loadSdf round trip through the reference's file layout
    path = str(tmp_path / "oct.bin").encode()
    L.saveSdf(h, path)
    h2 = L.loadSdf(path)
    assert h2 and L.getOctreeDataSize(h2) == n
    w2 = np.empty(n, dtype=np.uint32); L.getOctreeData(h2, w2.ctypes.data)
    assert np.array_equal(w2, words)
    assert np.float32(L.getDistance(h2, *map(float, pts[20]))).view(np.uint32) == d0[20].view(np.uint32)
    L.deleteSdf(h2); L.deleteSdf(h)

    # createExactOctreeSdf
    e = L.createExactOctreeSdf(v.ctypes.data, len(v), f.ctypes.data, f.size, *bb, 1, 5, 16, 1)
    assert e
    ex = oracle.Exact(om, box, 5, 1, 16)
    de = np.array([L.getDistance(e, *map(float, p)) for p in pts], dtype=np.float32)
    assert np.array_equal(bits(de), bits(ex.query(pts)))
    assert L.getStartGridSize(e) == 0 and L.getOctreeDataSize(e) == 0      # OctreeSdf only in the reference (dynamic_cast, :140-154)
    L.deleteSdf(e)
    assert L.loadSdf(str(tmp_path / "missing.bin").encode()) is None       # loadFromFile -> nullptr


def test_entry_points_never_throw_through_the_c_abi(tmp_path):
    """a failing constructor / loader (here: depth beyond the lattice limit, start depth > depth, a corrupt file; on a
    box without a GPU also 'no HIP device') must come back as NULL, not as a C++ exception unwinding into the host process."""
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
    L = _load()
    v, f = bumpy_icosphere(1)
    bb = [float(x) for x in box_with_margin(v)]
    v = np.ascontiguousarray(v, dtype=np.float32); f = np.ascontiguousarray(f, dtype=np.uint32)
    assert L.createOctreeSdf(v.ctypes.data, len(v), f.ctypes.data, f.size, *bb, 2, 11, 1e-3, 1) is None
    assert L.createOctreeSdf(v.ctypes.data, len(v), f.ctypes.data, f.size, *bb, 6, 3, 1e-3, 1) is None
    assert L.createExactOctreeSdf(v.ctypes.data, len(v), f.ctypes.data, f.size, *bb, 9, 3, 16, 1) is None
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\x01" + b"\x00" * 7)
    assert L.loadSdf(str(bad).encode()) is None
