"""ctypes binding of oracle/_ref/libtmd_ref.so — TEST INFRASTRUCTURE ONLY.

The library is the REAL reference's nearest-triangle search (tmd::TriangleMeshDistance, compiled by oracle/Makefile
from /root/reference/libs/InteractiveComputerGraphics/InteractiveComputerGraphics/TriangleMeshDistance.h with no
stand-ins, see oracle/ref_tmd.cpp).  It exists where /root/reference exists (or where a prebuilt copy travelled
to); `available()` says whether it can be used.  Used by tests/test_oracle_ref_pin.py and tests/golden/make_golden.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libtmd_ref.so")
_HDR = "/root/reference/libs/InteractiveComputerGraphics/InteractiveComputerGraphics/TriangleMeshDistance.h"
_LIB = None


def available():
    if os.path.exists(_SO):
        return True
    if os.path.exists(_HDR):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_ref/libtmd_ref.so"])
        return os.path.exists(_SO)
    return False


def lib():
    global _LIB
    if _LIB is None:
        if not available():
            raise RuntimeError("oracle/_ref/libtmd_ref.so is absent and /root/reference is not here to build it")
        L = C.CDLL(_SO)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        for name, (res, args) in {
            "tmdref_create": (vp, [vp, u32, vp, u32]), "tmdref_destroy": (None, [vp]),
            "tmdref_nearest": (None, [vp, vp, u64, vp, vp]), "tmdref_num_nodes": (u64, [vp]),
            "tmdref_export_nodes": (None, [vp, vp, vp, vp]),
        }.items():
            fn = getattr(L, name); fn.restype = res; fn.argtypes = args
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class RefMesh:
    """tmd::TriangleMeshDistance as SdfLib's ICG wrapper constructs and queries it (TrianglesInfluence.h:884-905)."""

    def __init__(self, vertices, triangles):
        self.v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        self.f = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        self.h = lib().tmdref_create(_p(self.v), len(self.v), _p(self.f), len(self.f))

    def __del__(self):
        if getattr(self, "h", None):
            lib().tmdref_destroy(self.h); self.h = None

    def nearest(self, pts, with_dist=False):
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
        ids = np.empty(len(pts), dtype=np.uint32)
        d = np.empty(len(pts), dtype=np.float64) if with_dist else None
        lib().tmdref_nearest(self.h, _p(pts), len(pts), _p(ids), _p(d))
        return (ids, d) if with_dist else ids

    def bvh_export(self):
        """(spheres[n, 8], left_right[n, 2], root_sphere[4]) in the layout of pyoracle.Mesh.bvh_export."""
        n = lib().tmdref_num_nodes(self.h)
        sph = np.empty((n, 8), dtype=np.float64); lr = np.empty((n, 2), dtype=np.int32); root = np.empty(4, dtype=np.float64)
        lib().tmdref_export_nodes(self.h, _p(sph), _p(lr), _p(root))
        return sph, lr, root
