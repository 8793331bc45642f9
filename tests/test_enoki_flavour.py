"""The reference's SDFLIB_USE_ENOKI=ON flavour of interpolateValue (include/SdfLib/InterpolationMethods.h:383-430; its CMake default):
libsdfhip_enoki.so against libsdf_oracle_enoki.so.  Enoki's headers are not in the image, so the order of enoki::dot is restated from its
semantics ((a0 b0 + a1 b1) + (a2 b2 + a3 b3): DPPS, and Enoki's generic hsum) and this flavour is "parity unpinned" like the glm side."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _enoki_value_numpy(c, f):
    """independent restatement in numpy float32 (one rounding per operation), straight from the reference's statement order"""
    f32 = np.float32
    c = np.asarray(c, dtype=np.float32); fx, fy, fz = (f32(v) for v in f)
    x1 = np.array([f32(1), fx, fx * fx, (fx * fx) * fx], dtype=np.float32)
    x2 = fy * x1; x3 = fy * x2; x4 = fy * x3

    def dot(a, b):
        p = (a * b).astype(np.float32)
        return f32(f32(p[0] + p[1]) + f32(p[2] + p[3]))

    total = None
    for k in range(4):
        if k:
            x1 = fz * x1; x2 = fz * x2; x3 = fz * x3; x4 = fz * x4
        v = c[16 * k:16 * k + 16]
        slab = f32(f32(f32(dot(x1, v[0:4]) + dot(x2, v[4:8])) + dot(x3, v[8:12])) + dot(x4, v[12:16]))
        total = slab if total is None else f32(total + slab)
    return total


def test_oracle_enoki_order_equals_an_independent_restatement(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(77)
    differ = 0
    for trial in range(400):
        c = (rng.standard_normal(64) * 10.0 ** rng.integers(-3, 2)).astype(np.float32)
        f = rng.random(3).astype(np.float32)
        if trial % 7 == 0:
            f[rng.integers(0, 3)] = np.float32(rng.choice([0.0, 0.5, 1.0]))          # the rules evaluate at 0, 0.5, 1
        got = np.float32(L.orc_tricubic_value_enoki(c.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p)))
        want = _enoki_value_numpy(c, f)
        assert got.view(np.uint32) == want.view(np.uint32), (trial, got, want)
        lit = np.float32(L.orc_tricubic_value_literal(c.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p)))
        assert abs(float(lit) - float(got)) <= 1e-4 * (1.0 + np.abs(c).sum())          # the same polynomial
        differ += int(lit.view(np.uint32) != got.view(np.uint32))
    assert differ > 50                                                                    # ... in another rounding order


def test_both_oracle_flavours_load_and_say_which_they_are(oracle):
    for name, flavour in (("libsdf_oracle.so", 0), ("libsdf_oracle_enoki.so", 1)):
        L = C.CDLL(os.path.join(ROOT, "oracle", name))
        L.orc_interpolation_flavour.restype = C.c_int
        assert L.orc_interpolation_flavour() == flavour


_CHILD = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import sdflib_amd as S
from sdflib_amd._lib import lib
from oracle import pyoracle as O
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
assert lib().sdfhip_interpolation_flavour() == 1 and O.lib().orc_interpolation_flavour() == 1
b = lambda a: np.ascontiguousarray(a).view(np.uint32)
v, f = bumpy_icosphere(4); box = box_with_margin(v)
m = S.Mesh(v, f); om = O.Mesh(v, f)
pts = random_points_in_box(box, 200001, seed=5); pts[::97] *= 2.0
ins = np.ones(len(pts), bool); ins[::97] = False          # (the gradient of a point outside the grid is not defined by the reference)
for alg, cont in ((S.ALG_NO_CONTINUITY, False), (S.ALG_CONTINUITY, True)):
    for rule in (O.RULE_TRAPEZOIDAL, O.RULE_SIMPSONS):
        t = S.OctreeSdf(m, box, 6, 2, 1e-3, init_algorithm=alg, termination_rule=rule, num_threads=2)          # (2 threads = the reference's subtree layout, the oracle's default)
        o = O.Octree(om, box, 6, 2, 1e-3, rule=rule, continuity=cont)
        assert np.array_equal(t.get_octree_data(), o.data()), (alg, rule)
        assert b(np.float32(t.info.min_border_value)) == b(np.float32(o.min_border))
        d0, g0 = o.query(pts, grad=True)
        d, g = t.get_distance(pts, gradient=True)
        assert np.array_equal(b(d), b(d0)) and np.array_equal(b(g[ins]), b(g0[ins])), (alg, rule)
        assert np.array_equal(b(t.get_distance(pts)), b(d0))
        assert np.array_equal(b(t.get_distance(pts[:9])), b(d0[:9]))          # the host path of small batches
# the leaf-driven lattice against the point kernel
t = S.OctreeSdf(m, box, 6, 2, 1e-3, num_threads=2)
bb = t.get_grid_bounding_box(); n = 96
org = bb[:3] + 0.37 * (bb[3:] - bb[:3]) / n; step = (bb[3:] - bb[:3]) / n * 0.99
dg, gg = t.get_distance_grid(org, step, (n, n, n), gradient=True)
k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
P = (org[None, :].astype(np.float32) + np.stack([i, j, k], -1).reshape(-1, 3).astype(np.float32) * step[None, :].astype(np.float32)).astype(np.float32)
dp, gp = t.get_distance(P, gradient=True)
assert np.array_equal(b(dg.reshape(-1)), b(dp)) and np.array_equal(b(gg.reshape(-1, 3)), b(gp))
# the reference-compatible C++ classes linked against libsdfhip_enoki.so (INTEGRATION.md: the flavour is a link-time choice)
import os, subprocess, tempfile
root = %r
exe = "/tmp/sdflib_amd_test_cpp_api_enoki"; libdir = os.path.join(root, "sdflib_amd")
subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "test_cpp_api.cpp"),
                       "-L", libdir, "-lsdfhip_enoki", "-Wl,-rpath," + libdir, "-o", exe])
v2, f2 = bumpy_icosphere(2); box2 = box_with_margin(v2); pts2 = random_points_in_box(box2, 5000, seed=17); pts2[:50] *= 3.0
with tempfile.TemporaryDirectory() as td:
    q = lambda n: os.path.join(td, n)
    v2.tofile(q("v.bin")); f2.tofile(q("f.bin")); pts2.tofile(q("p.bin"))
    env = dict(os.environ); env.pop("SDFLIB_DEVICES", None)
    r = subprocess.run([exe, q("v.bin"), q("f.bin"), q("p.bin"), q("d.bin"), q("e.bin"), q("oct.bin"), q("exact.bin")], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "scalar-vs-batched mismatches 0" in r.stdout, r.stdout + r.stderr
    oc2 = O.Octree(O.Mesh(v2, f2), box2, 5, 2, 1e-3, vertex_cache=False, layout=O.LAYOUT_SUBTREES)
    assert np.array_equal(b(np.fromfile(q("d.bin"), dtype=np.float32)), b(oc2.query(pts2)))
print("enoki flavour ok")
''' % (ROOT, ROOT)


@pytest.mark.gpu
def test_enoki_flavour_library_matches_the_enoki_flavour_oracle():
    """Both builders, two rules, getDistance with and without gradient, the small-batch host path and the leaf-driven lattice, in a child
    process (the flavour is chosen when the libraries are loaded: SDFLIB_USE_ENOKI=1)."""
    r = subprocess.run([sys.executable, "-c", _CHILD], capture_output=True, text=True, env=dict(os.environ, SDFLIB_USE_ENOKI="1"), timeout=900)
    assert r.returncode == 0 and "enoki flavour ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
