"""Pins SURVEY.md §8 row a5 of the ORACLE to the REAL reference, compiled here with no stand-ins.

oracle/_ref/libtmd_ref.so is tmd::TriangleMeshDistance from
/root/reference/libs/InteractiveComputerGraphics/InteractiveComputerGraphics/TriangleMeshDistance.h (it needs only the
C++ standard library) behind oracle/ref_tmd.cpp, constructed and queried exactly as SdfLib's ICG wrapper does
(include/SdfLib/TrianglesInfluence.h:884-905).  These tests fail if oracle/orc_bvh.h — the restatement every build test
trusts for "which triangle is nearest" — diverges from that header in any bit: the whole BVH node array (children,
leaf triangles, all eight doubles of every inner node's child spheres), the triangle ids and the distances, on every
fixture mesh, incl. points lying exactly on vertices / edges / faces (ties decided by the visiting order) and the
1.31 M-triangle mesh of BASELINE configs[3].  The product's host planner is compared with the reference directly too.

CPU only.  Needs /root/reference (build container) or a prebuilt oracle/_ref that travelled; skipped otherwise."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pyref  # noqa: E402
from sdflib_amd import meshgen  # noqa: E402
from refpoints import tie_points  # noqa: E402

pytestmark = pytest.mark.skipif(not pyref.available(), reason="oracle/_ref/libtmd_ref.so absent and /root/reference not present")


def check_mesh(oracle, v, f, rng, n_random=20000, what=""):
    v = np.ascontiguousarray(v, np.float32); f = np.ascontiguousarray(f, np.uint32)
    ref = pyref.RefMesh(v, f)
    om = oracle.Mesh(v, f)
    rs, rlr, _ = ref.bvh_export()
    os_, olr = om.bvh_export()
    assert rlr.shape == olr.shape, what
    assert np.array_equal(rlr, olr), f"{what}: BVH children / leaf triangles differ from the reference"
    inner = rlr[:, 0] != -1                                # a leaf's two spheres are never written (nor read) by the reference
    assert np.array_equal(rs[inner].view(np.uint64), np.asarray(os_)[inner].view(np.uint64)), f"{what}: BVH spheres differ from the reference"
    pts = tie_points(v, f, rng, n_random)
    rid, rd = ref.nearest(pts, with_dist=True)
    oid, od = om.nearest(pts, with_dist=True)
    bad = np.nonzero(rid != oid)[0]
    assert len(bad) == 0, f"{what}: {len(bad)} of {len(pts)} nearest-triangle ids differ from the reference, first at {pts[bad[0]]}: {rid[bad[0]]} vs {oid[bad[0]]}"
    assert np.array_equal(np.abs(rd).view(np.uint64), np.abs(od).view(np.uint64)), f"{what}: distances differ from the reference"
    return len(pts)


def test_orc_bvh_equals_the_reference_on_the_fixture_meshes(oracle):
    rng = np.random.default_rng(77)
    total = 0
    v, f = meshgen.icosphere(3)
    total += check_mesh(oracle, v, f, rng, what="icosphere s=3 (symmetric: tied sort keys on every axis)")
    for s in (2, 4, 5):
        v, f = meshgen.bumpy_icosphere(s)
        total += check_mesh(oracle, v, f, rng, what=f"bumpy icosphere s={s}")
    v, f = meshgen.bumpy_icosphere(4)
    total += check_mesh(oracle, v * np.array([1e-3, 7.0, 250.0], np.float32) + np.array([1e3, -2.0, 0.5], np.float32), f, rng, what="scaled / offset")
    total += check_mesh(oracle, v * np.float32(1e-3), f, rng, what="scale 1e-3")
    total += check_mesh(oracle, v * np.float32(1e3) + np.float32(5e4), f, rng, what="scale 1e3, offset 5e4")
    total += check_mesh(oracle, v, np.concatenate([f, f[::3], f[::5]]), rng, what="duplicated triangles")
    total += check_mesh(oracle, v, f[rng.permutation(len(f))], rng, what="permuted triangle order")
    total += check_mesh(oracle, v, f[:, [1, 2, 0]], rng, what="rotated corners (other sort keys)")
    cv, cf = meshgen.cube_mesh()
    total += check_mesh(oracle, cv, cf, rng, what="cube")
    total += check_mesh(oracle, cv, np.concatenate([cf, np.array([[0, 0, 1], [2, 2, 2]], np.uint32)]), rng, what="cube + degenerate triangles")
    total += check_mesh(oracle, cv, cf[:2], rng, what="two triangles")
    sv, sf = meshgen.triangle_soup(v, f[:700])
    total += check_mesh(oracle, sv, sf, rng, what="triangle soup")
    kv, kf = meshgen.torus_knot(nu=256, nv=40)
    total += check_mesh(oracle, kv, kf, rng, what="torus knot (non-star-shaped)")
    # hemisphere with an open rim (non-manifold boundary) and a sliver fan
    hv, hf = meshgen.icosphere(3)
    total += check_mesh(oracle, hv, hf[hv[hf].mean(axis=1)[:, 2] > 0], rng, what="open hemisphere")
    fan_v = np.array([[0, 0, 0]] + [[np.cos(a), np.sin(a), 1e-4 * i] for i, a in enumerate(np.linspace(0, 1e-2, 60))], np.float32)
    fan_f = np.array([[0, i, i + 1] for i in range(1, 60)], np.uint32)
    total += check_mesh(oracle, fan_v, fan_f, rng, what="sliver fan (valence-59 vertex)")
    assert total > 400000


def test_orc_bvh_equals_the_reference_at_c2_size(oracle):
    """BASELINE configs[1]/[2] stand-in: 327 680 triangles."""
    v, f = meshgen.bumpy_icosphere(7)
    check_mesh(oracle, v, f, np.random.default_rng(3), n_random=100000, what="bumpy icosphere s=7")
    v, f = meshgen.torus_knot()
    check_mesh(oracle, v, f, np.random.default_rng(4), n_random=100000, what="torus knot 327 680 triangles")


def test_orc_bvh_equals_the_reference_at_1m_triangles(oracle):
    """BASELINE configs[3]: the 1.31 M-triangle mesh."""
    v, f = meshgen.bumpy_icosphere(8)
    check_mesh(oracle, v, f, np.random.default_rng(5), n_random=150000, what="bumpy icosphere s=8 (1 310 720 triangles)")


def test_product_planner_equals_the_reference_tree():
    """The PRODUCT's host planner (sdflib_amd/csrc/bvh.hip, no device needed) against the reference's own tree:
    walked together from the root, all 64 bits of every child sphere, every leaf's triangle."""
    from test_planner_cpu import planned
    for v, f in (meshgen.bumpy_icosphere(5), meshgen.torus_knot(nu=256, nv=40), meshgen.cube_mesh(), meshgen.bumpy_icosphere(7)):
        v = np.ascontiguousarray(v, np.float32); f = np.ascontiguousarray(f, np.uint32)
        sph, kids = planned(v, f)
        rs, rlr, _ = pyref.RefMesh(v, f).bvh_export()
        stack = [(0, 0)]; leaves = 0
        while stack:
            a, b = stack.pop()
            assert rlr[b, 0] >= 0
            assert np.array_equal(sph[8 * a:8 * a + 8].view(np.uint64), rs[b].view(np.uint64)), (a, b)
            for side in (0, 1):
                ours, theirs = int(kids[2 * a + side]), int(rlr[b, side])
                if ours < 0:
                    assert rlr[theirs, 0] == -1 and rlr[theirs, 1] == ~ours
                    leaves += 1
                else:
                    stack.append((ours, theirs))
        assert leaves == len(f)


def test_golden_fixture_ids_come_from_the_reference():
    """tests/golden/golden_small.npz records `nearest_ids_source`; when it says the reference, the committed ids must be
    what the reference compiled here returns today."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_small.npz"))
    assert str(g["nearest_ids_source"]) == "reference:TriangleMeshDistance.h"
    ref = pyref.RefMesh(g["vertices"], g["triangles"])
    assert np.array_equal(ref.nearest(g["points"]), g["nearest_ids"])
    assert np.array_equal(ref.nearest(g["tie_points"]), g["tie_nearest_ids"])
