"""CPU tests that pin the ORACLE (oracle/) before it is trusted as the checker.

The reference ships no golden vectors and cannot be compiled here (glm/cereal/spdlog are not vendored), so the
oracle is pinned by (a) the reference's own tolerance tests restated (TriangleDistanceTest, SdfOctreeTest-style
error bounds), (b) independent ground truths (analytic sphere, brute force), (c) the expression-level check of
tools/check_ref_expressions.py when /root/reference is present, and (d) committed golden vectors (regression)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, bits
from sdflib_amd.meshgen import icosphere, bumpy_icosphere, cube_mesh, box_with_margin, random_points_in_box


def test_triangle_distance_kat(oracle):
    """src/tools/TriangleDistanceTest/main.cpp:12-64: raw-vertex vs TriangleData squared distance within 1e-3,
    and signed^2 == squared distance, for points in [-1,1]^3 around one fixed triangle."""
    tri = np.array([[-.5, -.5, 0], [.5, -.5, 0], [0, .5, 0]], dtype=np.float32)
    m = oracle.Mesh(tri, np.array([[0, 1, 2]], dtype=np.uint32))
    rng = np.random.default_rng(2222)
    pts = (rng.random((20000, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    for p in pts:
        a = oracle.sqdist_raw(p, tri[0], tri[1], tri[2])
        b = m.sqdist(0, p)
        s = m.signed(0, p)
        assert abs(a - b) < 1e-3
        assert abs(s * s - b) < 1e-3


def test_reference_triangle_distance_test_full_run(oracle):
    """The reference's own TriangleDistanceTest, restated loop for loop: srand(2222), 1 M points from rand(), both asserts
    (src/tools/TriangleDistanceTest/main.cpp:12-64; the survey ran the real tool: all 1 M points pass)."""
    bad, max_raw, max_signed, sum_raw, sum_data = oracle.triangle_distance_test(2222, 1000000)
    assert bad == 0
    assert max_raw < 1e-5 and max_signed < 1e-5           # far inside the tool's 1e-3 tolerance
    assert abs(sum_raw - sum_data) < 1e-3 * abs(sum_raw)


def test_fit_matrix_is_inverse_of_hermite_constraints(oracle):
    """Derivation check (src/tools/CalculateInterpolationParameters/main.cpp:22-143): M @ C == I where C maps the
    64 monomial coefficients to the 64 Hermite values (value + 7 derivatives at the 8 unit-cube corners)."""
    M = oracle.fit_matrix().astype(np.int64)
    slot = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]

    def dmono(p, e, x):      # d^e/dx^e of x^p at x in {0,1}
        if e == 0:
            return 1 if (p == 0 or x == 1) else 0
        return p * (1 if (p == 1 or x == 1) else 0) if p >= 1 else 0

    Cm = np.zeros((64, 64), dtype=np.int64)
    for v in range(8):
        bx, by, bz = v & 1, (v >> 1) & 1, v >> 2
        for q, (ex, ey, ez) in enumerate(slot):
            for n in range(64):
                i, j, k = n & 3, (n >> 2) & 3, n >> 4
                Cm[8 * v + q, n] = dmono(i, ex, bx) * dmono(j, ey, by) * dmono(k, ez, bz)
    assert np.array_equal(M @ Cm, np.eye(64, dtype=np.int64))


def test_tricubic_fit_reproduces_polynomial(oracle):
    rng = np.random.default_rng(1)
    c = rng.standard_normal(64)
    size = 0.37

    def ev(x, y, z, ex, ey, ez):
        tot = 0.0
        for n in range(64):
            i, j, k = n & 3, (n >> 2) & 3, n >> 4
            def t(p, e, u):
                if e > p: return 0.0
                return (p if e else 1) * u ** (p - e)
            tot += c[n] * t(i, ex, x) * t(j, ey, y) * t(k, ez, z)
        return tot / size ** (ex + ey + ez)
    slot = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
    vals = np.array([[ev(v & 1, (v >> 1) & 1, v >> 2, *s) for s in slot] for v in range(8)], dtype=np.float32)
    got = oracle.tricubic_fit(vals, size)
    np.testing.assert_allclose(got, c, atol=2e-4)
    f = np.array([0.3, 0.6, 0.9], dtype=np.float32)
    assert abs(oracle.tricubic_value(got, f) - ev(*f, 0, 0, 0)) < 1e-4
    g = oracle.tricubic_gradient(got, f)
    np.testing.assert_allclose(g, [ev(*f, 1, 0, 0) * size, ev(*f, 0, 1, 0) * size, ev(*f, 0, 0, 1) * size], atol=1e-3)
    vv = oracle.tricubic_vertex_values(got, f, size)
    np.testing.assert_allclose(vv, [ev(*f, *s) for s in slot], rtol=2e-3, atol=2e-3)


def test_reference_expression_rules_when_reference_present():
    if not os.path.exists("/root/reference/include/SdfLib/InterpolationMethods.h"):
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU box: /root/reference does not travel")
        pytest.fail("/root/reference is absent: the reference-text pin cannot run (not the GPU box, so not skipped)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ref_expressions.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "8 expressions identical" in r.stdout


def test_bvh_nearest_equals_bruteforce_distance(oracle):
    v, f = bumpy_icosphere(2)
    m = oracle.Mesh(v, f)
    rng = np.random.default_rng(5)
    pts = ((rng.random((300, 3), dtype=np.float32) * 2 - 1) * 1.5).astype(np.float32)
    ids, d = m.nearest(pts, with_dist=True)
    for i, p in enumerate(pts):
        ds = np.array([m.sqdist(t, p) for t in range(len(f))], dtype=np.float64)
        assert abs(np.sqrt(ds.min()) - d[i]) < 2e-6
        assert abs(np.sqrt(ds[ids[i]]) - d[i]) < 2e-6          # the returned id attains the minimum (up to ties)


def test_signed_distance_of_sphere_and_cube(oracle):
    v, f = icosphere(4)
    m = oracle.Mesh(v, f)
    rng = np.random.default_rng(6)
    pts = ((rng.random((3000, 3), dtype=np.float32) * 2 - 1) * 1.3).astype(np.float32)
    ids = m.nearest(pts)
    sd = np.array([m.signed(ids[i], pts[i]) for i in range(len(pts))])
    an = np.linalg.norm(pts, axis=1) - 1.0
    assert np.abs(sd - an).max() < 6e-3                       # polyhedron vs sphere (chord error of s=4)
    far = np.abs(an) > 0.02
    assert (np.sign(sd[far]) == np.sign(an[far])).all()
    v, f = cube_mesh()
    m = oracle.Mesh(v, f)
    ids = m.nearest(pts)
    sd = np.array([m.signed(ids[i], pts[i]) for i in range(len(pts))])
    q = np.abs(pts) - 0.5
    an = np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0)
    assert np.abs(sd - an).max() < 2e-6


def _walk(data, G3):
    """canonical pre-order (children 0..7) topology + leaf payloads of an OctreeSdf array"""
    topo, leaves = [], []
    stack = list(range(G3 - 1, -1, -1))
    while stack:
        at = stack.pop()
        w = int(data[at])
        if w & 0x80000000:
            topo.append(1); leaves.append(data[(w & 0x3FFFFFFF):(w & 0x3FFFFFFF) + 64])
        else:
            topo.append(0); stack.extend(range((w & 0x3FFFFFFF) + 7, (w & 0x3FFFFFFF) - 1, -1))
    return np.array(topo, dtype=np.uint8), np.stack(leaves)


def test_octree_layouts_hold_the_same_tree_and_bound_the_error(oracle):
    v, f = bumpy_icosphere(3)
    box = box_with_margin(v)
    m = oracle.Mesh(v, f)
    a = oracle.Octree(m, box, 5, 2, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    b = oracle.Octree(m, box, 5, 2, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_GLOBAL_DFS)
    ta, la = _walk(a.data(), 64); tb, lb = _walk(b.data(), 64)
    assert np.array_equal(ta, tb) and np.array_equal(la, lb)
    assert a.value_range == b.value_range and a.min_border == b.min_border
    # 1-thread emulation (lattice cache on) stays within the reference's own 1-vs-N-thread spread
    c = oracle.Octree(m, box, 5, 2, 1e-3, vertex_cache=True, layout=oracle.LAYOUT_GLOBAL_DFS)
    pts = random_points_in_box(box, 20000, seed=3)
    da, dc = a.query(pts), c.query(pts)
    assert np.abs(da - dc).max() < 5e-3
    # SdfOctreeTest-style bound (src/tools/SdfOctreeTest/main.cpp:37-76): octree vs exact distance
    ids = m.nearest(pts[:4000])
    ex = np.array([m.signed(ids[i], pts[i]) for i in range(4000)])
    err = np.abs(da[:4000] - ex)
    assert np.sqrt((err ** 2).mean()) < 2e-3 and err.max() < 3e-2
    # outside the box: box distance + min border value
    out = oracle.Octree.query(a, np.array([[5, 0, 0]], dtype=np.float32))
    assert out[0] > 3.0


def test_exact_octree_equals_bruteforce_bit_exact(oracle):
    v, f = bumpy_icosphere(3)
    box = box_with_margin(v)
    m = oracle.Mesh(v, f)
    pts = random_points_in_box(box, 3000, seed=4)
    brute = []
    for p in pts:
        ds = np.array([m.sqdist(t, p) for t in range(len(f))], dtype=np.float32)
        brute.append(m.signed(int(np.argmin(ds)), p))           # first minimum in ascending id, like the reference
    brute = np.array(brute, dtype=np.float32)
    for start, cache in ((1, False), (3, False), (1, True)):
        ex = oracle.Exact(m, box, 6, start, 32, vertex_cache=cache)
        d = ex.query(pts)
        assert np.array_equal(bits(d), bits(brute)), (start, cache)
        assert ex.max_tri_in_leafs >= 32 or ex.num_nodes > 0


def test_is_near_minimize_accepts_touching_and_rejects_far(oracle):
    half = 0.5
    radius = np.zeros(8, dtype=np.float32)
    near = np.array([[0.1, 0.1, 0.1], [0.3, 0.1, 0.1], [0.1, 0.3, 0.2]], dtype=np.float32)       # inside the box
    far = near + np.float32(10.0)
    assert oracle.is_near_minimize(half, radius, near, 0.05)[0] is True
    assert oracle.is_near_minimize(half, radius, far, 0.5)[0] is False
    assert oracle.is_near_minimize(half, radius, far, 20.0)[0] is True                            # threshold larger than the gap


def test_golden_vectors(oracle):
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_small.npz"))
    v, f = bumpy_icosphere(int(g["subdiv"]))
    assert np.array_equal(v, g["vertices"]) and np.array_equal(f, g["triangles"])
    m = oracle.Mesh(v, f)
    assert np.array_equal(bits(m.triangle_data()), bits(g["triangle_data"]))
    # reference outputs (recorded from oracle/_ref = the reference's own TriangleMeshDistance.h, see make_golden.py): ids of
    # random points, ids / distances of points exactly on vertices, edges and faces, and the reference's whole BVH
    assert str(g["nearest_ids_source"]) == "reference:TriangleMeshDistance.h"
    assert np.array_equal(m.nearest(g["points"]), g["nearest_ids"])
    tid, td = m.nearest(g["tie_points"], with_dist=True)
    assert np.array_equal(tid, g["tie_nearest_ids"])
    assert np.array_equal(td.view(np.uint64), np.abs(g["tie_distances"]).view(np.uint64))
    sph, lr = m.bvh_export()
    assert np.array_equal(lr, g["bvh_children"])
    inner = lr[:, 0] != -1
    assert np.array_equal(np.asarray(sph)[inner].view(np.uint64), g["bvh_spheres"][inner].view(np.uint64))
    oc = oracle.Octree(m, g["box"], int(g["depth"]), int(g["start_depth"]), 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    assert np.array_equal(oc.data(), g["octree_words"])
    d, gr = oc.query(g["points"], grad=True)
    assert np.array_equal(bits(d), bits(g["octree_dist"])) and np.array_equal(bits(gr), bits(g["octree_grad"]))
    ex = oracle.Exact(m, g["box"], int(g["exact_depth"]), 1, int(g["exact_min_tri"]))
    nodes, has, sets, masks = ex.data()
    assert np.array_equal(nodes[:, 0], g["exact_nodes"][:, 0]) and np.array_equal(sets, g["exact_sets"]) and np.array_equal(masks, g["exact_masks"])
    assert np.array_equal(bits(ex.query(g["points"])), bits(g["exact_dist"]))


def test_oracle_ids_equal_the_reference_fixture_at_full_size(oracle):
    """tests/golden/ref_nearest_large.npz = ids recorded from the REAL reference (oracle/_ref, make_golden.py) on the BASELINE
    configs' meshes; needs no /root/reference, so it also guards the oracle on the GPU box."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from refpoints import ref_fixture_cases, points_digest
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_nearest_large.npz"))
    assert str(g["source"]) == "reference:TriangleMeshDistance.h"
    for name, v, f, pts in ref_fixture_cases()[:2]:            # the 1.31 M-triangle case is covered by test_oracle_ref_pin.py and the GPU suite
        assert points_digest(pts) == int(g[name + "_digest"])
        assert np.array_equal(oracle.Mesh(v, f).nearest(pts), g[name + "_ids"]), name


def _leaf_histogram(data, G3, start):
    hist = {}
    stack = [(i, start) for i in range(G3)]
    while stack:
        at, d = stack.pop()
        w = int(data[at])
        assert not (w & 0x40000000), "mark bit left set"
        if w & 0x80000000:
            hist[d] = hist.get(d, 0) + 1
        else:
            stack.extend(((w & 0x3FFFFFFF) + c, d + 1) for c in range(8))
    return hist


def test_continuity_builder_reproduces_the_reference_probe_counts(oracle):
    """The only numbers the survey measured on the REAL reference (shim-compiled, SURVEY.md section 6 / BASELINE.md P1, P18):
    CONTINUITY build of the 5 120-triangle bumpy sphere, depth 6, start 3, thr 1e-3 -> 10.76 M words, leaves per depth
    d4 252 / d5 11 582 / d6 153 360.  The mesh here has the same geometry but another triangle order, so agreement is
    expected up to a few borderline node flips (the reference's own 1-vs-N-thread spread is of that size)."""
    v, f = bumpy_icosphere(4)
    box = box_with_margin(v)
    m = oracle.Mesh(v, f)
    oc = oracle.Octree(m, box, 6, 3, 1e-3, continuity=True)
    data = oc.data()
    hist = _leaf_histogram(data, 512, 3)
    assert abs(len(data) - 10.76e6) < 0.01e6
    assert hist[4] == 252 and abs(hist[5] - 11582) <= 5 and abs(hist[6] - 153360) <= 40
    # NO_CONTINUITY on the same input is a coarser tree, and both approximate the same field
    on = oracle.Octree(m, box, 6, 3, 1e-3)
    assert len(on.data()) < len(data)
    pts = random_points_in_box(box, 20000, seed=8)
    assert np.abs(oc.query(pts) - on.query(pts)).max() < 1e-2


def test_seam_welding_repairs_a_triangle_soup(oracle):
    """calculateMeshTriangleData's non-manifold pass (TriangleUtils.cpp:292-420): with the loader's bounding box, a
    soup of disconnected triangles gets the same edge pseudonormals as the indexed mesh and the same vertex
    pseudonormals up to summation order; without the box (raw-pointer Mesh ctor) seams keep the default (0,0,1)."""
    from sdflib_amd.meshgen import icosphere, triangle_soup
    v, f = icosphere(2)
    sv, sf = triangle_soup(v, f)
    bbox = np.concatenate([sv.min(axis=0), sv.max(axis=0)])
    ref = oracle.Mesh(v, f).triangle_data()
    raw = oracle.Mesh(sv, sf).triangle_data()
    welded = oracle.Mesh(sv, sf, bbox).triangle_data()
    assert np.array_equal(raw[:, 19:28], np.tile(np.float32([0, 0, 1]), (len(f), 3)))
    np.testing.assert_array_equal(welded[:, :19], ref[:, :19])
    np.testing.assert_allclose(welded[:, 19:28], ref[:, 19:28], rtol=0, atol=1e-6)       # n1+n2 vs n2+n1 is exact; transform identical
    np.testing.assert_allclose(welded[:, 28:], ref[:, 28:], rtol=0, atol=2e-6)
    # signed distance through the welded data has the right sign everywhere around the sphere
    m = oracle.Mesh(sv, sf, bbox)
    rng = np.random.default_rng(5)
    pts = ((rng.random((3000, 3), dtype=np.float32) * 2 - 1) * 1.3).astype(np.float32)
    ids = m.nearest(pts)
    sd = np.array([m.signed(ids[i], pts[i]) for i in range(len(pts))])
    r = np.linalg.norm(pts, axis=1)
    far = np.abs(r - 1.0) > 0.05
    assert np.array_equal(np.sign(sd[far]), np.sign(r[far] - 1.0))


@pytest.mark.parametrize("subdiv,depth,start,min_tri", [(3, 5, 1, 16), (3, 5, 2, 16), (4, 6, 3, 32), (3, 4, 2, 8), (2, 5, 3, 4)])
def test_parallel_exact_build_equals_the_sequential_one(oracle, subdiv, depth, start, min_tri):
    """The OpenMP-over-start-cells Exact build (used by the full-size GPU parity test to keep the oracle under a minute) must write
    the very arrays of the sequential canonical build: nodes, written-flags, bit-packed sets, byte masks, leaf statistics, cull count."""
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
    v, f = bumpy_icosphere(subdiv)
    box = box_with_margin(v)
    om = oracle.Mesh(v, f)
    a = oracle.Exact(om, box, depth, start, min_tri, threads=1)
    for threads in (3, 0):
        b = oracle.Exact(om, box, depth, start, min_tri, threads=threads)
        assert (a.num_nodes, a.num_set_words, a.num_mask_bytes, a.max_tri_in_leafs, a.max_tri_encoded, a.cull_tests) == \
               (b.num_nodes, b.num_set_words, b.num_mask_bytes, b.max_tri_in_leafs, b.max_tri_encoded, b.cull_tests)
        for x, y in zip(a.data(), b.data()):
            assert np.array_equal(x, y)


def test_continuity_oracle_is_thread_count_invariant(oracle, monkeypatch):
    """Iter 1 of the CONTINUITY oracle runs under OpenMP (like the reference's); canonical mode must not depend on the thread count."""
    import ctypes
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
    v, f = bumpy_icosphere(3)
    box = box_with_margin(v)
    om = oracle.Mesh(v, f)
    omp = ctypes.CDLL("libgomp.so.1")
    outs = []
    for threads in (1, 5):
        omp.omp_set_num_threads(threads)
        t = oracle.Octree(om, box, 5, 2, 1e-3, continuity=True)
        outs.append((t.data(), t.value_range, t.min_border, t.num_bvh_queries))
    omp.omp_set_num_threads(os.cpu_count() or 1)
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]
