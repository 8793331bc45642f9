"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import os
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _mesh(s):
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
    v, f = bumpy_icosphere(s)
    return v, f, box_with_margin(v)


@pytest.fixture(scope="module")
def small(oracle, gpu_ctx):
    import sdflib_amd as S
    v, f, box = _mesh(3)
    return dict(v=v, f=f, box=box, om=oracle.Mesh(v, f), gm=S.Mesh(v, f, gpu_ctx))


def test_triangle_data_matches_oracle(small):
    a = small["om"].triangle_data()
    b = small["gm"].triangle_data()
    # frames, edge pseudonormals AND vertex pseudonormals: bit-exact (the corner angles' acosf is the host libm's, as in the reference)
    assert np.array_equal(bits(a), bits(b))


def test_device_acosf_equals_libm(gpu_ctx):
    """dev_math.h::acosfGlibc as the DEVICE compiles it (IEEE sqrt and division) against the host's libm: every 64th float and the
    neighbourhoods of the routine's branch points (+-0.5, +-1, 2^-57); tests/test_abi.py checks the host compilation on every float."""
    import ctypes as C
    from sdflib_amd._lib import lib, check
    bad = C.c_uint64(0)
    for first, stride, count in ((0, 64, 1 << 26), (0x3f000000 - (1 << 20), 1, 1 << 21), (0xbf000000 - (1 << 20), 1, 1 << 21),
                                 (0x3f800000 - (1 << 21), 1, (1 << 21) + 16), (0xbf800000 - (1 << 21), 1, (1 << 21) + 16), (0x23000000 - 4096, 1, 8192), (0xa3000000 - 4096, 1, 8192)):
        check(lib().sdfhip_test_acosf_device(gpu_ctx.h, first, stride, count, C.byref(bad)))
        assert bad.value == 0, (hex(first), stride, count, bad.value)


def test_host_acos_path_gives_the_same_triangle_data(gpu_ctx, oracle):
    """The arc cosines of the corner angles on the host's libm (the path a process takes by itself when the self-check of the restated
    glibc routine against the running libm fails) and on the device give the same TriangleData, bit for bit — and the oracle's."""
    import sdflib_amd as S
    from sdflib_amd._lib import lib
    from sdflib_amd import meshgen
    v, f = meshgen.bumpy_icosphere(3)
    ref = oracle.Mesh(v, f).triangle_data()
    try:
        for mode in (1, 0, -1):
            lib().sdfhip_test_set_host_acos(mode)
            td = S.Mesh(v, f, gpu_ctx).triangle_data()
            assert np.array_equal(bits(ref), bits(td)), mode
    finally:
        lib().sdfhip_test_set_host_acos(-1)


def test_nearest_triangle_ids_bit_exact(small, oracle):
    rng = np.random.default_rng(7)
    pts = ((rng.random((20000, 3), dtype=np.float32) * 2 - 1) * 1.4).astype(np.float32)
    ids_o = small["om"].nearest(pts)
    ids_g = small["gm"].nearest_triangle(pts)
    assert np.array_equal(ids_o, ids_g)


def test_point_values_match(small):
    rng = np.random.default_rng(8)
    pts = ((rng.random((5000, 3), dtype=np.float32) * 2 - 1) * 1.4).astype(np.float32)
    ids = small["om"].nearest(pts)
    a = small["om"].point_values(pts, ids)
    b = small["gm"].point_values(pts, ids)
    assert np.array_equal(bits(a), bits(b))


def test_tricubic_fit_bit_exact(oracle, gpu_ctx):
    import sdflib_amd as S
    rng = np.random.default_rng(3)
    vals = (rng.standard_normal((512, 8, 8))).astype(np.float32)
    ns = (0.01 + rng.random(512)).astype(np.float32)
    ref = np.stack([oracle.tricubic_fit(vals[i], ns[i]) for i in range(512)])
    got = S.tricubic_fit(vals, ns, gpu_ctx)
    assert np.array_equal(bits(ref), bits(got))


@pytest.mark.parametrize("depth,start,layout", [(5, 2, 1), (5, 2, 0), (4, 0, 1), (4, 1, 0), (6, 3, 1)])
def test_octree_build_bit_exact_topology_and_coefficients(small, oracle, depth, start, layout):
    import sdflib_amd as S
    oc = oracle.Octree(small["om"], small["box"], depth, start, 1e-3, vertex_cache=False, layout=layout)
    gt = S.OctreeSdf(small["gm"], small["box"], depth, start, 1e-3, num_threads=2 if layout == 1 else 1)
    a, b = oc.data(), gt.get_octree_data()
    assert a.shape == b.shape
    assert np.array_equal(a, b), f"first mismatch at word {np.flatnonzero(a != b)[:5]}"
    i = gt.info
    assert i.start_grid_size == oc.start_grid_size
    assert np.float32(i.value_range) == np.float32(oc.value_range)
    assert np.float32(i.min_border_value) == np.float32(oc.min_border)
    np.testing.assert_array_equal(np.array(list(i.box_min) + list(i.box_max), dtype=np.float32), oc.box)


def test_octree_query_exact_and_fast(small, oracle):
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    oc = oracle.Octree(small["om"], small["box"], 6, 3, 1e-3)
    gt = S.OctreeSdf(small["gm"], small["box"], 6, 3, 1e-3, num_threads=2)
    pts = random_points_in_box(small["box"], 200000, seed=11)
    # a few points outside the grid as well
    pts[:200] *= 3.0
    d_o, g_o = oc.query(pts, grad=True)
    d_g, g_g = gt.get_distance(pts, gradient=True, eval_mode=S.EVAL_EXACT)
    inside = np.ones(len(pts), bool); inside[:200] = False
    assert np.array_equal(bits(d_o), bits(d_g))
    assert np.array_equal(bits(g_o[inside]), bits(g_g[inside]))
    d_f, g_f = gt.get_distance(pts, gradient=True, eval_mode=S.EVAL_FAST)
    np.testing.assert_allclose(d_f, d_o, rtol=0, atol=1e-5)
    np.testing.assert_allclose(g_f[inside], g_o[inside], rtol=0, atol=2e-4)
    d_v = gt.get_distance(pts, eval_mode=S.EVAL_EXACT)
    assert np.array_equal(bits(d_v), bits(d_o))


def test_gpu_matches_committed_golden_vectors(gpu_ctx):
    """Checker-independent fixture (tests/golden/golden_small.npz, produced by tests/golden/make_golden.py)."""
    import os
    import sdflib_amd as S
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_small.npz"))
    gm = S.Mesh(g["vertices"], g["triangles"], gpu_ctx)
    td = gm.triangle_data()
    assert np.array_equal(bits(td[:, :28]), bits(g["triangle_data"][:, :28]))
    # nearest ids are REFERENCE outputs (oracle/_ref: the reference's TriangleMeshDistance.h compiled as it is, make_golden.py),
    # incl. points exactly on vertices / edges / faces where several triangles tie and the visiting order decides
    assert str(g["nearest_ids_source"]) == "reference:TriangleMeshDistance.h"
    assert np.array_equal(gm.nearest_triangle(g["points"]), g["nearest_ids"])
    assert np.array_equal(gm.nearest_triangle(g["tie_points"]), g["tie_nearest_ids"])
    t = S.OctreeSdf(gm, g["box"], int(g["depth"]), int(g["start_depth"]), 1e-3, num_threads=2)
    assert np.array_equal(t.get_octree_data(), g["octree_words"])
    assert np.float32(t.info.value_range) == g["octree_value_range"] and np.float32(t.info.min_border_value) == g["octree_min_border"]
    d, gr = t.get_distance(g["points"], gradient=True, eval_mode=S.EVAL_EXACT)
    assert np.array_equal(bits(d), bits(g["octree_dist"]))
    inside = np.ones(len(d), bool); inside[:64] = False
    assert np.array_equal(bits(gr[inside]), bits(g["octree_grad"][inside]))


def test_grid_query_matches_point_query(small):
    import sdflib_amd as S
    gt = S.OctreeSdf(small["gm"], small["box"], 5, 2, 1e-3, num_threads=2)
    bb = gt.get_grid_bounding_box()
    size = np.float32(bb[3] - bb[0]); n = 32
    step = np.full(3, size / np.float32(n), dtype=np.float32)
    origin = (bb[:3] + np.float32(0.5) * step).astype(np.float32)
    d, g = gt.get_distance_grid(origin, step, (n, n, n), gradient=True, eval_mode=S.EVAL_EXACT)
    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    pts = np.stack([origin[0] + i.astype(np.float32) * step[0], origin[1] + j.astype(np.float32) * step[1], origin[2] + k.astype(np.float32) * step[2]], axis=-1).reshape(-1, 3).astype(np.float32)
    d2, g2 = gt.get_distance(pts, gradient=True, eval_mode=S.EVAL_EXACT)
    assert np.array_equal(bits(d), bits(d2)) and np.array_equal(bits(g), bits(g2))
    # EVAL_FAST on lattices: inside the box, sticking out of it, odd sides
    for org, stp, dims in ((origin, step, (n, n, n)), (origin - 6 * step, (step * np.float32(1.5)).astype(np.float32), (32, 28, 36)), (origin, step, (31, 30, 29))):
        df, gf = gt.get_distance_grid(org, stp, dims, gradient=True, eval_mode=S.EVAL_FAST)
        df_only = gt.get_distance_grid(org, stp, dims, gradient=False, eval_mode=S.EVAL_FAST)
        kk, jj, ii = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
        q = np.stack([org[0] + ii.astype(np.float32) * stp[0], org[1] + jj.astype(np.float32) * stp[1], org[2] + kk.astype(np.float32) * stp[2]], axis=-1).reshape(-1, 3).astype(np.float32)
        de, ge = gt.get_distance(q, gradient=True, eval_mode=S.EVAL_EXACT)
        np.testing.assert_allclose(df, de, rtol=0, atol=1e-5)
        np.testing.assert_allclose(df_only, de, rtol=0, atol=1e-5)
        ok = np.isfinite(ge).all(axis=1) & np.isfinite(gf).all(axis=1)
        np.testing.assert_allclose(gf[ok], ge[ok], rtol=0, atol=2e-4)        # unit gradients: normalisation amplifies the 1e-5 of the components


def test_leaf_driven_lattice_equals_the_point_kernel_bit_for_bit(small):
    """EVAL_FAST lattices are answered leaf by leaf (octree_query.hip, k_lattice_columns): the leaf a lattice point lands in, its
    local coordinates and the order of the FMAs are the point kernel's, so the two paths must agree in every bit — on aligned and
    unaligned lattices, anisotropic steps, lattices sticking out of the box on either side, single-row lattices, a lattice far
    coarser than the leaves (the plan falls back to the point kernel) and a repeated call (cached plan)."""
    import sdflib_amd as S
    rng = np.random.default_rng(11)
    for depth, start in ((6, 2), (5, 0), (4, 3)):
        gt = S.OctreeSdf(small["gm"], small["box"], depth, start, 1e-3, num_threads=2)
        bb = gt.get_grid_bounding_box()
        size = np.float32(bb[3] - bb[0])
        cases = []
        n = 64
        st = np.full(3, size / np.float32(n), dtype=np.float32)
        cases.append((bb[:3] + np.float32(0.5) * st, st, (n, n, n)))                                 # cell centres of depth-6 cells
        cases.append((bb[:3].copy(), st, (n + 1, n + 1, n + 1)))                                     # cell corners: every point on a boundary
        cases.append((bb[:3] - 5 * st, (st * np.float32(1.37)).astype(np.float32), (70, 41, 55)))    # sticks out on both sides
        cases.append((bb[:3] + rng.uniform(0, 0.1, 3).astype(np.float32), (st * rng.uniform(0.3, 3.0, 3)).astype(np.float32), (33, 97, 18)))
        cases.append((bb[:3] + np.float32(0.3) * size, (st * np.float32(0.01)).astype(np.float32), (50, 1, 50)))   # a thin slab inside one or two leaves
        cases.append((bb[:3] + np.float32(0.01), np.full(3, size / np.float32(3), dtype=np.float32), (3, 3, 3)))   # coarser than any leaf
        cases.append(cases[0])
        for org, stp, dims in cases:
            org = np.asarray(org, dtype=np.float32); stp = np.asarray(stp, dtype=np.float32)
            kk, jj, ii = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
            q = np.stack([org[0] + ii.astype(np.float32) * stp[0], org[1] + jj.astype(np.float32) * stp[1], org[2] + kk.astype(np.float32) * stp[2]], axis=-1)
            q = q.reshape(-1, 3).astype(np.float32)
            dp, gp = gt.get_distance(q, gradient=True, eval_mode=S.EVAL_FAST)
            dl, gl = gt.get_distance_grid(org, stp, dims, gradient=True, eval_mode=S.EVAL_FAST)
            dv = gt.get_distance_grid(org, stp, dims, gradient=False, eval_mode=S.EVAL_FAST)
            dvp = gt.get_distance(q, gradient=False, eval_mode=S.EVAL_FAST)
            assert np.array_equal(bits(dl), bits(dp)), (depth, start, dims, int((bits(dl) != bits(dp)).sum()))
            assert np.array_equal(bits(dv), bits(dvp)), (depth, start, dims)
            same = bits(gl) == bits(gp)
            both_nan = np.isnan(gl) & np.isnan(gp)
            assert (same | both_nan).all(), (depth, start, dims, int((~(same | both_nan)).sum()))
            # EVAL_EXACT goes the same way (k_lattice_columns_exact: the z-independent prefix of every term of the reference's literal
            # sum is computed once per column): the reference-order point kernel is the yardstick
            de, ge = gt.get_distance(q, gradient=True, eval_mode=S.EVAL_EXACT)
            dle, gle = gt.get_distance_grid(org, stp, dims, gradient=True, eval_mode=S.EVAL_EXACT)
            dve = gt.get_distance_grid(org, stp, dims, gradient=False, eval_mode=S.EVAL_EXACT)
            assert np.array_equal(bits(dle), bits(de)) and np.array_equal(bits(dve), bits(de)), (depth, start, dims, int((bits(dle) != bits(de)).sum()))
            same = bits(gle) == bits(ge)
            both_nan = np.isnan(gle) & np.isnan(ge)
            assert (same | both_nan).all(), (depth, start, dims, int((~(same | both_nan)).sum()))


def test_sharded_build_emits_the_same_array(small, oracle):
    """Two shards built separately and concatenated by hand == the single-device array (no collective involved)."""
    import sdflib_amd as S
    full = S.OctreeSdf(small["gm"], small["box"], 5, 2, 1e-3, num_threads=2).get_octree_data()
    G3 = 64
    parts = [(0, 23), (23, 64)]
    shards = [S.OctreeShard(small["gm"], small["box"], 5, 2, 1e-3, cells=c) for c in parts]
    sizes = [s.info.body_words for s in shards]
    out = np.zeros(G3 + sum(sizes), dtype=np.uint32)
    off = G3
    for s, c, n in zip(shards, parts, sizes):
        grid = np.zeros(c[1] - c[0], dtype=np.uint32); body = np.zeros(n, dtype=np.uint32)
        s.emit(off, grid, body)
        out[c[0]:c[1]] = grid; out[off:off + n] = body
        off += n
    assert np.array_equal(out, full)
    assert max(s.info.value_range for s in shards) == S.OctreeSdf(small["gm"], small["box"], 5, 2, 1e-3, num_threads=2).info.value_range


def test_mfma_fit_is_the_correctly_rounded_product(oracle, gpu_ctx):
    """FIT_MFMA (hi/lo split, exact hi pass) reproduces the fp64 product to ~1 ulp; the reference-ordered fp32 sum carries
    ~1e-5 of rounding noise at unit scale, so the two differ by the REFERENCE's noise, not by MFMA's."""
    import sdflib_amd as S
    rng = np.random.default_rng(13)
    n = 2000
    vals = np.zeros((n, 8, 8), np.float32)
    vals[:, :, 0] = (0.7 + 0.01 * rng.standard_normal((n, 8))).astype(np.float32)
    g = rng.standard_normal((n, 8, 3)); g /= np.linalg.norm(g, axis=2, keepdims=True); vals[:, :, 1:4] = g
    vals[:, :, 4:] = (0.1 * rng.standard_normal((n, 8, 4))).astype(np.float32)
    ns = np.full(n, 0.05, np.float32)
    exact = S.tricubic_fit(vals, ns, gpu_ctx, fit_mode=S.FIT_EXACT)
    mfma = S.tricubic_fit(vals, ns, gpu_ctx, fit_mode=S.FIT_MFMA)
    s32 = vals.copy()
    sq = (ns * ns).astype(np.float32); cu = (sq * ns).astype(np.float32)
    s32[:, :, 1:4] *= ns[:, None, None]; s32[:, :, 4:7] *= sq[:, None, None]; s32[:, :, 7] *= cu[:, None]
    truth = s32.reshape(n, 64).astype(np.float64) @ oracle.fit_matrix().astype(np.float64).T
    err_mfma = np.abs(mfma - truth).max(); err_exact = np.abs(exact - truth).max()
    assert err_mfma < 1e-6 and err_mfma < 0.2 * err_exact
    assert np.abs(mfma - exact).max() < 5e-5


def test_mfma_build_has_identical_topology_and_close_coefficients(small):
    import sdflib_amd as S
    a = S.OctreeSdf(small["gm"], small["box"], 6, 3, 1e-3, fit_mode=S.FIT_EXACT, num_threads=2)
    b = S.OctreeSdf(small["gm"], small["box"], 6, 3, 1e-3, fit_mode=S.FIT_MFMA, num_threads=2)
    da, db = a.get_octree_data(), b.get_octree_data()
    assert da.shape == db.shape
    # walk: node words identical (bit-exact topology); leaf payloads agree to the fp32 rounding noise of the 64-term sums
    G3 = 512
    stack = list(range(G3)); maxrel = 0.0
    while stack:
        at = stack.pop(); w = int(da[at])
        assert w == int(db[at])
        base = w & 0x3FFFFFFF
        if w & 0x80000000:
            ca, cb = da[base:base + 64].view(np.float32), db[base:base + 64].view(np.float32)
            maxrel = max(maxrel, float(np.abs(ca - cb).max() / np.abs(ca).max()))
        else:
            stack.extend(range(base, base + 8))
    # the difference is the reference-ordered sum's own fp32 noise (see test_mfma_fit_is_the_correctly_rounded_product)
    assert maxrel < 1e-4
    from sdflib_amd.meshgen import random_points_in_box
    pts = random_points_in_box(small["box"], 200000, seed=77)
    np.testing.assert_allclose(a.get_distance(pts), b.get_distance(pts), rtol=0, atol=5e-5)
    assert 0 < b.info.fit_rechecks < 0.5 * b.info.num_nodes
    assert abs(a.info.min_border_value - b.info.min_border_value) < 5e-5 and a.info.value_range == b.info.value_range


@pytest.mark.parametrize("subdiv,depth,start", [(6, 7, 3)])       # (7, 8, 3) = BASELINE configs[1]: tests/test_gpu_baseline_configs.py
def test_full_size_octree_matches_oracle(oracle, gpu_ctx, subdiv, depth, start):
    """BASELINE.json configs[1] at full size (s=7: 327 680 triangles, depth 8): whole node array + 10 M queries."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
    v, f = bumpy_icosphere(subdiv)
    box = box_with_margin(v)
    gm = S.Mesh(v, f, gpu_ctx)
    gt = S.OctreeSdf(gm, box, depth, start, 1e-3, num_threads=2)
    om = oracle.Mesh(v, f)
    ot = oracle.Octree(om, box, depth, start, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    assert np.array_equal(ot.data(), gt.get_octree_data())
    assert np.float32(gt.info.value_range) == np.float32(ot.value_range) and np.float32(gt.info.min_border_value) == np.float32(ot.min_border)
    n = 10_000_000 if subdiv == 7 else 1_000_000
    pts = random_points_in_box(box, n, seed=1234)
    d0 = ot.query(pts); d1 = gt.get_distance(pts, eval_mode=S.EVAL_EXACT)
    assert np.array_equal(bits(d0), bits(d1))
    # size-independent properties: |gradient| = 1, error against the exact distance bounded by a few thresholds
    d2, g2 = gt.get_distance(pts[:200000], gradient=True, eval_mode=S.EVAL_FAST)
    np.testing.assert_allclose(np.linalg.norm(g2, axis=1), 1.0, atol=1e-4)
    np.testing.assert_allclose(d2, d1[:200000], rtol=0, atol=1e-5)
    ids = gm.nearest_triangle(pts[:200000])
    ex = gm.point_values(pts[:200000], ids)[:, 0]
    assert np.sqrt(((d1[:200000] - ex) ** 2).mean()) < 2e-3


def test_sharded_build_through_rccl_world1(small):
    """The N>1 code path (shard build -> all-gather over the 'nccl' = RCCL backend -> from_data) with a 1-rank group:
    exercises the real collectives on the GPU; the 2-rank logic is covered on CPU by tests/test_distributed_cpu.py."""
    import os
    import torch
    import torch.distributed as dist
    import sdflib_amd as S
    from sdflib_amd import distributed as sdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0)); created = True
    try:
        dev = torch.device("cuda", 0)
        tree, info = sdist.build_octree_sharded(small["gm"], small["box"], 5, 2, 1e-3, 0, 1, dev)
        full = S.OctreeSdf(small["gm"], small["box"], 5, 2, 1e-3, num_threads=2)
        assert np.array_equal(tree.get_octree_data(), full.get_octree_data())
        assert tree.info.value_range == full.info.value_range and tree.info.min_border_value == full.info.min_border_value
        assert list(tree.info.leaves_per_depth) == list(full.info.leaves_per_depth)
        # CONTINUITY: the traversal exchange (acquire / all-reduce callbacks) over RCCL
        ct, tm = sdist.build_continuity_sharded(small["gm"], small["box"], 5, 2, 1e-3, 0, 1, dev)
        c1 = S.OctreeSdf(small["gm"], small["box"], 5, 2, 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2)
        assert tm["exchange_bytes"] > 0 and np.array_equal(ct.get_octree_data(), c1.get_octree_data())
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("rule,params", [(2, (1e-3, 0.0)), (3, (1e-3, 0.05)), (0, (1e-3, 0.0))])
@pytest.mark.parametrize("algorithm", [1, 2])
def test_other_termination_rules_match_oracle(small, oracle, rule, params, algorithm):
    """SIMPSONS_RULE, BY_DISTANCE_RULE and NONE (OctreeSdfUtils.h:213-238, 87-138) through both builders."""
    import sdflib_amd as S
    depth, start = (4, 2) if rule == 0 else (5, 2)
    oc = oracle.Octree(small["om"], small["box"], depth, start, params[0], rule=rule, param1=params[1], continuity=(algorithm == 2))
    gt = S.OctreeSdf(small["gm"], small["box"], depth, start, init_algorithm=algorithm, termination_rule=rule, rule_params=params, num_threads=2)
    assert np.array_equal(oc.data(), gt.get_octree_data())
    if rule == 0:
        assert gt.info.num_leaves == 8 ** depth


def test_edge_cases_and_error_codes(small, gpu_ctx):
    import sdflib_amd as S
    t = S.OctreeSdf(small["gm"], small["box"], 4, 2, 1e-3, num_threads=2)
    # empty query batch, single query, queries far outside the box (box distance + min border value)
    assert len(t.get_distance(np.zeros((0, 3), np.float32))) == 0
    d = t.get_distance(np.array([[100.0, 0.0, 0.0]], np.float32))
    assert 90 < d[0] < 110
    with pytest.raises(S.SdfHipError):
        S.OctreeSdf(small["gm"], small["box"], 3, 5, 1e-3, num_threads=2)                       # start depth > depth
    with pytest.raises(S.SdfHipError):
        S.OctreeSdf(small["gm"], small["box"], 4, 2, 1e-3, init_algorithm=S.ALG_UNIFORM, num_threads=2)
    with pytest.raises(S.SdfHipError):
        S.Mesh(small["v"], np.array([[0, 1, 10 ** 6]], np.uint32), gpu_ctx)     # index out of range
    with pytest.raises(S.SdfHipError):
        S.ExactOctreeSdf(small["gm"], small["box"], 4, 3, 16)                   # start depth must be <= depth - 2
    bad = small["box"].copy(); bad[3] = bad[0]
    with pytest.raises(S.SdfHipError):
        S.OctreeSdf(small["gm"], bad, 4, 2, 1e-3, num_threads=2)                                # empty box
    nanbox = small["box"].copy(); nanbox[1] = np.nan
    with pytest.raises(S.SdfHipError):
        S.OctreeSdf(small["gm"], nanbox, 4, 2, 1e-3, num_threads=2)                             # NaN box
    with pytest.raises(S.SdfHipError):
        S.ExactOctreeSdf(small["gm"], nanbox, 4, 1, 16)
    vn = small["v"].copy(); vn[3, 1] = np.inf
    with pytest.raises(S.SdfHipError):
        S.Mesh(vn, small["f"], gpu_ctx)                                          # non-finite vertex
    # depth 11: beyond the 10-bit-per-axis node coordinates of this build -> SDFHIP_E_UNSUPPORTED (-5) from all three builders, with a text
    # that says why (the reference's own limit is its 30-bit word index, OctreeSdf.h:53-55); INTEGRATION.md documents it
    for build in (lambda: S.OctreeSdf(small["gm"], small["box"], 11, 2, 1e-3, num_threads=2),
                  lambda: S.OctreeSdf(small["gm"], small["box"], 11, 2, 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2),
                  lambda: S.ExactOctreeSdf(small["gm"], small["box"], 11, 3, 16)):
        with pytest.raises(S.SdfHipError, match=r"sdfhip error -5: depth 11 is above this build's limit of 10"):
            build()
    with pytest.raises(S.SdfHipError):
        S.OctreeSdf(small["gm"], small["box"], 4, 2, 1e-3, termination_rule=7, num_threads=2)   # unknown rule
    # NaN / infinite query points are answered (NaN in, NaN or the box distance out), never a fault
    q = np.array([[np.nan, 0, 0], [np.inf, 0, 0], [0, -np.inf, 0], [1e30, 1e30, 1e30]], np.float32)
    out = t.get_distance(q, gradient=True)
    assert len(out[0]) == 4
    e = S.ExactOctreeSdf(small["gm"], small["box"], 4, 1, 16)
    assert len(e.get_distance(q)) == 4


def test_seam_welding_matches_oracle(oracle, gpu_ctx):
    """sdfhip_mesh_create_ex with the loader's box: welded TriangleData equals the oracle's (reference
    TriangleUtils.cpp:292-420), and an octree built on the welded soup equals the oracle's."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, triangle_soup, box_with_margin
    v, f = bumpy_icosphere(3)
    sv, sf = triangle_soup(v, f)
    bbox = np.concatenate([sv.min(axis=0), sv.max(axis=0)])
    om, gm = oracle.Mesh(sv, sf, bbox), S.Mesh(sv, sf, gpu_ctx, bbox=bbox)
    st = gm.edge_stats()
    assert st["unmatched_edges"] == 3 * len(sf) and st["welded_half_edges"] == 3 * len(sf)
    a, b = om.triangle_data(), gm.triangle_data()
    assert np.array_equal(bits(a), bits(b))
    raw = S.Mesh(sv, sf, gpu_ctx)
    assert raw.edge_stats() == {"unmatched_edges": 3 * len(sf), "welded_half_edges": 0}
    assert np.array_equal(bits(raw.triangle_data()), bits(oracle.Mesh(sv, sf).triangle_data()))
    box = box_with_margin(sv)
    ot = oracle.Octree(om, box, 5, 2, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    gt = S.OctreeSdf(gm, box, 5, 2, 1e-3, num_threads=2)
    go, oo = gt.get_octree_data(), ot.data()
    assert go.shape == oo.shape
    assert np.array_equal(go, oo)
    # ExactOctreeSdf takes its TriangleData from the same calculateMeshTriangleData(mesh) call (ExactOctreeSdf.cpp:25): the welded
    # pseudonormals decide the sign of its distances too (found by tools/gpu_fuzz.py: the oracle used to skip the welding here)
    from sdflib_amd.meshgen import random_points_in_box
    pts = random_points_in_box(box, 30000, seed=21)
    oe, ge = oracle.Exact(om, box, 4, 1, 16), S.ExactOctreeSdf(gm, box, 4, 1, 16)
    assert np.array_equal(bits(oe.query(pts)), bits(ge.get_distance(pts)))


@pytest.mark.parametrize("ntri", [1, 2, 3])
def test_tiny_open_meshes(oracle, gpu_ctx, ntri):
    """One, two and three triangles (open surfaces: every edge single-owner): the BVH degenerates to a leaf root / one inner
    node; nearest ids, samples and the built array still equal the oracle's."""
    import sdflib_amd as S
    v = np.array([[0, 0, 0], [1, 0, 0.1], [0.2, 1, 0], [1.1, 1.2, 0.3], [-0.5, 0.8, 0.6]], np.float32)
    f = np.array([[0, 1, 2], [1, 3, 2], [0, 2, 4]], np.uint32)[:ntri]
    box = np.array([-1, -1, -1, 2, 2, 2], np.float32)
    om, gm = oracle.Mesh(v, f), S.Mesh(v, f, gpu_ctx)
    rng = np.random.default_rng(3)
    pts = (rng.random((5000, 3), dtype=np.float32) * 3 - 1).astype(np.float32)
    assert np.array_equal(om.nearest(pts), gm.nearest_triangle(pts))
    for alg, cont in ((S.ALG_NO_CONTINUITY, False), (S.ALG_CONTINUITY, True)):
        ot = oracle.Octree(om, box, 4, 1, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_GLOBAL_DFS, continuity=cont)
        gt = S.OctreeSdf(gm, box, 4, 1, 1e-3, init_algorithm=alg, num_threads=1)
        assert np.array_equal(ot.data(), gt.get_octree_data())
        assert np.array_equal(bits(ot.query(pts)), bits(gt.get_distance(pts)))


def test_one_million_triangle_build_properties(gpu_ctx):
    """BASELINE configs[3] (1.31 M triangles, depth 8, start 3) is too large for the CPU oracle inside a test run; checked through
    size-independent properties instead: shards over disjoint cell ranges emit exactly the words of the single build
    (idempotence of the decomposition), the tree's distances agree with brute-force-exact ones within the threshold's error
    budget, and grid and point queries coincide."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
    v, f = bumpy_icosphere(8)
    assert len(f) == 1310720
    box = box_with_margin(v)
    mesh = S.Mesh(v, f, gpu_ctx)
    full = S.OctreeSdf(mesh, box, 8, 3, 1e-3, num_threads=2)
    words = full.get_octree_data()
    info = full.info
    assert info.num_traversals < info.num_samples and info.num_leaves > 250000
    # three shards with ragged cell ranges, emitted at the offsets a 3-rank run would compute
    cuts = (0, 100, 317, 512)
    shards = [S.OctreeShard(mesh, box, 8, 3, 1e-3, cells=(a, b)) for a, b in zip(cuts, cuts[1:])]
    # a shard traverses the BVH for ITS cells' samples only (until round 5 every shard sampled the two speculative levels complete):
    # the shards together do little more than the single build (the a-priori levels down to the start depth are every shard's)
    assert sum(int(sh.info.num_traversals) for sh in shards) < 1.15 * int(info.num_traversals), ([int(sh.info.num_traversals) for sh in shards], int(info.num_traversals))
    offset = 512
    for sh, (a, b) in zip(shards, zip(cuts, cuts[1:])):
        n = int(sh.info.body_words)
        grid = np.zeros(b - a, np.uint32); body = np.zeros(max(n, 1), np.uint32)
        sh.emit(offset, grid, body)
        assert np.array_equal(grid, words[a:b]) and np.array_equal(body[:n], words[offset:offset + n])
        offset += n
        sh.close()
    assert offset == len(words)
    pts = random_points_in_box(full.get_grid_bounding_box(), 200000, seed=5)
    d = full.get_distance(pts)
    ids = mesh.nearest_triangle(pts)
    r = np.linalg.norm(pts, axis=1)
    # the mesh is a bumpy unit sphere (|r - 1| <= 0.1): far from it the sign is known, and |d| is bounded by the distance to the shell
    assert np.all(d[r > 1.15] > 0) and np.all(d[r < 0.85] < 0)
    assert np.all(np.abs(d) <= np.abs(r - 1.0) + 0.11)
    assert ids.max() < len(f)


@pytest.mark.parametrize("scale,offset", [(1e-3, (0.0, 0.0, 0.0)), (1.0, (1000.0, -2000.0, 500.0)), (250.0, (-5000.0, 0.0, 12345.0)), (1e3, (0.0, 0.0, 0.0))])
def test_nearest_triangle_far_from_the_origin_and_at_other_scales(oracle, gpu_ctx, scale, offset):
    """The fp32 sphere bracket of the BVH search scales its slack with the mesh's coordinate magnitude; meshes that are tiny,
    huge or far from the origin (fp32 spacing up to 1e-3 there) must still give the oracle's ids and the oracle's tree."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere
    v, f = bumpy_icosphere(3)
    v = (v * np.float32(scale) + np.float32(offset)).astype(np.float32)
    lo, hi = v.min(axis=0), v.max(axis=0)
    m = np.float32(0.2) * (hi - lo).max()
    box = np.concatenate([lo - m, hi + m]).astype(np.float32)
    om, gm = oracle.Mesh(v, f), S.Mesh(v, f, gpu_ctx)
    rng = np.random.default_rng(11)
    pts = (lo - m + rng.random((30000, 3), dtype=np.float32) * (hi - lo + 2 * m)).astype(np.float32)
    near = (v[rng.integers(0, len(v), 10000)] * (1 + rng.normal(0, 1e-4, (10000, 1)))).astype(np.float32)     # on top of the surface: tie-heavy
    pts = np.concatenate([pts, near])
    assert np.array_equal(om.nearest(pts), gm.nearest_triangle(pts))
    ot = oracle.Octree(om, box, 5, 2, 1e-3 * scale, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    gt = S.OctreeSdf(gm, box, 5, 2, 1e-3 * scale, num_threads=2)
    assert np.array_equal(ot.data(), gt.get_octree_data())


def test_degenerate_triangles_behave_like_the_reference(oracle, gpu_ctx):
    """Zero-area triangles (repeated vertices, a collinear sliver): the reference's degenerate branch is disabled
    (`if(false && ...)`, TriangleUtils.cpp:45), so their frames are NaN, their neighbours' pseudonormals inherit NaNs, and the
    fp64 search never adopts them (NaN distances fail every '<').  Same NaN pattern, same ids, same tree."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import icosphere, box_with_margin
    v, f = icosphere(2)
    mid = ((v[3] + v[7]) * np.float32(0.5)).astype(np.float32)
    v = np.concatenate([v, mid[None]])
    f = np.concatenate([f, np.array([[0, 0, 5], [1, 2, 2], [3, 7, len(v) - 1]], np.uint32)])
    box = box_with_margin(v)
    om, gm = oracle.Mesh(v, f), S.Mesh(v, f, gpu_ctx)
    a, b = om.triangle_data(), gm.triangle_data()
    assert np.isnan(a).any() and np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(a[:, :28], b[:, :28], equal_nan=True)
    rng = np.random.default_rng(0)
    pts = ((rng.random((50000, 3), dtype=np.float32) * 2 - 1) * 1.3).astype(np.float32)
    ids = gm.nearest_triangle(pts)
    assert np.array_equal(om.nearest(pts), ids) and ids.max() < len(f) - 3
    ot = oracle.Octree(om, box, 5, 2, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    gt = S.OctreeSdf(gm, box, 5, 2, 1e-3, num_threads=2)
    assert np.array_equal(ot.data(), gt.get_octree_data())
    assert np.array_equal(ot.query(pts), gt.get_distance(pts), equal_nan=True)


def _assorted_meshes():
    from sdflib_amd.meshgen import cube_mesh, icosphere
    v, f = cube_mesh()
    yield "cube", v, f
    a, fa = icosphere(2)
    two = np.concatenate([a * np.float32(0.5) + np.float32([-0.6, 0, 0]), a * np.float32(0.3) + np.float32([0.7, 0.1, 0])]).astype(np.float32)
    yield "two spheres", two, np.concatenate([fa, fa + len(a)]).astype(np.uint32)
    keep = fa[(a[fa].mean(axis=1)[:, 2] > -0.1)]                       # open hemisphere: boundary edges have one owner
    yield "open hemisphere", a, keep.astype(np.uint32)


@pytest.mark.parametrize("name,v,f", list(_assorted_meshes()), ids=lambda x: x if isinstance(x, str) else None)
def test_assorted_meshes_both_builders_and_exact(oracle, gpu_ctx, name, v, f):
    """Flat faces and symmetric sample points (exact distance ties everywhere: the BVH's visiting order decides), disjoint
    components, and an open surface, through the three structures."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import box_with_margin
    box = box_with_margin(v)
    om, gm = oracle.Mesh(v, f), S.Mesh(v, f, gpu_ctx)
    g = np.linspace(box[0], box[3], 33, dtype=np.float32)                # symmetric lattice incl. the mesh's symmetry planes
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float32)
    assert np.array_equal(om.nearest(pts), gm.nearest_triangle(pts))
    for cont in (False, True):
        ot = oracle.Octree(om, box, 5, 2, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_GLOBAL_DFS, continuity=cont)
        gt = S.OctreeSdf(gm, box, 5, 2, 1e-3, init_algorithm=S.ALG_CONTINUITY if cont else S.ALG_NO_CONTINUITY, num_threads=1)
        assert np.array_equal(ot.data(), gt.get_octree_data()), f"{name}: continuity={cont}"
        assert np.array_equal(bits(ot.query(pts)), bits(gt.get_distance(pts)))
    oe = oracle.Exact(om, box, 4, 1, 4)
    ge = S.ExactOctreeSdf(gm, box, 4, 1, 4)
    on, oh, osets, omasks = oe.data(); gn, gh, gsets, gmasks = ge.download()
    assert np.array_equal(on[:, 0], gn[:, 0]) and np.array_equal(osets, gsets) and np.array_equal(omasks, gmasks)
    d, t = ge.get_distance(pts, triangle=True)
    do, to = oe.query(pts, tri=True)
    assert np.array_equal(bits(do), bits(d)) and np.array_equal(to, t.astype(np.uint32))


def test_deep_tree_depth_9_matches_oracle(oracle, gpu_ctx):
    """Depth 9 with a tight threshold: 428 M words (1.7 GB), 6.6 M leaves, 10-bit lattice coordinates in the sampler keys — the
    array still equals the oracle's word for word (depth 10 / 517 M words was checked the same way by hand)."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
    v, f = bumpy_icosphere(5)
    box = box_with_margin(v)
    gt = S.OctreeSdf(S.Mesh(v, f, gpu_ctx), box, 9, 3, 2e-4, num_threads=2)
    assert gt.info.num_words > 400_000_000 and gt.info.num_traversals < gt.info.num_samples
    ot = oracle.Octree(oracle.Mesh(v, f), box, 9, 3, 2e-4, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    assert np.array_equal(ot.data(), gt.get_octree_data())


def test_concurrent_host_threads_query_one_tree(small):
    """queries are pure reads and may come from several host threads at once (the reference's OctreeSdf is
    re-entrant; its ExactOctreeSdf is not — ours is).  Four threads query the same OctreeSdf and ExactOctreeSdf with host
    buffers of different sizes, repeatedly; every answer must equal the single-threaded one."""
    import threading
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    tree = S.OctreeSdf(small["gm"], small["box"], 6, 3, 1e-3, num_threads=2)
    exact = S.ExactOctreeSdf(small["gm"], small["box"], 5, 2, 16)
    sets = [random_points_in_box(small["box"], n, seed=40 + i) for i, n in enumerate((1, 777, 30000, 200000))]
    want = [(tree.get_distance(p, gradient=True), exact.get_distance(p)) for p in sets]
    errors = []

    def work(i):
        try:
            for _ in range(12):
                d, g = tree.get_distance(sets[i], gradient=True)
                e = exact.get_distance(sets[i])
                if not (np.array_equal(bits(d), bits(want[i][0][0])) and np.array_equal(bits(g), bits(want[i][0][1])) and np.array_equal(bits(e), bits(want[i][1]))):
                    errors.append(f"thread {i}: result differs from the single-threaded query")
                    return
        except Exception as ex:      # noqa: BLE001
            errors.append(f"thread {i}: {ex!r}")

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(sets))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_large_query_arrays_are_answered_in_chunks():
    """Point queries run in chunks (32-bit grid dimensions / sort sizes; host arrays staged chunk by chunk).  The chunk loop is
    exercised on a small input by forcing the chunk size down (read once per process, hence the subprocess)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
import sdflib_amd as S
from oracle import pyoracle as O
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
v, f = bumpy_icosphere(2); box = box_with_margin(v)
m = S.Mesh(v, f)
t = S.OctreeSdf(m, box, 5, 2, 1e-3, num_threads=2); e = S.ExactOctreeSdf(m, box, 4, 1, 8)
om = O.Mesh(v, f); ot = O.Octree(om, box, 5, 2, 1e-3); oe = O.Exact(om, box, 4, 1, 8)
pts = random_points_in_box(box, 40001, seed=3); pts[::97] *= 3.0
d0, g0 = ot.query(pts, grad=True); e0 = oe.query(pts)
b = lambda a: np.ascontiguousarray(a).view(np.uint32)
d, g = t.get_distance(pts, gradient=True)                        # host arrays
assert np.array_equal(b(d), b(d0)) and np.array_equal(b(g), b(g0))
assert np.array_equal(b(e.get_distance(pts)), b(e0))
tp = torch.from_numpy(pts).cuda()                                # device arrays; the default context has its OWN stream: the calls are fenced
assert np.array_equal(b(t.get_distance(tp).cpu().numpy()), b(d0))
assert np.array_equal(b(e.get_distance(tp).cpu().numpy()), b(e0))
ctx2 = S.Context(0, use_torch_stream=True)                       # engine on torch's stream: no fences, plain stream order
m2 = S.Mesh(v, f, ctx2); e2 = S.ExactOctreeSdf(m2, box, 4, 1, 8)
assert np.array_equal(b(e2.get_distance(tp).cpu().numpy()), b(e0))
print("chunks ok")
''' % ROOT
    env = dict(os.environ, SDFHIP_QUERY_CHUNK="16411")          # 3 chunks, the last one ragged; >= 16384 so the sorted exact path runs too
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "chunks ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_concurrent_builds_on_one_context_are_serialised(small, oracle):
    """Three host threads build an OctreeSdf, a CONTINUITY OctreeSdf and an ExactOctreeSdf on the SAME context while a fourth
    queries an existing tree: builds take the context's build lock one at a time, queries do not wait for it; every result equals
    the sequential one."""
    import threading
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    gm, box = small["gm"], small["box"]
    pts = random_points_in_box(box, 50000, seed=2)
    ref = dict(a=S.OctreeSdf(gm, box, 6, 3, 1e-3, num_threads=2).get_octree_data(), b=S.OctreeSdf(gm, box, 5, 2, 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2).get_octree_data(),
               c=S.ExactOctreeSdf(gm, box, 5, 2, 16).download())
    live = S.OctreeSdf(gm, box, 5, 2, 1e-3, num_threads=2); want = live.get_distance(pts)
    out, errors = {}, []

    def run(name, fn):
        try:
            for _ in range(3):
                out[name] = fn()
        except Exception as ex:      # noqa: BLE001
            errors.append(f"{name}: {ex!r}")

    jobs = [("a", lambda: S.OctreeSdf(gm, box, 6, 3, 1e-3, num_threads=2).get_octree_data()),
            ("b", lambda: S.OctreeSdf(gm, box, 5, 2, 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2).get_octree_data()),
            ("c", lambda: S.ExactOctreeSdf(gm, box, 5, 2, 16).download()),
            ("q", lambda: [live.get_distance(pts) for _ in range(10)][-1])]
    th = [threading.Thread(target=run, args=j) for j in jobs]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors
    assert np.array_equal(out["a"], ref["a"]) and np.array_equal(out["b"], ref["b"])
    assert all(np.array_equal(x, y) for x, y in zip(out["c"], ref["c"]))
    assert np.array_equal(bits(out["q"]), bits(want))


def test_bvh_export_import(small, oracle, gpu_ctx):
    """sdfhip_mesh_bvh_export / _import: a tree planned on one mesh object (another rank, in the multi-GPU flow) and installed in a
    second one gives the same nearest triangles and the same octree as planning it there."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    sph, kids = small["gm"].bvh_arrays()
    assert sph.shape == (8 * (len(small["f"]) - 1),) and kids.shape == (2 * (len(small["f"]) - 1),)
    other = S.Mesh(small["v"], small["f"], gpu_ctx)
    other.set_bvh(sph, kids)
    assert other.build_bvh() == 0.0                                  # nothing left to plan
    pts = random_points_in_box(small["box"], 50000, seed=12)
    assert np.array_equal(other.nearest_triangle(pts), small["om"].nearest(pts))
    assert np.array_equal(S.OctreeSdf(other, small["box"], 5, 2, 1e-3, num_threads=2).get_octree_data(), S.OctreeSdf(small["gm"], small["box"], 5, 2, 1e-3, num_threads=2).get_octree_data())
    a, b = other.bvh_arrays()
    assert np.array_equal(a.view(np.uint64), sph.view(np.uint64)) and np.array_equal(b, kids)
    # one-triangle mesh: a single dummy node
    tiny = S.Mesh(small["v"][:3], np.array([[0, 1, 2]], np.uint32), gpu_ctx)
    s1, k1 = tiny.bvh_arrays()
    assert s1.shape == (8,) and k1.shape == (2,)


def test_scalar_and_small_host_batches_are_answered_like_the_device(small, oracle):
    """The host-side entry of the library (a handful of points with host pointers: the scalar getDistance of the reference's API) runs
    the same code on host copies of the arrays: results must equal the device path's and the oracle's bit for bit, both eval modes,
    with gradients, inside and outside the box."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    gt = S.OctreeSdf(small["gm"], small["box"], 6, 3, 1e-3, num_threads=2)
    ot = oracle.Octree(small["om"], small["box"], 6, 3, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    pts = random_points_in_box(small["box"], 4000, seed=17)
    pts[::7] *= 2.5
    d0, g0 = ot.query(pts, grad=True)
    for mode in (S.EVAL_EXACT, S.EVAL_FAST):
        dd, gd = gt.get_distance(pts, gradient=True, eval_mode=mode)                    # device (4000 points)
        for k in range(0, 4000, 23):
            m = 1 + (k % 31)
            ds, gs = gt.get_distance(pts[k:k + m], gradient=True, eval_mode=mode)       # host (<= 32 points)
            assert np.array_equal(bits(ds), bits(dd[k:k + m])) and np.array_equal(bits(gs), bits(gd[k:k + m]))
            assert np.array_equal(bits(gt.get_distance(pts[k:k + 1], eval_mode=mode)), bits(dd[k:k + 1]))
        if mode == S.EVAL_EXACT:
            assert np.array_equal(bits(dd), bits(d0)) and np.array_equal(bits(gd), bits(g0))


def test_cooperative_fetch_kernel_on_ragged_batches(small, oracle):
    """k_octree_query_coop shares every coefficient fetch among 16 lanes and passes the rows through LDS: batches whose size is not a
    multiple of 16 / 64 / 256, batches that are entirely outside the grid, and batches where only some lanes of a wave are inside must
    give the oracle's bits in every lane (device-resident points: no host shortcut)."""
    import torch
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    gt = S.OctreeSdf(small["gm"], small["box"], 6, 3, 1e-3, num_threads=2)
    ot = oracle.Octree(small["om"], small["box"], 6, 3, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES)
    base = random_points_in_box(small["box"], 3000, seed=31)
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1023, 3000):
        for variant in range(3):
            pts = base[:n].copy()
            if variant == 1: pts *= 3.0                                        # (almost) everything outside the grid
            if variant == 2: pts[::3] = pts[::3] * 2.2 + 0.1                   # inside and outside lanes interleaved
            d0, g0 = ot.query(pts, grad=True)
            tp = torch.from_numpy(pts).cuda()
            for mode in (S.EVAL_EXACT, S.EVAL_FAST):
                d1, g1 = gt.get_distance(tp, gradient=True, eval_mode=mode)
                d2 = gt.get_distance(tp, gradient=False, eval_mode=mode)
                d1, g1, d2 = d1.cpu().numpy(), g1.cpu().numpy(), d2.cpu().numpy()
                assert np.array_equal(bits(d1), bits(d2)), (n, variant, mode)
                if mode == S.EVAL_EXACT:
                    assert np.array_equal(bits(d1), bits(d0)), (n, variant)
                    same = (bits(g1) == bits(g0)) | (np.isnan(g1) & np.isnan(g0))
                    assert same.all(), (n, variant)
                else:
                    np.testing.assert_allclose(d1, d0, rtol=0, atol=1e-5)


def _host_pointer_large_case(gm, box):
    """Host arrays of 3 M points: same bits as the device-resident path; misaligned views of larger arrays."""
    import torch
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    gt = S.OctreeSdf(gm, box, 6, 3, 1e-3, num_threads=2)
    n = 3_000_001
    big = np.empty((n + 5, 3), dtype=np.float32)
    big[:] = random_points_in_box(box, n + 5, seed=23)
    pts = big[3:3 + n]                                   # 36 bytes into the allocation: not page aligned
    want, wantg = gt.get_distance(torch.from_numpy(pts.copy()).cuda(), gradient=True)
    outbuf = np.empty(n + 3, dtype=np.float32); gbuf = np.empty((n + 1, 3), dtype=np.float32)
    d, g = gt.get_distance(pts, gradient=True, out=outbuf[1:1 + n], out_grad=gbuf[1:1 + n])
    assert np.array_equal(bits(d), bits(want.cpu().numpy())) and np.array_equal(bits(g), bits(wantg.cpu().numpy()))
    d2 = gt.get_distance(pts)
    assert np.array_equal(bits(d2), bits(d))


def _host_pointer_huge_case(gm, box):
    """23 M points = 276 MB of coordinates, above the 256 MB limit of the context's staging buffers (the call does not hold the staging
    lock and allocates its own buffers), and 3 M points (staged): same bits as the device-resident path."""
    import torch
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    gt = S.OctreeSdf(gm, box, 6, 3, 1e-3, num_threads=2)
    for n in (23_000_000, 3_000_000):
        pts = random_points_in_box(box, n, seed=29)
        want = gt.get_distance(torch.from_numpy(pts).cuda()).cpu().numpy()
        d = gt.get_distance(pts)
        assert np.array_equal(bits(d), bits(want)), n


def test_large_host_pointer_queries_are_identical(small):
    """The host-pointer path (plain copies through the context's buffers) on 3 M and 23 M points, misaligned views included."""
    _host_pointer_large_case(small["gm"], small["box"])
    _host_pointer_huge_case(small["gm"], small["box"])


def test_imported_bvh_of_another_shape_is_refused(gpu_ctx):
    """The two-phase nearest search navigates the planner's tree by arithmetic (pre-order numbering, midpoint splits); an import that
    is not of that shape would silently give wrong ids (round-2 advisor finding), so it is refused; the planner's own export is accepted."""
    import sdflib_amd as S
    from sdflib_amd._lib import SdfHipError
    from sdflib_amd.meshgen import bumpy_icosphere, random_points_in_box, box_with_margin
    v, f = bumpy_icosphere(3)
    a = S.Mesh(v, f, gpu_ctx); a.build_bvh()
    sph, kids = a.bvh_arrays()
    b = S.Mesh(v, f, gpu_ctx)
    b.set_bvh(sph, kids)                                    # the planner's tree: accepted
    pts = random_points_in_box(box_with_margin(v), 5000, seed=2)
    assert np.array_equal(a.nearest_triangle(pts), b.nearest_triangle(pts))
    kids2 = kids.copy().reshape(-1, 2)
    kids2[0] = kids2[0][::-1]                               # root's children swapped: still a valid binary tree, not the planner's numbering
    c = S.Mesh(v, f, gpu_ctx)
    with pytest.raises(SdfHipError):
        c.set_bvh(sph, kids2.ravel())
    kids3 = kids.copy(); leaf = np.nonzero(kids3 < 0)[0]
    kids3[leaf[0]] = kids3[leaf[1]]                         # one triangle twice, another never
    with pytest.raises(SdfHipError):
        c.set_bvh(sph, kids3)


@pytest.mark.parametrize("subtree,build", [("1", "host"), ("4096", "device"), pytest.param("512", "device", marks=pytest.mark.soak),
                                           pytest.param("512", "host", marks=pytest.mark.soak), pytest.param("8192", "host", marks=pytest.mark.soak),
                                           pytest.param("8192", "device", marks=pytest.mark.soak)])
def test_hybrid_bvh_plan_equals_the_oracles_tree(oracle, gpu_ctx, subtree, build):
    """The tree as the DEVICE holds it after sdfhip_mesh_build_bvh — top planned on the host, every range of at most 4096 triangles built by
    k_bvh_subtrees (ordered fp64 centre sums, libstdc++'s introsort restated per lane) — walked together with the oracle's from the root:
    all 64 bits of every child sphere, every leaf's triangle.  Meshes chosen for what decides the tree: tied sort keys (symmetric meshes,
    shared first vertices, duplicated triangles), degenerate triangles, scales, sizes around the hand-over threshold."""
    import subprocess, sys
    # the switch is read once per process: the hybrid plans run in a child
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_octree as t; t._hybrid_cases_check()" % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    # build == "device": the top of the tree is sorted on the device as well (SDFHIP_BVH_BUILD=device: introsort as rounds over global memory,
    # centre sums in parallel, verified chunk by chunk, the serial chain for the nodes whose verification fails: k_csum_*) — no host plan at
    # all unless a long range exhausts introsort's depth limit
    env = dict(os.environ, SDFHIP_BVH_DEVICE_SUBTREES=subtree, SDFHIP_BVH_BUILD="device" if build.startswith("device") else build, SDFHIP_TIMING="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "hybrid cases ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    if build.startswith("device"):
        # both ways of summing a node's centre were exercised: verified in parallel (the rule) and redone by the serial chain (the case whose
        # coordinates span seven decades: its additions round)
        lines = [l for l in r.stderr.splitlines() if "centre sums of" in l and "in parallel" in l]
        par = sum(int(l.split("centre sums of ")[1].split()[0]) for l in lines); ser = sum(int(l.split("), ")[1].split()[0]) for l in lines)
        assert par >= 100 and 1 <= ser < par // 4, (par, ser)
        built = r.stderr.count("bvh: built on the device")
        assert built >= len(_hybrid_cases()) - 2, f"only {built} trees were built on the device:\n" + "\n".join(l for l in r.stderr.splitlines() if "gave up" in l)[-2000:]


def _hybrid_cases():
    from sdflib_amd import meshgen
    rng = np.random.default_rng(11)
    cases = []
    for s_ in (5, 6):
        cases.append(meshgen.icosphere(s_))                                   # symmetric: tied keys on every axis, 20 480 / 81 920 triangles
    v, f = meshgen.bumpy_icosphere(6)
    cases.append((v, f))
    cases.append((v * np.array([1e-3, 7.0, 250.0], np.float32) + np.array([1e3, -2.0, 0.5], np.float32), f))
    cases.append((v, np.concatenate([f, f[::3], f[::5]])))                    # duplicated triangles: fully tied pairs
    cases.append((v, f[rng.permutation(len(f))]))
    cases.append((v, f[:, [1, 2, 0]]))
    cases.append((v, np.concatenate([f, np.array([[0, 0, 1], [2, 2, 2]], np.uint32)])))
    v5, f5 = meshgen.bumpy_icosphere(5)
    for n in (4095, 4096, 4097, 8191, 8193, 12289):                           # around the threshold: the root handed over whole, or split just above it
        cases.append((v5, f5[:n]))
    cases.append(meshgen.torus_knot(nu=512, nv=80))
    # coordinates over seven decades (a soup as far as geometry goes: the tree only sees vertices): fp64 sums of such fp32 values ROUND, the
    # verification of the parallel centre sums fails for the long nodes and the serial chain redoes them
    wide = (v * np.float32(10.0) ** rng.uniform(-4.0, 3.0, size=(len(v), 1)).astype(np.float32)).astype(np.float32)
    cases.append((wide, f))
    cases.append(meshgen.bumpy_icosphere(7))
    cases.append(meshgen.bumpy_icosphere(8))                                  # 1.31 M triangles: ranges that exhaust introsort's depth limit (heap sort)
    return cases


def _hybrid_cases_check():
    import sdflib_amd as S
    from test_planner_cpu import same_tree
    assert os.environ.get("SDFHIP_BVH_DEVICE_SUBTREES")
    ctx = S.default_context(0)
    for v, f in _hybrid_cases():
        m = S.Mesh(v, f, ctx)
        m.build_bvh()
        assert same_tree(v, f, arrays=m.bvh_arrays())
        m.close()
    print("hybrid cases ok")


@pytest.mark.parametrize("seed,subdiv", [(0, 3), (1, 4), (2, 5)])
def test_scan_like_mesh_with_slivers_t_junctions_self_intersection(oracle, gpu_ctx, seed, subdiv):
    """meshgen.scan_like_mesh: slivers (aspect 1e3..1e5), a valence-64 vertex, T-junctions, a self-intersecting component, unwelded and
    jittered seams, a flipped patch, zero-area triangles, an isolated far triangle — with and without the loader's bounding box (seam
    welding).  TriangleData, nearest ids, both OctreeSdf builders' arrays, the ExactOctreeSdf arrays and queries against the oracle."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import scan_like_mesh, box_with_margin, random_points_in_box
    v, f = scan_like_mesh(seed, subdiv)
    box = box_with_margin(v, margin=0.1)
    pts = random_points_in_box(box, 40000, seed=seed + 5)
    tri = v[f[::3]]
    near = (tri.mean(1) + np.random.default_rng(seed).normal(0, 1e-3, (len(tri), 3))).astype(np.float32)       # points hugging the surface (slivers, seams)
    pts = np.ascontiguousarray(np.concatenate([pts, near, v[::5]]), np.float32)
    for bbox in (None, np.concatenate([v.min(0), v.max(0)]).astype(np.float32)):
        om = oracle.Mesh(v, f, bbox=bbox) if bbox is not None else oracle.Mesh(v, f)
        gm = S.Mesh(v, f, gpu_ctx, bbox=bbox) if bbox is not None else S.Mesh(v, f, gpu_ctx)
        a, b = om.triangle_data(), gm.triangle_data()
        assert np.array_equal(bits(a), bits(b)) or np.array_equal(a, b, equal_nan=True), "TriangleData"
        assert np.array_equal(om.nearest(pts), gm.nearest_triangle(pts)), "nearest ids"
        depth = 6 if subdiv < 5 else 7
        for cont in (False, True):
            ot = oracle.Octree(om, box, depth, 2, 1e-3, vertex_cache=False, layout=oracle.LAYOUT_SUBTREES, continuity=cont)
            gt = S.OctreeSdf(gm, box, depth, 2, 1e-3, init_algorithm=S.ALG_CONTINUITY if cont else S.ALG_NO_CONTINUITY, num_threads=2)
            assert np.array_equal(ot.data(), gt.get_octree_data()), f"octree array (continuity={cont}, welded={bbox is not None})"
            d0, g0 = ot.query(pts, grad=True); d1, g1 = gt.get_distance(pts, gradient=True)
            assert np.array_equal(bits(d0), bits(d1)) and (np.array_equal(bits(g0), bits(g1)) or np.array_equal(g0, g1, equal_nan=True))
        oe = oracle.Exact(om, box, 6, 2, 16, threads=0); ge = S.ExactOctreeSdf(gm, box, 6, 2, 16)
        for name, x, y in zip(("nodes", "has", "sets", "masks"), oe.data(), ge.download()):
            if name == "has": continue
            if name == "nodes": x, y = x[:, 0], y[:, 0]
            assert x.shape == y.shape and np.array_equal(x, y), f"exact {name}"
        e0, t0 = oe.query(pts, tri=True); e1, t1 = ge.get_distance(pts, triangle=True)
        assert np.array_equal(bits(e0), bits(e1)) and np.array_equal(t0, t1.astype(np.uint32)), "exact queries"
        eg0 = oe.query(pts, grad=True); eg1 = ge.get_distance(pts, gradient=True)
        assert np.array_equal(bits(eg0[0]), bits(eg1[0])) and (np.array_equal(bits(eg0[1]), bits(eg1[1])) or np.array_equal(eg0[1], eg1[1], equal_nan=True)), "exact gradients"


def test_large_host_batches_are_overlapped_and_bit_identical(small, gpu_ctx):
    """Host-pointer batches of 2^21 points and more are cut into pieces whose upload + evaluation overlaps the previous piece's download
    (two host threads, two streams, plain pageable copies): values and gradients equal the device-pointer path's, for sizes around the
    piece boundaries, repeatedly (the second stream and the events are created and destroyed per call)."""
    import torch
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    t = S.OctreeSdf(small["gm"], small["box"], 6, 2, 1e-3, num_threads=2)
    for n in (2_097_152, 2_097_153, 3_145_728 + 17, 5_000_001):
        pts = random_points_in_box(small["box"], n, seed=n % 1000)
        pts[::7919] *= 3.0                                  # some points outside the box
        dt, gt = t.get_distance(torch.from_numpy(pts).cuda(), gradient=True)
        for rep in range(2):
            d, g = t.get_distance(pts, gradient=True)
            assert np.array_equal(bits(d), bits(dt.cpu().numpy())), n
            assert np.array_equal(bits(g), bits(gt.cpu().numpy())), n
        d1 = t.get_distance(pts)
        assert np.array_equal(bits(d1), bits(dt.cpu().numpy())), n
    # two host threads in the overlapped path at once on one context (queries are documented as concurrent; the side streams are shared)
    import threading
    a = random_points_in_box(small["box"], 2_500_000, seed=1); b = random_points_in_box(small["box"], 3_000_000, seed=2)
    want = [t.get_distance(torch.from_numpy(x).cuda()).cpu().numpy() for x in (a, b)]
    got = [None, None]
    def work(i, x):
        for _ in range(3):
            got[i] = t.get_distance(x)
    th = [threading.Thread(target=work, args=(i, x)) for i, x in enumerate((a, b))]
    for x in th: x.start()
    for x in th: x.join()
    assert np.array_equal(bits(got[0]), bits(want[0])) and np.array_equal(bits(got[1]), bits(want[1]))
