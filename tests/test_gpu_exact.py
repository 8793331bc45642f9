"""GPU parity of the ExactOctreeSdf path (through the C ABI) vs the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(oracle, gpu_ctx):
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
    v, f = bumpy_icosphere(3)
    return dict(v=v, f=f, box=box_with_margin(v), om=oracle.Mesh(v, f), gm=S.Mesh(v, f, gpu_ctx))


def test_is_near_minimize_bit_exact_decisions(oracle, gpu_ctx):
    import sdflib_amd as S
    from sdflib_amd._lib import lib, check
    rng = np.random.default_rng(21)
    n = 20000
    half = (0.05 + rng.random(n) * 0.5).astype(np.float32)
    radius = (rng.random((n, 8)) * 0.3).astype(np.float32)
    tri = ((rng.random((n, 3, 3)) * 2 - 1) * 1.2).astype(np.float32)
    thr = (rng.random(n) * 0.4).astype(np.float32)
    out = np.zeros(n, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().sdfhip_is_near_minimize(gpu_ctx.h, p(half), p(radius), p(tri), p(thr), n, p(out)))
    ref = np.array([oracle.is_near_minimize(half[i], radius[i], tri[i], thr[i])[0] for i in range(n)], dtype=np.uint8)
    assert np.array_equal(out, ref)
    assert 0.05 < ref.mean() < 0.95


@pytest.mark.parametrize("depth,start,min_tri", [(5, 1, 16), (6, 3, 32), (5, 2, 64), (4, 1, 8)])
def test_exact_build_matches_oracle_arrays(small, oracle, depth, start, min_tri):
    import sdflib_amd as S
    ex = oracle.Exact(small["om"], small["box"], depth, start, min_tri, vertex_cache=False)
    gx = S.ExactOctreeSdf(small["gm"], small["box"], depth, start, min_tri)
    i = gx.info
    assert (i.num_nodes, i.num_set_words, i.num_mask_bytes) == (ex.num_nodes, ex.num_set_words, ex.num_mask_bytes)
    assert (i.bits_per_index, i.max_triangles_in_leafs, i.max_triangles_encoded_in_leafs) == (ex.bits_per_index, ex.max_tri_in_leafs, ex.max_tri_encoded)
    assert i.cull_tests == ex.cull_tests
    n0, h0, s0, m0 = ex.data()
    n1, h1, s1, m1 = gx.download()
    assert np.array_equal(n0[:, 0], n1[:, 0])                       # topology: childrenIndex / leaf bits
    assert np.array_equal(h0, h1)
    assert np.array_equal(n0[h0 == 1, 1], n1[h1 == 1, 1])           # trianglesArrayIndex wherever the reference writes it
    assert np.array_equal(s0, s1)                                   # bit-packed triangle sets
    assert np.array_equal(m0, m1)                                   # byte masks


def test_exact_query_bit_exact(small, oracle):
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    ex = oracle.Exact(small["om"], small["box"], 6, 2, 32)
    gx = S.ExactOctreeSdf(small["gm"], small["box"], 6, 2, 32)
    pts = random_points_in_box(small["box"], 100000, seed=31)
    pts[:100] *= 3.0
    d0, g0, t0 = ex.query(pts, grad=True, tri=True)
    d1, g1, t1 = gx.get_distance(pts, gradient=True, triangle=True)
    inside = np.ones(len(pts), bool); inside[:100] = False
    assert np.array_equal(t0[inside], t1[inside])
    assert np.array_equal(bits(d0), bits(d1))
    assert np.array_equal(bits(g0[inside]), bits(g1[inside]))
    d2 = gx.get_distance(pts)
    assert np.array_equal(bits(d2), bits(d0))
    # and equal to the brute-force nearest over ALL triangles (signed through the fp64 BVH id can differ on ties: compare values)
    ids = small["om"].nearest(pts[100:2100])
    bf = np.array([small["om"].signed(ids[k], pts[100 + k]) for k in range(2000)], dtype=np.float32)
    np.testing.assert_allclose(d1[100:2100], bf, rtol=0, atol=1e-6)


def test_exact_gpu_matches_golden(gpu_ctx):
    import os
    import sdflib_amd as S
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_small.npz"))
    gm = S.Mesh(g["vertices"], g["triangles"], gpu_ctx)
    gx = S.ExactOctreeSdf(gm, g["box"], int(g["exact_depth"]), 1, int(g["exact_min_tri"]))
    n1, h1, s1, m1 = gx.download()
    assert np.array_equal(n1[:, 0], g["exact_nodes"][:, 0]) and np.array_equal(s1, g["exact_sets"]) and np.array_equal(m1, g["exact_masks"])
    d, gr, t = gx.get_distance(g["points"], gradient=True, triangle=True)
    assert np.array_equal(bits(d), bits(g["exact_dist"]))
    assert np.array_equal(t[64:], g["exact_tri"][64:]) and np.array_equal(bits(gr[64:]), bits(g["exact_grad"][64:]))


def test_exact_full_size_properties(gpu_ctx):
    """BASELINE.json configs[2] at full size (327 680 triangles, depth 7, min 128): the oracle's single-thread build would
    take minutes, so size-independent properties are checked instead: exact query == nearest-triangle distance through the
    independent fp64 BVH path, every packed set is strictly ascending, masks select subsets, leaves respect the threshold."""
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
    v, f = bumpy_icosphere(7)
    box = box_with_margin(v)
    gm = S.Mesh(v, f, gpu_ctx)
    gx = S.ExactOctreeSdf(gm, box, 7, 3, 128)
    i = gx.info
    assert i.bits_per_index == 19 and i.max_triangles_in_leafs >= 128 and i.num_nodes > 500000
    pts = random_points_in_box(box, 300000, seed=5)
    d, t = gx.get_distance(pts, triangle=True)
    ids = gm.nearest_triangle(pts)
    ref = gm.point_values(pts, ids)[:, 0]
    np.testing.assert_allclose(d, ref, rtol=0, atol=2e-6)            # same nearest feature up to ties
    assert (t == ids).mean() > 0.05                                   # ids differ only on ties (shared edges / vertices)
    nodes, has, sets, masks = gx.download()
    bits_per = i.bits_per_index
    # walk from the start grid, collect leaves at depth <= bitEnc (they own bit-packed sets), decode and check them
    G3 = i.start_grid_size ** 3
    todo = [(c, i.start_depth) for c in range(0, G3, 3)]
    leaves = []
    while todo and len(leaves) < 400:
        n_, dep = todo.pop()
        w = int(nodes[n_, 0])
        if w & 0x80000000:
            if dep <= i.bit_encoding_start_depth:
                leaves.append(n_)
        elif dep < i.bit_encoding_start_depth:
            base = w & 0x7FFFFFFF
            todo.extend((base + c, dep + 1) for c in range(0, 8, 3))
    assert len(leaves) > 100
    for n_ in leaves:
        assert has[n_] == 1
        at = int(nodes[n_, 1]); cnt = int(sets[at])
        assert 0 < cnt <= i.max_triangles_in_leafs
        words = sets[at + 1: at + 3 + (cnt * bits_per + 31) // 32].astype(np.uint64)
        vals = []
        for k in range(cnt):
            b = k * bits_per; w_ = b >> 5; off = b & 31
            two = (int(words[w_]) << 32) | int(words[w_ + 1])
            vals.append((two >> (64 - off - bits_per)) & ((1 << bits_per) - 1))
        vals = np.array(vals)
        assert (vals < len(f)).all() and (np.diff(vals) > 0).all()


@pytest.mark.parametrize("depth,start,min_tri,cuts", [(5, 2, 16, (0, 20, 21, 64)), (6, 3, 32, (0, 100, 300, 512)), (5, 1, 16, (0, 3, 8)), (4, 0, 8, (0, 1))])
def test_exact_shards_reassemble_to_the_single_build(small, depth, start, min_tri, cuts):
    """sdfhip_exact_build_shard / emit_shard: shards over ranges of start cells (emission order), emitted at the offsets the
    prefix sums give and assembled by sdflib_amd.distributed.assemble_exact, are bit-identical to the one-GPU build."""
    import sdflib_amd as S
    from sdflib_amd import distributed as sdist
    full = S.ExactOctreeSdf(small["gm"], small["box"], depth, start, min_tri)
    fn, fh, fs, fm = full.download()
    shards = [S.ExactShard(small["gm"], small["box"], depth, start, min_tri, (a, b)) for a, b in zip(cuts, cuts[1:])]
    sizes = [(s.info.num_nodes, s.info.num_set_words, s.info.num_mask_bytes) for s in shards]
    offs = sdist.exact_offsets(sizes, 8 ** start)
    order = np.argsort(sdist.exact_emission_rank(start), kind="stable")
    parts = []
    for s, o, (a, b) in zip(shards, offs, zip(cuts, cuts[1:])):
        assert np.array_equal(s.cells(), np.sort(order[a:b]))
        parts.append(dict(cells=s.cells(), **s.emit(*o)))
    nodes, has, sets, masks = sdist.assemble_exact(parts, 8 ** start)
    assert np.array_equal(nodes, fn) and np.array_equal(has, fh) and np.array_equal(sets, fs) and np.array_equal(masks, fm)
    assert max(s.info.max_triangles_in_leafs for s in shards) == full.info.max_triangles_in_leafs
    assert max(s.info.max_triangles_encoded_in_leafs for s in shards) == full.info.max_triangles_encoded_in_leafs


def test_exact_sharded_build_through_rccl_world1(small):
    """The N>1 path of the Exact build (shard -> all-gather over RCCL -> from_parts) with a 1-rank group, then queries."""
    import os
    import torch
    import torch.distributed as dist
    import sdflib_amd as S
    from sdflib_amd import distributed as sdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29613")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0)); created = True
    try:
        dev = torch.device("cuda", 0)
        tree, _ = sdist.build_exact_sharded(small["gm"], small["box"], 5, 2, 16, 0, 1, dev)
        full = S.ExactOctreeSdf(small["gm"], small["box"], 5, 2, 16)
        for a, b in zip(tree.download(), full.download()):
            assert np.array_equal(a, b)
        rng = np.random.default_rng(9)
        pts = ((rng.random((20000, 3), dtype=np.float32) * 2 - 1) * 1.5).astype(np.float32)
        assert np.array_equal(bits(tree.get_distance(pts)), bits(full.get_distance(pts)))
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.gpu
def test_tree_keeps_a_temporary_mesh_alive(gpu_ctx, oracle):
    """ExactOctreeSdf(Mesh(...), ...) with a temporary Mesh — the tree reads the mesh's TriangleData on the device,
    so it must hold a reference (the C++ class keeps mMesh, from_parts sets _mesh)."""
    import gc
    import sdflib_amd as S
    from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
    v, f = bumpy_icosphere(3)
    box = box_with_margin(v)
    t = S.ExactOctreeSdf(S.Mesh(v, f, gpu_ctx), box, 5, 1, 16)
    gc.collect()
    junk = [S.Mesh(v * 0.5, f, gpu_ctx) for _ in range(3)]          # would recycle the freed device blocks
    pts = random_points_in_box(box, 2000, seed=3)
    want = oracle.Exact(oracle.Mesh(v, f), box, 5, 1, 16).query(pts)
    assert np.array_equal(bits(t.get_distance(pts)), bits(want))
    del junk


def test_exact_scalar_host_entry_equals_device_and_oracle(small, oracle):
    import sdflib_amd as S
    from sdflib_amd.meshgen import random_points_in_box
    ex = oracle.Exact(small["om"], small["box"], 6, 2, 32)
    gx = S.ExactOctreeSdf(small["gm"], small["box"], 6, 2, 32)
    pts = random_points_in_box(small["box"], 20000, seed=41)
    pts[::9] *= 2.0
    d0, g0, t0 = ex.query(pts, grad=True, tri=True)
    dd, gd, td = gx.get_distance(pts, gradient=True, triangle=True)           # device (sorted path)
    box = gx.get_grid_bounding_box()
    inside = ((pts >= box[:3]) & (pts < box[3:])).all(axis=1)
    for k in range(0, 20000, 97):
        m = 1 + (k % 32)
        ds, gs, ts = gx.get_distance(pts[k:k + m], gradient=True, triangle=True)      # host copies (<= 32 points)
        assert np.array_equal(bits(ds), bits(d0[k:k + m])) and np.array_equal(bits(ds), bits(dd[k:k + m]))
        ins = inside[k:k + m]
        assert np.array_equal(bits(gs[ins]), bits(g0[k:k + m][ins])) and np.array_equal(ts[ins], t0[k:k + m][ins])


@pytest.mark.parametrize("env", [{}, {"SDFHIP_EXACT_LISTS_MB": "0"}], ids=["lists", "lists-over-the-cap"])
def test_both_batched_query_kernels_answer_like_the_oracle(env):
    """Round 4: the batched query reads the leaves' DECODED triangle lists (made once per tree) through a pipelined kernel; trees whose lists
    would exceed SDFHIP_EXACT_LISTS_MB (or half of the device's free memory, or whose allocation fails) keep round 3's decoding kernel.  Both
    paths (the switch is read once per process, hence the child processes) against the oracle: distances, gradients and triangle ids, incl. points outside the grid, empty leaves and long runs."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import sdflib_amd as S
from oracle import pyoracle as O
from sdflib_amd.meshgen import bumpy_icosphere, torus_knot, box_with_margin, random_points_in_box
b = lambda a: np.ascontiguousarray(a).view(np.uint32)
for name, (v, f), (depth, start, mt) in (("sphere", bumpy_icosphere(4), (6, 2, 16)), ("knot", torus_knot(24, 8), (5, 1, 32)), ("sphere-coarse", bumpy_icosphere(3), (3, 1, 8))):
    box = box_with_margin(v)
    m = S.Mesh(v, f); e = S.ExactOctreeSdf(m, box, depth, start, mt)
    oe = O.Exact(O.Mesh(v, f), box, depth, start, mt)
    pts = random_points_in_box(box, 150001, seed=11); pts[::89] *= 2.5
    pts[1000:3000] = pts[1000] + (pts[1000:3000] - pts[1000]) * 1e-3          # 2000 points in one leaf: runs longer than a wave
    d0, g0, t0 = oe.query(pts, grad=True, tri=True)
    box = np.asarray(box, dtype=np.float32)
    inside = np.all((pts >= box[:3]) & (pts <= box[3:]), axis=1)          # gradient / id of a point outside the grid are not defined by the reference
    for rep in range(2):          # the first call makes the tables
        d, g, t = e.get_distance(pts, gradient=True, triangle=True)
        assert np.array_equal(b(d), b(d0)) and np.array_equal(b(g[inside]), b(g0[inside])) and np.array_equal(t[inside], t0[inside]), name
    assert np.array_equal(b(e.get_distance(pts)), b(d0)), name
print("exact paths ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    assert r.returncode == 0 and "exact paths ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
