"""Every environment switch the library still reads (20 after round 5's pruning, 22 with round 6's SDFHIP_WELD and SDFHIP_HOST_OVERLAP; `grep -o 'getenv("SDFHIP_[A-Z_0-9]*")' sdflib_amd/csrc/*`)
is flipped by a test: the ones below, plus SDFHIP_BVH_BUILD / _DEVICE_SUBTREES / SDFHIP_TIMING (test_gpu_octree.py: the hybrid BVH walk),
SDFHIP_BVH_SORT_THREADS / _PAR_DEPTH / _MIN_PARALLEL / _PAR_PARTITION (test_planner_cpu.py), SDFHIP_MULTI_CUTS (test_gpu_baseline_configs.py),
SDFHIP_EXACT_LISTS_MB (test_gpu_exact.py) and SDFHIP_QUERY_CHUNK (test_gpu_octree.py).  The switches are read once per process, hence the
child processes; every child compares what the switched library produces with the oracle, bit for bit."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_COMMON = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import sdflib_amd as S
from oracle import pyoracle as O
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
b = lambda a: np.ascontiguousarray(a).view(np.uint32)
v, f = bumpy_icosphere(4); box = box_with_margin(v)
ctx = S.Context(0); gm = S.Mesh(v, f, ctx); om = O.Mesh(v, f)
pts = random_points_in_box(box, 40000, seed=5)
def trees_equal_the_oracle():
    t = S.OctreeSdf(gm, box, 6, 2, 1e-3, num_threads=2)
    ot = O.Octree(om, box, 6, 2, 1e-3, vertex_cache=False, layout=O.LAYOUT_SUBTREES)
    assert np.array_equal(t.get_octree_data(), ot.data())
    d, g = t.get_distance(pts, gradient=True); d0, g0 = ot.query(pts, grad=True)
    assert np.array_equal(b(d), b(d0)) and np.array_equal(b(g), b(g0))
    c = S.OctreeSdf(gm, box, 5, 2, 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2)
    oc = O.Octree(om, box, 5, 2, 1e-3, continuity=True)
    assert np.array_equal(c.get_octree_data(), oc.data())
    e = S.ExactOctreeSdf(gm, box, 5, 1, 16); oe = O.Exact(om, box, 5, 1, 16)
    n0, h0, s0, m0 = oe.data(); n1, h1, s1, m1 = e.download()
    assert np.array_equal(s0, s1) and np.array_equal(m0, m1[:len(m0)]) and np.array_equal(n0[:, 0], n1[:, 0])
    assert np.array_equal(b(e.get_distance(pts)), b(oe.query(pts)))
    return t
''' % ROOT

_CASES = {
    # the step-by-step reference traversal instead of the two-phase search: same ids, same trees
    "nearest-exact": ({"SDFHIP_NEAREST": "exact"}, "trees_equal_the_oracle(); assert np.array_equal(gm.nearest_triangle(pts), om.nearest(pts))"),
    # allocation diagnostics: plain hipMalloc for every transient block / fresh blocks filled with a pattern / guard words + overlap registry
    "no-pool": ({"SDFHIP_NO_POOL": "1"}, "trees_equal_the_oracle()"),
    # large host-pointer batches as one upload / one launch / one download (the overlapped two-thread path off): same bits as the device-pointer path
    "host-overlap-0": ({"SDFHIP_HOST_OVERLAP": "0"}, r'''
import torch
t = trees_equal_the_oracle()
big = random_points_in_box(box, 2_600_000, seed=9)
d, g = t.get_distance(big, gradient=True)
dt, gt = t.get_distance(torch.from_numpy(big).cuda(), gradient=True)
assert np.array_equal(b(d), b(dt.cpu().numpy())) and np.array_equal(b(g), b(gt.cpu().numpy()))
'''),
    # the seam welding's round-1..5 host planner (std::map based) instead of the device passes: same TriangleData, same tree on a welded soup
    "weld-host": ({"SDFHIP_WELD": "host"}, r'''
from sdflib_amd.meshgen import triangle_soup
sv, sf = triangle_soup(v, f)
bbox = np.concatenate([sv.min(axis=0), sv.max(axis=0)])
wm, wo = S.Mesh(sv, sf, ctx, bbox=bbox), O.Mesh(sv, sf, bbox)
assert wm.edge_stats()["welded_half_edges"] == 3 * len(sf)
assert np.array_equal(b(wm.triangle_data()), b(wo.triangle_data()))
assert np.array_equal(S.OctreeSdf(wm, box, 5, 2, 1e-3, num_threads=2).get_octree_data(), O.Octree(wo, box, 5, 2, 1e-3, vertex_cache=False, layout=O.LAYOUT_SUBTREES).data())
'''),
    "poison-alloc": ({"SDFHIP_POISON_ALLOC": "0xCD"}, "trees_equal_the_oracle()"),
    "alloc-check": ({"SDFHIP_ALLOC_CHECK": "1"}, "trees_equal_the_oracle()"),
    # nothing may stay cached when a build returns
    "cache-keep-0": ({"SDFHIP_CACHE_KEEP_MB": "0"}, "t = trees_equal_the_oracle(); print('CACHED', ctx.cached_bytes())"),
    # a tree that arrives as an array gives its array up to the first query's layout from 0 MB on
    "compact-above-0": ({"SDFHIP_COMPACT_ABOVE_MB": "0"}, r'''
t = trees_equal_the_oracle(); w = t.get_octree_data(); i = t.info
a = S.OctreeSdf.from_data(ctx, w, i.box_min, i.box_max, i.start_grid_size, i.max_depth, i.value_range, i.min_border_value, cell_size=i.start_grid_cell_size)
d = a.get_distance(pts)
assert a.device_bytes() < 1.1 * 4 * len(w), (a.device_bytes(), 4 * len(w))
assert np.array_equal(b(d), b(t.get_distance(pts))) and np.array_equal(a.get_octree_data(), w)
'''),
    # lattices answered by the point kernel instead of the leaf-driven column kernels: same bits
    "lattice-points": ({"SDFHIP_LATTICE_POINTS": "1"}, r'''
t = trees_equal_the_oracle(); n = 40
bb = t.get_grid_bounding_box(); step = np.full(3, (bb[3] - bb[0]) / n, np.float32); org = (bb[:3] + 0.5 * step).astype(np.float32)
k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
P = (org + np.stack([i, j, k], -1).reshape(-1, 3).astype(np.float32) * step).astype(np.float32)
for mode in (S.EVAL_EXACT, S.EVAL_FAST):
    dl, gl = t.get_distance_grid(org, step, (n, n, n), gradient=True, eval_mode=mode)
    dp, gp = t.get_distance(P, gradient=True, eval_mode=mode)
    assert np.array_equal(b(dl).ravel(), b(dp)) and np.array_equal(b(gl).reshape(-1, 3), b(gp))
'''),
    # the device BVH build with parts of at most 64 keys leaving the global-memory rounds and subtrees of at most 256 triangles finished in
    # LDS: many more rounds and levels on a small mesh; the tree must be the oracle's in every bit
    "bvh-part": ({"SDFHIP_BVH_PART": "64", "SDFHIP_BVH_DEVICE_SUBTREES": "256"}, r'''
sys.path.insert(0, os.path.join(%r, "tests"))
from test_planner_cpu import same_tree
v5, f5 = bumpy_icosphere(5); m5 = S.Mesh(v5, f5, ctx); m5.build_bvh()
assert same_tree(v5, f5, arrays=m5.bvh_arrays())
trees_equal_the_oracle()
''' % ROOT),
    # the in-process multi-device build over staged copies (what SDFHIP_MULTI_TRANSPORT=copy selects also between DISTINCT devices, where RCCL
    # is the default: tools/gpu_multi_smoke.sh flips it there): two logical devices on this GPU == the single build
    "multi-transport-copy": ({"SDFHIP_MULTI_TRANSPORT": "copy"}, r'''
import ctypes as C
from sdflib_amd._lib import lib, check, OctreeParams
L = lib(); devs = (C.c_int * 2)(0, 0); M = C.c_void_p()
check(L.sdfhip_multi_create(devs, 2, C.byref(M)))
L.sdfhip_multi_transport.restype = C.c_char_p
assert L.sdfhip_multi_transport(M) == b"copy"
single = S.OctreeSdf(gm, box, 6, 2, 1e-3, num_threads=2).get_octree_data()
p = OctreeParams()
for k in range(3): p.box_min[k] = box[k]; p.box_max[k] = box[3 + k]
p.depth, p.start_depth, p.rule, p.algorithm, p.layout, p.fit_mode = 6, 2, S.RULE_TRAPEZOIDAL, S.ALG_NO_CONTINUITY, S.LAYOUT_SUBTREES, S.FIT_EXACT
p.rule_params[0] = 1e-3
trees = (C.c_void_p * 2)()
vv = np.ascontiguousarray(v, np.float32); ff = np.ascontiguousarray(f, np.uint32)
check(L.sdfhip_multi_octree_build(M, vv.ctypes.data_as(C.c_void_p), len(vv), ff.ctypes.data_as(C.c_void_p), len(ff), None, C.byref(p), None, trees))
for r in range(2):
    got = np.empty(len(single), np.uint32)
    check(L.sdfhip_octree_download(trees[r], got.ctypes.data_as(C.c_void_p), 0))
    assert np.array_equal(got, single), r
    L.sdfhip_octree_destroy(trees[r])
L.sdfhip_multi_destroy(M)
'''),
}


@pytest.mark.parametrize("name", sorted(_CASES))
def test_switch_changes_no_bit(name):
    env, body = _CASES[name]
    r = subprocess.run([sys.executable, "-c", _COMMON + body + "\nprint('switch ok')\n"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    assert r.returncode == 0 and "switch ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    if name == "cache-keep-0":
        # what stays cached when a build returns: with a mark of 0 MB only what the mark does not govern (the nearest search's lists), with
        # the default mark (1/32 of the device) the build's transient blocks as well
        kept0 = int(r.stdout.split("CACHED")[1].split()[0])
        r2 = subprocess.run([sys.executable, "-c", _COMMON + body + "\nprint('switch ok')\n"], capture_output=True, text=True, env=dict(os.environ), timeout=600)
        assert r2.returncode == 0, r2.stderr[-2000:]
        kept = int(r2.stdout.split("CACHED")[1].split()[0])
        assert kept0 < kept and kept0 <= (96 << 20), (kept0, kept)
