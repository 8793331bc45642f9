"""BVH planner (host code of the product, sdflib_amd/csrc/bvh.hip) against the oracle's restatement of
tmd::TriangleMeshDistance::_build_tree (TriangleMeshDistance.h:421-490) — no GPU involved.  The two trees are walked together from
the root: every pair of child spheres must agree in all 64 bits of every double, every leaf must hold the same triangle.  Meshes are
chosen for what decides the tree: std::sort's permutation of tied keys (shared first vertices, duplicated triangles, symmetric
meshes), degenerate triangles, widely different scales."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def planned(v, f):
    from sdflib_amd._lib import lib
    v = np.ascontiguousarray(v, np.float32); f = np.ascontiguousarray(f, np.uint32)
    n = max(len(f) - 1, 1)
    sph = np.empty(8 * n, np.float64); kids = np.empty(2 * n, np.int32)
    rc = lib().sdfhip_test_plan_bvh(v.ctypes.data_as(C.c_void_p), len(v), f.ctypes.data_as(C.c_void_p), len(f), sph.ctypes.data_as(C.c_void_p),
                                    kids.ctypes.data_as(C.c_void_p), None)
    assert rc == 0
    return sph, kids


def same_tree(v, f, arrays=None):
    """arrays = (spheres, children) of a tree planned elsewhere (the GPU tests pass what the device holds); default: the host planner's."""
    from oracle import pyoracle as O
    sph, kids = arrays if arrays is not None else planned(v, f)
    om = O.Mesh(np.ascontiguousarray(v, np.float32), np.ascontiguousarray(f, np.uint32))
    osph, olr = om.bvh_export()
    osph = np.asarray(osph, np.float64).reshape(-1, 8); olr = np.asarray(olr, np.int32).reshape(-1, 2)
    if len(f) == 1:
        return True                                       # a single leaf: nothing to compare but the dummy node's shape
    stack = [(0, 0)]                                      # (our inner node, oracle node)
    leaves = 0
    while stack:
        a, b = stack.pop()
        assert olr[b, 0] >= 0, "oracle node is a leaf where the planner has an inner node"
        assert np.array_equal(sph[8 * a:8 * a + 8].view(np.uint64), osph[b].view(np.uint64)), (a, b, sph[8 * a:8 * a + 8], osph[b])
        for side in (0, 1):
            ours, theirs = int(kids[2 * a + side]), int(olr[b, side])
            if ours < 0:
                assert olr[theirs, 0] == -1 and olr[theirs, 1] == ~ours, (a, side, ~ours, olr[theirs])
                leaves += 1
            else:
                stack.append((ours, theirs))
    assert leaves == len(f)
    return True


def test_planner_tree_equals_the_oracles_tree():
    from sdflib_amd import meshgen
    rng = np.random.default_rng(5)
    v, f = meshgen.icosphere(3)
    assert same_tree(v, f)                                                       # symmetric: tied keys on every axis
    v, f = meshgen.bumpy_icosphere(4)
    assert same_tree(v, f)
    assert same_tree(v * np.array([1e-3, 7.0, 250.0], np.float32) + np.array([1e3, -2.0, 0.5], np.float32), f)     # scales / offsets
    assert same_tree(v, np.concatenate([f, f[::3], f[::5]]))                     # duplicated triangles: fully tied pairs
    assert same_tree(v, f[rng.permutation(len(f))])                              # another input order, another tie permutation
    assert same_tree(v, f[:, [1, 2, 0]])                                         # another first vertex = other keys
    cv, cf = meshgen.cube_mesh()
    assert same_tree(cv, cf)
    assert same_tree(cv, np.concatenate([cf, np.array([[0, 0, 1], [2, 2, 2]], np.uint32)]))      # degenerate triangles
    assert same_tree(cv[:3], np.array([[0, 1, 2]], np.uint32))                   # one triangle
    assert same_tree(cv, cf[:2])                                                 # two triangles
    sv, sf = meshgen.triangle_soup(v, f[:700])
    assert same_tree(sv, sf)


def test_planner_result_does_not_depend_on_its_threading():
    from sdflib_amd import meshgen
    v, f = meshgen.bumpy_icosphere(6)                                            # 81 920 triangles: wide-node path, parallel partition off / on
    ref = planned(v, f)
    for env in ({"SDFHIP_BVH_SORT_THREADS": "1", "SDFHIP_BVH_PAR_DEPTH": "0"}, {"SDFHIP_BVH_MIN_PARALLEL": "512", "SDFHIP_BVH_PAR_PARTITION": "1024"},
                {"SDFHIP_BVH_PAR_DEPTH": "3"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            got = planned(v, f)
        finally:
            for k, val in old.items():
                if val is None: os.environ.pop(k, None)
                else: os.environ[k] = val
        assert np.array_equal(got[0].view(np.uint64), ref[0].view(np.uint64)) and np.array_equal(got[1], ref[1]), env


def test_planner_works_in_a_forked_child():
    """The planner's worker pool is created on first use and a fork()ed child inherits the pool object without its threads: the child
    must plan inline (same tree) instead of waiting for workers that do not exist."""
    from sdflib_amd import meshgen
    v, f = meshgen.bumpy_icosphere(5)                        # 20 480 triangles: parallel phases are used in the parent
    ref = planned(v, f)
    pid = os.fork()
    if pid == 0:
        code = 3
        try:
            got = planned(v, f)
            code = 0 if (np.array_equal(got[0].view(np.uint64), ref[0].view(np.uint64)) and np.array_equal(got[1], ref[1])) else 4
        finally:
            os._exit(code)
    import signal, time
    deadline = time.time() + 60
    while True:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done: break
        if time.time() > deadline:
            os.kill(pid, signal.SIGKILL); os.waitpid(pid, 0)
            raise AssertionError("the forked child hung in the planner")
        time.sleep(0.05)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
