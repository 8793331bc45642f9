"""Two ranks on ONE GPU (gloo for the collectives, both processes on cuda:0): the complete N>1 build path — shard build on the
device, offset exchange, emit with absolute indices, all-gather, reassembly — against the single-process build; and the
CONTINUITY build with its traversals shared out through sdfhip_exchange."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, host_bvh=False):
    try:
        if host_bvh:
            os.environ["SDFHIP_BVH_BUILD"] = "host"      # before the library is loaded: rank 0 plans, the arrays are broadcast
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sys.path.insert(0, ROOT)
        import sdflib_amd as S
        from sdflib_amd import distributed as sdist
        from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        ctx = S.Context(0, use_torch_stream=True)
        v, f = bumpy_icosphere(4)
        box = box_with_margin(v)
        mesh = S.Mesh(v, f, ctx)
        sdist.share_bvh(mesh, rank, world, dev)                      # built by every rank on the device; host planner: rank 0 plans, broadcast, import
        assert sdist.bvh_built_on_device() == (not host_bvh)
        assert mesh.build_bvh() == 0.0
        ref_mesh = S.Mesh(v, f, ctx)                                 # plans its own
        probe = ((np.random.default_rng(5).random((20000, 3), dtype=np.float32) * 2 - 1) * 1.4).astype(np.float32)
        assert np.array_equal(mesh.nearest_triangle(probe), ref_mesh.nearest_triangle(probe)), "shared BVH differs from a locally planned one"
        tree, _ = sdist.build_octree_sharded(mesh, box, 6, 3, 1e-3, rank, world, dev)
        single = S.OctreeSdf(mesh, box, 6, 3, 1e-3, num_threads=2)
        assert np.array_equal(tree.get_octree_data(), single.get_octree_data()), "sharded OctreeSdf differs from the single build"
        assert tree.info.value_range == single.info.value_range and tree.info.min_border_value == single.info.min_border_value
        assert list(tree.info.leaves_per_depth) == list(single.info.leaves_per_depth)
        ex, _ = sdist.build_exact_sharded(mesh, box, 5, 2, 16, rank, world, dev)
        ex1 = S.ExactOctreeSdf(mesh, box, 5, 2, 16)
        for a, b in zip(ex.download(), ex1.download()):
            assert np.array_equal(a, b), "sharded ExactOctreeSdf differs from the single build"
        assert ex.info.max_triangles_in_leafs == ex1.info.max_triangles_in_leafs
        # CONTINUITY: every rank builds the whole tree, the traversals of each sample batch are shared through the exchange
        ct, tm = sdist.build_continuity_sharded(mesh, box, 6, 3, 1e-3, rank, world, dev)
        c1 = S.OctreeSdf(mesh, box, 6, 3, 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2)      # exchange removed again: a plain local build
        assert tm["exchange_bytes"] > 0
        assert np.array_equal(ct.get_octree_data(), c1.get_octree_data()), "CONTINUITY tree built with shared traversals differs from the single build"
        assert ct.info.value_range == c1.info.value_range and ct.info.min_border_value == c1.info.min_border_value
        rng = np.random.default_rng(rank)
        pts = ((rng.random((20000, 3), dtype=np.float32) * 2 - 1) * 1.4).astype(np.float32)
        assert np.array_equal(tree.get_distance(pts).view(np.uint32), single.get_distance(pts).view(np.uint32))
        assert np.array_equal(ex.get_distance(pts).view(np.uint32), ex1.get_distance(pts).view(np.uint32))
        # queries: rank 0's tree reaches the others with one broadcast; each rank answers its contiguous share
        got = sdist.broadcast_octree(single if rank == 0 else None, ctx, dev, src=0)
        assert np.array_equal(got.get_octree_data(), single.get_octree_data()) and got.info.min_border_value == single.info.min_border_value
        b, e = sdist.query_range(len(pts), rank, world)
        assert np.array_equal(got.get_distance(pts[b:e]).view(np.uint32), single.get_distance(pts)[b:e].view(np.uint32))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:       # surface the failure in the parent
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world,host_bvh", [(2, False), (3, False), (2, True)])
def test_sharded_builds_with_several_ranks_on_one_gpu(world, host_bvh):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + (os.getpid() % 2000) + world + (10 if host_bvh else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, host_bvh)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(msg == "ok" for _, msg in results), "\n".join(str(m) for _, m in results)


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_two_ranks_on_one_gpu(launcher):
    """launcher=self: `python bench.py --gpus 2` with no WORLD_SIZE in the environment must start its own two ranks.
    bench.py's N>1 path (rendezvous from the environment, sharded build, barrier-bracketed timing, MAX over ranks, one JSON
    line from rank 0) with two ranks sharing the GPU through the gloo test hook."""
    import json
    import subprocess
    env = dict(os.environ, SDFHIP_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(33500 + os.getpid() % 2000)]
    cmd = (head if launcher == "torchrun" else [sys.executable]) + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--subdiv", "5", "--depth", "6", "--queries", "1000000",
           "--no-cpu-baseline", "--no-build-1m" if launcher == "torchrun" else "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * d["per_gpu_mqueries_s"]) < 1e-6 * d["value"] + 0.02
    assert d["roofline"]["frac"] > 0 and d["build"]["exchange_s"] >= 0
    assert d["collectives"]["ranks_seen"] == 2 and d["collectives"]["rank_sum_ok"] and d["collectives"]["octree_bytes_all_gathered_per_rank"] > 0
    if launcher == "torchrun":
        c = d["extras"]["continuity_octree"]
        assert c["ranks_sharing_traversals"] == 2 and c["exchange_bytes"] > 0 and c["words"] > 0
    else:       # the 1.31 M-triangle build at N = 2 with the split north_star asks for: serial / sharded / exchange
        b = d["build_1m"]
        assert b["n_gpus"] == 2 and b["triangles"] == 1310720 and b["words"] == 20058064
        sp = b["split"]
        assert sp["serial_s"] > 0 and sp["sharded_s"] > 0 and sp["exchange_s"] >= 0 and sp["exchange_bytes_per_rank"] == 4 * b["words"]
        assert b["end_to_end_s"] >= sp["serial_s"] + sp["sharded_s"]


def test_bench_runs_a_mesh_file(tmp_path):
    """`bench.py --mesh PATH`: the loaders take a PLY / OBJ when one is supplied — here a small PLY written by meshio —
    and the line says so (`data: "file"`, the file's name and triangle count in `config.workload`)."""
    import json
    import subprocess
    from sdflib_amd import meshio
    from sdflib_amd.meshgen import bumpy_icosphere
    v, f = bumpy_icosphere(4)
    path = os.path.join(str(tmp_path), "bumpy4.ply")
    meshio.write_ply(path, v, f)
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mesh", path, "--depth", "6", "--steps", "2", "--warmup", "1", "--queries", "500000", "--no-cpu-baseline", "--no-build-1m", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["data"] == "file" and "bumpy4.ply" in d["config"]["workload"] and str(len(f)) in d["config"]["workload"]
    assert d["value"] > 0 and d["build"]["time_to_first_query_s"] > 0
    assert d["config"]["mesh_edges"] == {"unmatched_edges": 0, "welded_half_edges": 0}
    # an unwelded export of the same surface: the file's box switches the seam welding on (loader semantics), every edge is re-paired
    from sdflib_amd.meshgen import triangle_soup
    sv, sf = triangle_soup(v, f)
    spath = os.path.join(str(tmp_path), "soup4.ply")
    meshio.write_ply(spath, sv, sf)
    cmd[cmd.index("--mesh") + 1] = spath
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d2 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d2["config"]["mesh_edges"] == {"unmatched_edges": 3 * len(sf), "welded_half_edges": 3 * len(sf)}
    assert d2["build"]["mesh_prep_s"] > 0 and d2["config"]["octree_leaves"] > 0
