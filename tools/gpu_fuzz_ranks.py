"""Differential fuzzing of the N > 1 paths on ONE GPU (gloo for the collectives, all ranks on cuda:0): random meshes / parameters /
world sizes; sharded OctreeSdf, sharded ExactOctreeSdf and the CONTINUITY build with shared traversals must equal the single-process
builds bit for bit on every rank.  Usage: tools/gpu_fuzz_ranks.py [cases] [first seed]."""
import os, sys, time, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, seed0, cases, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import sdflib_amd as S
        from sdflib_amd import distributed as sdist
        from sdflib_amd.meshgen import icosphere, bumpy_icosphere, cube_mesh, box_with_margin
        dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
        ctx = S.Context(0, use_torch_stream=True)
        bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
        for s in range(seed0, seed0 + cases):
            rng = np.random.default_rng(s)
            kind = int(rng.integers(0, 3))
            v, f = (icosphere(int(rng.integers(1, 4))) if kind == 0 else bumpy_icosphere(int(rng.integers(1, 5))) if kind == 1 else cube_mesh())
            v = (v * np.float32(rng.uniform(0.3, 3)) + rng.normal(0, 5, 3).astype(np.float32)).astype(np.float32)
            box = box_with_margin(v, margin=float(rng.uniform(0.05, 0.4)))
            start = int(rng.integers(1, 4)) if world <= 8 else 2
            depth = int(rng.integers(start + 1, 7)); thr = float(10 ** rng.uniform(-3.5, -2))
            mesh = S.Mesh(v, f, ctx)
            pts = (box[:3] + rng.random((20000, 3), dtype=np.float32) * (box[3:] - box[:3])).astype(np.float32)
            tree, _ = sdist.build_octree_sharded(mesh, box, depth, start, thr, rank, world, dev)
            single = S.OctreeSdf(mesh, box, depth, start, thr, num_threads=2)
            assert np.array_equal(tree.get_octree_data(), single.get_octree_data()), f"seed {s}: sharded OctreeSdf"
            dt, ds = tree.get_distance(pts), single.get_distance(pts)
            if not np.array_equal(bits(dt), bits(ds)):      # which side is wrong, and does it stay wrong
                from oracle import pyoracle as O
                i = single.info
                ref = O.octree_query_raw(single.get_octree_data(), single.get_grid_bounding_box(), i.start_grid_size, i.min_border_value, pts)
                dt2, ds2 = tree.get_distance(pts), single.get_distance(pts)
                raise AssertionError(f"seed {s} rank {rank}: sharded OctreeSdf answers: reassembled tree differs from the oracle at {int((bits(dt) != bits(ref)).sum())} points "
                                     f"(asked again: {int((bits(dt2) != bits(ref)).sum())}), single build at {int((bits(ds) != bits(ref)).sum())} (again: {int((bits(ds2) != bits(ref)).sum())}); "
                                     f"cell sizes {tree.info.start_grid_cell_size!r} / {i.start_grid_cell_size!r}, first differing point {pts[np.nonzero(bits(dt) != bits(ds))[0][0]]}, box {box}")
            ct, _ = sdist.build_continuity_sharded(mesh, box, depth, start, thr, rank, world, dev)
            c1 = S.OctreeSdf(mesh, box, depth, start, thr, init_algorithm=S.ALG_CONTINUITY, num_threads=2)
            assert np.array_equal(ct.get_octree_data(), c1.get_octree_data()), f"seed {s}: CONTINUITY with shared traversals"
            edepth = int(rng.integers(start + 2, start + 5)); mint = int(rng.choice([2, 8, 32]))
            ex, _ = sdist.build_exact_sharded(mesh, box, edepth, start, mint, rank, world, dev)
            e1 = S.ExactOctreeSdf(mesh, box, edepth, start, mint)
            for a, b in zip(ex.download(), e1.download()):
                assert np.array_equal(a, b), f"seed {s}: sharded ExactOctreeSdf"
            assert np.array_equal(bits(ex.get_distance(pts)), bits(e1.get_distance(pts))), f"seed {s}: sharded ExactOctreeSdf answers"
            got = sdist.broadcast_octree(single if rank == 0 else None, ctx, dev, src=0)
            assert np.array_equal(bits(got.get_distance(pts)), bits(single.get_distance(pts))), f"seed {s}: broadcast tree answers"
            if rank == 0: print(f"seed {s}: world {world} T={len(f)} d={depth}/{start} exact d={edepth} min={mint} ok", flush=True)
        dist.barrier(); dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:      # noqa: BLE001
        q.put((rank, traceback.format_exc()))
        raise


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    worlds = tuple(int(w) for w in os.environ.get("FUZZ_WORLDS", "2,3,5").split(","))
    for world in worlds:
        c = mp.get_context("spawn"); q = c.Queue()
        procs = [c.Process(target=worker, args=(r, world, 34000 + world + (os.getpid() % 500), seed0, cases, q)) for r in range(world)]
        for p in procs: p.start()
        res = [q.get(timeout=1200) for _ in range(world)]
        for p in procs: p.join(timeout=60)
        for r, msg in res:
            if msg != "ok": bad += 1; print(f"world {world} rank {r}: {msg}")
    print(f"{len(worlds) * cases} multi-rank cases, {bad} failing ranks")
    sys.exit(1 if bad else 0)
