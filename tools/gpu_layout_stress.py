"""Dev stress: built trees (born with the query layout) against copies that arrive as arrays (layout derived by the first query) and against
the oracle's answers, many small random cases, optionally several processes at once on one GPU (STRESS_PROCS).  Prints every disagreement
with which side is wrong.  Usage: tools/gpu_layout_stress.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def work(rank, cases, seed0):
    import torch
    import sdflib_amd as S
    from oracle import pyoracle as O
    from sdflib_amd.meshgen import icosphere, bumpy_icosphere, cube_mesh, box_with_margin
    torch.cuda.set_device(0)
    ctx = S.Context(0, use_torch_stream=True)
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
    bad = 0
    for s in range(seed0 + 100000 * rank, seed0 + 100000 * rank + cases):
        rng = np.random.default_rng(s)
        kind = int(rng.integers(0, 3))
        v, f = (icosphere(int(rng.integers(1, 4))) if kind == 0 else bumpy_icosphere(int(rng.integers(1, 5))) if kind == 1 else cube_mesh())
        v = (v * np.float32(rng.uniform(0.3, 3)) + rng.normal(0, 5, 3).astype(np.float32)).astype(np.float32)
        box = box_with_margin(v, margin=float(rng.uniform(0.05, 0.4)))
        start = int(rng.integers(1, 4)); depth = int(rng.integers(start + 1, 7)); thr = float(10 ** rng.uniform(-3.5, -2))
        alg = S.ALG_CONTINUITY if rng.random() < 0.3 else S.ALG_NO_CONTINUITY
        mesh = S.Mesh(v, f, ctx)
        pts = (box[:3] + rng.random((20000, 3), dtype=np.float32) * (box[3:] - box[:3])).astype(np.float32)
        t = S.OctreeSdf(mesh, box, depth, start, thr, init_algorithm=alg, num_threads=2)
        w = t.get_octree_data(); i = t.info
        a = S.OctreeSdf.from_data(ctx, w, i.box_min, i.box_max, i.start_grid_size, i.max_depth, i.value_range, i.min_border_value, cell_size=i.start_grid_cell_size)
        da, dt = a.get_distance(pts), t.get_distance(pts)
        if not np.array_equal(bits(da), bits(dt)):
            bad += 1
            ref = O.octree_query_raw(w, t.get_grid_bounding_box(), i.start_grid_size, i.min_border_value, pts)
            na, nt = int((bits(da) != bits(ref)).sum()), int((bits(dt) != bits(ref)).sum())
            da2, dt2 = a.get_distance(pts), t.get_distance(pts)
            print(f"rank {rank} seed {s}: T={len(f)} d={depth}/{start} alg={alg}: array-born differs from the oracle at {na} points, layout-born at {nt}; asked again: {int((bits(da2) != bits(ref)).sum())} / {int((bits(dt2) != bits(ref)).sum())}; array re-downloaded equal: {np.array_equal(t.get_octree_data(), w)}", flush=True)
        a.close(); t.close(); mesh.close()
    print(f"rank {rank}: {cases} cases, {bad} disagreements", flush=True)


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000000
    procs = int(os.environ.get("STRESS_PROCS", "1"))
    if procs == 1:
        work(0, cases, seed0)
    else:
        import torch.multiprocessing as mp
        c = mp.get_context("spawn")
        ps = [c.Process(target=work, args=(r, cases, seed0)) for r in range(procs)]
        for p in ps: p.start()
        for p in ps: p.join()
