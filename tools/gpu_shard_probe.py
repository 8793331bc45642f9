"""How long one rank's share of the sharded NO_CONTINUITY build takes on its own (one GPU, the other ranks absent): world 1, 2, 4, 8;
every rank of the world in turn, the slowest and the mean (the build's wall time on N GPUs is the slowest shard + the exchange)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdflib_amd as S
from sdflib_amd import api, distributed as D
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
sub = int(os.environ.get("PROBE_SUBDIV", "7"))
v, f = bumpy_icosphere(sub); box = box_with_margin(v)
ctx = S.Context(0, use_torch_stream=True)
m = S.Mesh(v, f, ctx); m.build_bvh()
for world in (1, 2, 4, 8):
    ranges = D.partition_cells(8 ** 3, world, D.cell_weights(m.vertices, box, 3))
    ts = []
    for r in range(world):
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sh = api.OctreeShard(m, box, 8, 3, 1e-3, cells=ranges[r]); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0); sh.close()
        ts.append(best)
    print(f"{len(f)} triangles, world {world}: slowest shard {1e3 * max(ts):.2f} ms, mean {1e3 * np.mean(ts):.2f} ms, sum {1e3 * sum(ts):.1f} ms", flush=True)
