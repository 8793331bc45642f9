"""One rank's share (default: the slowest 1/8) of the sharded NO_CONTINUITY build, three times in a row: the trace_gaps.sh subject for the
question "where do a small shard's milliseconds go" (PROBE_SUBDIV=7|8, PROBE_WORLD, PROBE_RANK=-1 picks the slowest)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdflib_amd as S
from sdflib_amd import api, distributed as D
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
sub = int(os.environ.get("PROBE_SUBDIV", "7")); world = int(os.environ.get("PROBE_WORLD", "8")); rank = int(os.environ.get("PROBE_RANK", "-1"))
v, f = bumpy_icosphere(sub); box = box_with_margin(v)
ctx = S.Context(0, use_torch_stream=True)
m = S.Mesh(v, f, ctx); m.build_bvh()
ranges = D.partition_cells(8 ** 3, world, D.cell_weights(m.vertices, box, 3))
def one(r):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sh = api.OctreeShard(m, box, 8, 3, 1e-3, cells=ranges[r]); torch.cuda.synchronize()
    dt = time.perf_counter() - t0; sh.close(); return dt
if rank < 0:
    ts = [min(one(r) for _ in range(2)) for r in range(world)]
    rank = int(np.argmax(ts))
    print("shards ms:", [round(1e3 * t, 2) for t in ts], "-> rank", rank, flush=True)
time.sleep(0.01)
torch.cuda.synchronize()
for rep in range(3):
    time.sleep(0.005)
    print(f"build {one(rank):.4f} s, rank {rank} of {world}, cells {ranges[rank]}", flush=True)
