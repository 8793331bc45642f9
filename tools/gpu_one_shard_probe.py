"""Four builds of ONE shard (world PROBE_WORLD, rank PROBE_RANK) for kernel traces: tools/trace_build.sh PROBE_SCRIPT=gpu_one_shard_probe.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdflib_amd as S
from sdflib_amd import api, distributed as D
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
v, f = bumpy_icosphere(int(os.environ.get("PROBE_SUBDIV", "7"))); box = box_with_margin(v)
world, rank = int(os.environ.get("PROBE_WORLD", "8")), int(os.environ.get("PROBE_RANK", "3"))
ctx = S.Context(0, use_torch_stream=True)
m = S.Mesh(v, f, ctx); m.build_bvh()
ranges = D.partition_cells(8 ** 3, world, D.cell_weights(m.vertices, box, 3))
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sh = api.OctreeShard(m, box, 8, 3, 1e-3, cells=ranges[rank]); torch.cuda.synchronize()
    print(f"build {time.perf_counter() - t0:.4f} s, cells {ranges[rank]}, traversals {sh.info.num_traversals}", flush=True); sh.close()
