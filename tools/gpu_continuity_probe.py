"""CONTINUITY build of the C2 configuration with the phase times (SDFHIP_TIMING=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not os.environ.get("PROBE_QUIET"): os.environ.setdefault("SDFHIP_TIMING", "1")
import torch
import sdflib_amd as S
from sdflib_amd import meshgen
v, f = meshgen.bumpy_icosphere(int(os.environ.get("PROBE_SUBDIV", "7")))
box = meshgen.box_with_margin(v)
mesh = S.Mesh(v, f); mesh.build_bvh()
for rep in range(3):
    time.sleep(0.01)          # (separates the builds for tools/trace_gaps.sh: a burst = launches without a 2 ms gap)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tree = S.OctreeSdf(mesh, box, 8, 3, 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2)
    torch.cuda.synchronize(); print(f"build {time.perf_counter() - t0:.4f} s, words {tree.info.num_words}", flush=True)
    del tree
