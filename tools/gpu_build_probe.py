"""Three NO_CONTINUITY builds of the C2 configuration (for kernel traces: tools/trace_build.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdflib_amd as S
from sdflib_amd import meshgen
gen = meshgen.torus_knot if os.environ.get("PROBE_KNOT") else (lambda: meshgen.bumpy_icosphere(int(os.environ.get("PROBE_SUBDIV", "7"))))
v, f = gen()
box = meshgen.box_with_margin(v)
mesh = S.Mesh(v, f); mesh.build_bvh()
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tree = S.OctreeSdf(mesh, box, 8, 3, 1e-3, num_threads=2)
    torch.cuda.synchronize(); print(f"build {time.perf_counter() - t0:.4f} s, words {tree.info.num_words}, traversals {tree.info.num_traversals}", flush=True)
    del tree
