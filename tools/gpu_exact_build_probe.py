"""Four ExactOctreeSdf builds of the C3 configuration (depth 7, start 3, min 128) for kernel traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdflib_amd as S
from sdflib_amd import meshgen
v, f = meshgen.bumpy_icosphere(7)
box = meshgen.box_with_margin(v)
mesh = S.Mesh(v, f)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ex = S.ExactOctreeSdf(mesh, box, 7, 3, 128)
    torch.cuda.synchronize(); print(f"build {time.perf_counter() - t0:.4f} s, nodes {ex.info.num_nodes}", flush=True)
    ex.close()
