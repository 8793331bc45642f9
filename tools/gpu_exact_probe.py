"""Dev probe: ExactOctreeSdf build/query timings on the GPU."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import sdflib_amd as S
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin

s = int(sys.argv[1]) if len(sys.argv) > 1 else 7
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 7
start = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nq = int(float(sys.argv[4])) if len(sys.argv) > 4 else 10_000_000
v, f = bumpy_icosphere(s); box = box_with_margin(v)
ctx = S.Context(0, use_torch_stream=True)
m = S.Mesh(v, f, ctx)
for it in range(2):
    t = time.time(); ex = S.ExactOctreeSdf(m, box, depth, start, 128); torch.cuda.synchronize(); dt = time.time() - t
    i = ex.info
    print(f"exact build {dt:.3f}s nodes={i.num_nodes} sets={i.num_set_words} masks={i.num_mask_bytes} maxLeaf={i.max_triangles_in_leafs} maxEnc={i.max_triangles_encoded_in_leafs} cullTests={i.cull_tests}")
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
bb = ex.get_grid_bounding_box()
lo = torch.tensor(bb[:3], device=dev); size = float(bb[3] - bb[0])
pts = lo + torch.rand((nq, 3), generator=g, device=dev) * (size * 0.999999)
for grad in (False, True):
    ex.get_distance(pts, gradient=grad); torch.cuda.synchronize()
    t = time.time(); ex.get_distance(pts, gradient=grad); torch.cuda.synchronize(); dt = time.time() - t
    print(f"exact query grad={grad}: {dt*1e3:.2f} ms / {nq/1e6:.0f}M = {nq/dt/1e6:.1f} Mq/s")
