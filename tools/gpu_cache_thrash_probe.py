import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import sdflib_amd as S
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
v, f = bumpy_icosphere(7); box = box_with_margin(v)
ctx = S.Context(0, use_torch_stream=True)
m = S.Mesh(v, f, ctx); m.build_bvh()
t = S.OctreeSdf(m, box, 8, 3, 1e-3, num_threads=2); torch.cuda.synchronize()
print("after octree build: cached MB", ctx.cached_bytes() >> 20)
if os.environ.get("DL"): w = t.get_octree_data(); print("after download: cached MB", ctx.cached_bytes() >> 20)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = S.ExactOctreeSdf(m, box, 7, 3, 128); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"exact build {i}: {dt*1e3:.1f} ms, cached MB {ctx.cached_bytes() >> 20}")
    e.close()
