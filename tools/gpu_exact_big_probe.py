import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import sdflib_amd as S
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
v, f = bumpy_icosphere(8); box = box_with_margin(v)
ctx = S.Context(0, use_torch_stream=True)
m = S.Mesh(v, f, ctx); m.build_bvh()
free0 = torch.cuda.mem_get_info(0)[0]
for depth in (7, 8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = S.ExactOctreeSdf(m, box, depth, 3, 128); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    i = e.info
    print(f"1.31 M triangles, ExactOctreeSdf depth {depth}: {dt*1e3:.1f} ms, nodes {i.num_nodes}, max leaf {i.max_triangles_in_leafs}, cached MB {ctx.cached_bytes() >> 20}, free dropped by {(free0 - torch.cuda.mem_get_info(0)[0]) >> 20} MB", flush=True)
    pts = random_points_in_box(box, 2000000, seed=3)
    d = e.get_distance(pts); 
    ids = m.nearest_triangle(pts[:200000])
    print("  queries ok", float(np.abs(d).max()))
    e.close()
