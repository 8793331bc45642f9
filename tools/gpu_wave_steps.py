"""Dev probe: where does the BVH search time of a build go?  Reconstructs the mid-point samples of one octree level from a built
tree (Morton order, deduplicated, as the sampler feeds them), runs the statistics kernel and reports per distance-to-surface
bucket: points, mean node visits per lane, and the share of WAVE loop iterations (a wave runs until its slowest lane is done)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import sdflib_amd as S
from sdflib_amd._lib import lib, check
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin

s = int(sys.argv[1]) if len(sys.argv) > 1 else 7
level = int(sys.argv[2]) if len(sys.argv) > 2 else 7
depth, start = 8, 3
v, f = bumpy_icosphere(s); box = box_with_margin(v)
m = S.Mesh(v, f); m.build_bvh()
tree = S.OctreeSdf(m, box, depth, start, 1e-3, num_threads=2)
words = tree.get_octree_data()
bb = tree.get_grid_bounding_box(); size = float(bb[3] - bb[0])
G = 2 ** start
k, j, i = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij")
idx = (k * G * G + j * G + i).reshape(-1).astype(np.int64)
co = np.stack([i.reshape(-1), j.reshape(-1), k.reshape(-1)], 1).astype(np.int64)
for d in range(start, level):
    w = words[idx]
    inner = (w >> 31) == 0
    base = (w[inner] & 0x3FFFFFFF).astype(np.int64)
    c = np.arange(8)
    idx = (base[:, None] + c[None, :]).reshape(-1)
    off = np.stack([c & 1, (c >> 1) & 1, (c >> 2) & 1], 1)
    co = (2 * co[inner][:, None, :] + off[None, :, :]).reshape(-1, 3)
print(f"level {level}: {len(idx)} nodes")
rel = np.array([(a, b, c) for c in range(3) for b in range(3) for a in range(3) if (a == 1) + (b == 1) + (c == 1) >= 1], dtype=np.int64)
lat = (2 * co[:, None, :] + rel[None, :, :]).reshape(-1, 3)
def spread(x):
    x = x.astype(np.uint64); r = np.zeros_like(x)
    for b in range(12): r |= ((x >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
    return r
key = spread(lat[:, 0]) | (spread(lat[:, 1]) << np.uint64(1)) | (spread(lat[:, 2]) << np.uint64(2))
key, first = np.unique(key, return_index=True)
lat = lat[first]
pts = np.ascontiguousarray((bb[:3] + lat.astype(np.float32) * np.float32(size / 2 ** (level + 1))).astype(np.float32))
n = len(pts) // 128 * 128; pts = pts[:n]
dist = np.abs(tree.get_distance(pts))
out = np.zeros((n, 4), np.uint32)
check(lib().sdfhip_mesh_nearest_stats(m.h, pts.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)))
enters, iters, tris = out[:, 1].astype(np.int64), out[:, 2].astype(np.int64), out[:, 3].astype(np.int64)
w_it = iters.reshape(-1, 64).max(1)
w_d = dist.reshape(-1, 64).mean(1)
print(f"{n} unique points; per lane: enters {enters.mean():.1f} tris {tris.mean():.1f}; wave iterations mean {w_it.mean():.1f} max {w_it.max()}; lane utilisation {(enters + tris).sum() / (w_it.sum() * 64.0):.2f}")
cell = size / 2 ** level
edges = [0, 1, 2, 4, 8, 16, 1e9]
for a, b in zip(edges[:-1], edges[1:]):
    sel = (w_d >= a * cell) & (w_d < b * cell)
    if sel.any():
        print(f"  waves with mean |d| in [{a}, {b}) cells: {sel.mean()*100:5.1f} % of the waves, {w_it[sel].sum() / w_it.sum() * 100:5.1f} % of the wave iterations, mean {w_it[sel].mean():.0f} iterations")
tp = torch.from_numpy(pts).cuda(); to = torch.empty(n, dtype=torch.int32, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t = time.time()
    check(lib().sdfhip_mesh_nearest(m.h, C.c_void_p(tp.data_ptr()), n, C.c_void_p(to.data_ptr()), 1))
    torch.cuda.synchronize(); dt = time.time() - t
print(f"k_nearest on these points: {dt*1e3:.2f} ms = {n/dt/1e6:.1f} M traversals/s, {w_it.sum()/dt/1e9:.2f} G wave-iterations/s")
