"""Dev bench of the nearest-triangle search alone: the deduplicated mid-point samples of one octree level (Morton order, as the
sampler feeds them) and 2 M uniform points, device resident, timed with synchronised wall clock; prints a checksum of the ids so
that runs with SDFHIP_NEAREST=exact / two-phase can be compared.  Usage: python tools/gpu_nearest_bench.py [subdiv] [level]"""
import sys, os, time, ctypes as C, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import sdflib_amd as S
from sdflib_amd._lib import lib, check
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box

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
rel = np.array([(a, b, c) for c in range(3) for b in range(3) for a in range(3) if (a == 1) + (b == 1) + (c == 1) >= 1], dtype=np.int64)
lat = (2 * co[:, None, :] + rel[None, :, :]).reshape(-1, 3)
def spread(x):
    x = x.astype(np.uint64); r = np.zeros_like(x)
    for b in range(12): r |= ((x >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
    return r
key = spread(lat[:, 0]) | (spread(lat[:, 1]) << np.uint64(1)) | (spread(lat[:, 2]) << np.uint64(2))
key, first = np.unique(key, return_index=True)
lat = lat[first]
lvl = np.ascontiguousarray((bb[:3] + lat.astype(np.float32) * np.float32(size / 2 ** (level + 1))).astype(np.float32))
uni = random_points_in_box(box, 2_000_000, seed=3)
mode = os.environ.get("SDFHIP_NEAREST", "two-phase")
for name, pts in ((f"level-{level} samples", lvl), ("uniform", uni)):
    n = len(pts)
    tp = torch.from_numpy(pts).cuda(); to = torch.empty(n, dtype=torch.int32, device="cuda")
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t = time.time()
        check(lib().sdfhip_mesh_nearest(m.h, C.c_void_p(tp.data_ptr()), n, C.c_void_p(to.data_ptr()), 1))
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    ids = to.cpu().numpy()
    print(f"[{mode}] {name}: {n} points, {best*1e3:.2f} ms = {n/best/1e6:.1f} M/s, crc {zlib.crc32(ids.tobytes()):08x}", flush=True)
if os.environ.get("PROBE_PREFIX"):
    # T(n) of the search on contiguous pieces of the level's Morton-ordered samples (same difficulty per query): a fixed cost per launch shows
    # as an intercept.  Slices from the MIDDLE of the order, k pieces each (the mean over pieces is printed).
    tp = torch.from_numpy(lvl).cuda(); N = len(lvl)
    for n in (4096, 16384, 65536, 131072, 262144, 524287, 524288, 1048576, N):
        n = min(n, N); ts = []
        for piece in range(4):
            off = ((N - n) * piece // 3) // 128 * 128 if n < N else 0
            sub = tp[off:off + n].contiguous(); to = torch.empty(n, dtype=torch.int32, device="cuda")
            best = 1e9
            for _ in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); a.record()
                check(lib().sdfhip_mesh_nearest(m.h, C.c_void_p(sub.data_ptr()), n, C.c_void_p(to.data_ptr()), 1))
                b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
            ts.append(best)
        print(f"prefix n={n:8d}: {np.mean(ts):.3f} ms (min {min(ts):.3f}, max {max(ts):.3f}) = {n/np.mean(ts)/1e3:.1f} M/s", flush=True)
