"""Dev probe: distribution of the candidate search's work over the sample points of a build.  Reconstructs the deduplicated
mid-point samples of octree levels from a built tree (Morton order, as the sampler feeds them), runs the two-phase search with
per-query counters (sdfhip_mesh_nearest_stats) and prints, per level: expansions / triangle evaluations per query, their
quantiles, the share of all expansions spent in each bucket of |distance| (in cells of that level), and the same with every
query seeded with its own answer (sdfhip_mesh_nearest_stats_preseeded: the fewest visits ANY visiting order needs with this tree and these bounds).
Usage: python tools/gpu_near_hist.py [subdiv] [levels, e.g. 5,6,7,8]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import sdflib_amd as S
from sdflib_amd._lib import lib, check
from sdflib_amd import meshgen

s = int(sys.argv[1]) if len(sys.argv) > 1 else 7
levels = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "5,6,7,8").split(",")]
depth, start = 8, 3
v, f = (meshgen.torus_knot() if os.environ.get("PROBE_KNOT") else meshgen.bumpy_icosphere(s)); box = meshgen.box_with_margin(v)
m = S.Mesh(v, f); m.build_bvh()
tree = S.OctreeSdf(m, box, depth, start, 1e-3, num_threads=2)
words = tree.get_octree_data()
bb = tree.get_grid_bounding_box(); size = float(bb[3] - bb[0])
G = 2 ** start
rel = np.array([(a, b, c) for c in range(3) for b in range(3) for a in range(3) if (a == 1) + (b == 1) + (c == 1) >= 1], dtype=np.int64)
def spread(x):
    x = x.astype(np.uint64); r = np.zeros_like(x)
    for b in range(12): r |= ((x >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
    return r
def level_points(level):
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
    lat = (2 * co[:, None, :] + rel[None, :, :]).reshape(-1, 3)
    key = spread(lat[:, 0]) | (spread(lat[:, 1]) << np.uint64(1)) | (spread(lat[:, 2]) << np.uint64(2))
    key, first = np.unique(key, return_index=True)
    lat = lat[first]
    return len(idx), np.ascontiguousarray((bb[:3] + lat.astype(np.float32) * np.float32(size / 2 ** (level + 1))).astype(np.float32))
def run(pts, preseed=False):
    n = len(pts); out = np.zeros((n, 4), np.uint32)
    fn = lib().sdfhip_mesh_nearest_stats_preseeded if preseed else lib().sdfhip_mesh_nearest_stats
    for a in range(0, n, 3_000_000):
        b = min(n, a + 3_000_000)
        check(fn(m.h, pts[a:b].ctypes.data_as(C.c_void_p), b - a, out[a:b].ctypes.data_as(C.c_void_p)))
    return out
for level in levels:
    nodes, pts = level_points(level)
    dist = np.abs(tree.get_distance(pts))
    cell = size / 2 ** level
    print(f"== level {level}: {nodes} nodes, {len(pts)} unique mid-points, cell {cell:.5f}", flush=True)
    for mode in ("default", "preseed"):
        out = run(pts, preseed=(mode == "preseed"))
        ex, it, tr = out[:, 1].astype(np.int64), out[:, 2].astype(np.int64), out[:, 3].astype(np.int64)
        long = (it == 0).sum()
        q = np.percentile(ex, [10, 50, 90, 99, 99.9])
        print(f" [{mode}] expansions/query {ex.mean():.1f} (p10 {q[0]:.0f} p50 {q[1]:.0f} p90 {q[2]:.0f} p99 {q[3]:.0f} p99.9 {q[4]:.0f} max {ex.max()}), triangles {tr.mean():.1f}, iterations {it.mean():.1f}; {long} queries went to the long kernel", flush=True)
        edges = [0, 0.5, 1, 2, 4, 8, 16, 32, 1e9]
        for a, b in zip(edges[:-1], edges[1:]):
            sel = (dist >= a * cell) & (dist < b * cell)
            if sel.any():
                print(f"    |d| in [{a}, {b}) cells: {sel.mean()*100:5.1f} % of the points, {ex[sel].sum() / max(ex.sum(), 1) * 100:5.1f} % of the expansions, mean {ex[sel].mean():.0f} expansions {tr[sel].mean():.0f} triangles", flush=True)
