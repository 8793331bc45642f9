"""Dev probe (CPU; the oracle only generates the sample points and their true distances): the FLOOR of a nearest-triangle search on
search-only structures, order-free — a node must be opened by any correct traversal iff its lower bound does not exceed the true
distance, so   floor = #{nodes : lb(node, q) <= d(q)}   whatever the traversal order and however good the seed.

Structures: implicit W-wide AABB trees over the triangles sorted by the Morton code of their centroids, leaves of L consecutive
triangles, every inner node the union of W consecutive children (what a device LBVH would be, minus treelet optimisation).
Printed per distance band: inner nodes opened (each costs W box tests), leaves opened (each costs L triangle tests), and the same
two figures for what phase 1 walks today (4-wide nodes of the reference's sphere tree, profiles/r05a_near_hist_by_distance.txt).
Usage: python tools/sim_search_structures.py [subdiv] [depth] [level] [sample]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import pyoracle as O
from sdflib_amd import meshgen

s = int(sys.argv[1]) if len(sys.argv) > 1 else 7
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
level = int(sys.argv[3]) if len(sys.argv) > 3 else depth - 1
nsample = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
start = 3
v, f = meshgen.bumpy_icosphere(s); box = meshgen.box_with_margin(v)
m = O.Mesh(v, f); m.build_bvh()
tree = O.Octree(m, box, depth, start, 1e-3)
words = tree.data(); bb = tree.box; size = float(bb[3] - bb[0])
G0 = 2 ** start
rel = np.array([(a, b, c) for c in range(3) for b in range(3) for a in range(3) if (a == 1) + (b == 1) + (c == 1) >= 1], dtype=np.int64)
def spread(x):
    x = x.astype(np.uint64); r = np.zeros_like(x)
    for b in range(12): r |= ((x >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
    return r
def level_points(level):
    k, j, i = np.meshgrid(np.arange(G0), np.arange(G0), np.arange(G0), indexing="ij")
    idx = (k * G0 * G0 + j * G0 + i).reshape(-1).astype(np.int64)
    co = np.stack([i.reshape(-1), j.reshape(-1), k.reshape(-1)], 1).astype(np.int64)
    for d in range(start, level):
        w = words[idx]; inner = (w >> 31) == 0
        base = (w[inner] & 0x3FFFFFFF).astype(np.int64); c = np.arange(8)
        idx = (base[:, None] + c[None, :]).reshape(-1)
        off = np.stack([c & 1, (c >> 1) & 1, (c >> 2) & 1], 1)
        co = (2 * co[inner][:, None, :] + off[None, :, :]).reshape(-1, 3)
    lat = (2 * co[:, None, :] + rel[None, :, :]).reshape(-1, 3)
    key = spread(lat[:, 0]) | (spread(lat[:, 1]) << np.uint64(1)) | (spread(lat[:, 2]) << np.uint64(2))
    key, first = np.unique(key, return_index=True)
    return np.ascontiguousarray((bb[:3] + lat[first].astype(np.float32) * np.float32(size / 2 ** (level + 1))).astype(np.float32))

pts = level_points(level)
rng = np.random.default_rng(1)
pts = pts[rng.choice(len(pts), nsample, replace=False)]
ids, dmin = m.nearest(pts, with_dist=True)
d = np.abs(dmin).astype(np.float64); P = pts.astype(np.float64)
cell = size / 2 ** level
T = len(f)
A, B, C = (v[f[:, k]].astype(np.float64) for k in range(3))
cen = (A + B + C) / 3
lo0, hi0 = np.minimum(np.minimum(A, B), C), np.maximum(np.maximum(A, B), C)
q = np.clip(((cen - bb[:3]) / size * 1024).astype(np.int64), 0, 1023)
order = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2)), kind="stable")
lo0, hi0 = lo0[order], hi0[order]

def box_dist(P, lo, hi):          # [nq, nb]
    out = np.empty((len(P), len(lo)))
    for i0 in range(0, len(P), 64):
        p = P[i0:i0 + 64, None, :]
        dd = np.maximum(np.maximum(lo[None] - p, p - hi[None]), 0.0)
        out[i0:i0 + 64] = np.sqrt((dd * dd).sum(2))
    return out

def group(lo, hi, w):
    n = (len(lo) + w - 1) // w; pad = n * w - len(lo)
    if pad: lo = np.concatenate([lo, np.repeat(lo[-1:], pad, 0)]); hi = np.concatenate([hi, np.repeat(hi[-1:], pad, 0)])
    return lo.reshape(n, w, 3).min(1), hi.reshape(n, w, 3).max(1)

bands = [(0, 0.5), (0.5, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 1e9)]
print(f"T = {T}, level {level}: {nsample} of the unique mid-points, cell {cell:.5f}")
for W, L in ((8, 4), (8, 2), (4, 4), (4, 1), (16, 8)):
    lo, hi = group(lo0, hi0, L)
    leaf_open = (box_dist(P, lo, hi) <= d[:, None]).sum(1)
    inner_open = np.zeros(nsample); nodes = 0
    while len(lo) > 1:
        lo, hi = group(lo, hi, W); nodes += len(lo)
        inner_open += (box_dist(P, lo, hi) <= d[:, None]).sum(1)
    print(f"== {W}-wide AABB tree over Morton order, leaves of {L}: {nodes} inner nodes")
    print(f"   all points: inner nodes opened {inner_open.mean():.1f} (= {W * inner_open.mean():.0f} box tests), leaves opened {leaf_open.mean():.1f} (= {L * leaf_open.mean():.0f} triangle tests)")
    for a, b in bands:
        sel = (d >= a * cell) & (d < b * cell)
        if sel.sum() >= 5: print(f"   |d| in [{a}, {b}) cells: {100 * sel.mean():5.1f} % of the points, inner {inner_open[sel].mean():6.1f}, leaves {leaf_open[sel].mean():6.1f}")
