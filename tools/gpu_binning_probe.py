"""Measurement asked for by the round-3 review: would binning a >= 1 M-query batch by start cell (so that an XCD's L2 keeps its cells'
leaves) pay?  Times the C2 query kernel on 10 M uniform points (a) as they come, (b) pre-sorted by start cell (9 bits), (c) pre-sorted by
the leaf they end in (the most locality any binning could give), and the data movement a binning pass cannot avoid (gathering the 12-byte
points into sorted order + scattering the 4-byte results back, permutation given for free)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import sdflib_amd as S
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
dev = torch.device("cuda", 0)
v, f = bumpy_icosphere(7); box = box_with_margin(v)
ctx = S.Context(0, use_torch_stream=True)
m = S.Mesh(v, f, ctx); m.build_bvh()
t = S.OctreeSdf(m, box, 8, 3, 1e-3, num_threads=2)
n = 10_000_000
bb = t.get_grid_bounding_box(); size = float(bb[3] - bb[0])
gen = torch.Generator(device=dev); gen.manual_seed(1234)
pts = (torch.tensor(bb[:3], device=dev) + torch.rand((n, 3), generator=gen, device=dev) * (size * 0.999999)).contiguous()
out = torch.empty(n, dtype=torch.float32, device=dev)
def ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
q = lambda p: (lambda: t.get_distance(p, out=out))
base = ms(q(pts))
cell = ((pts - torch.tensor(bb[:3], device=dev)) / (size / 8)).floor().clamp(0, 7).to(torch.int64)
key = (cell[:, 2] * 8 + cell[:, 1]) * 8 + cell[:, 0]
perm = torch.argsort(key)
ps = pts[perm].contiguous()
binned = ms(q(ps))
fine = ((pts - torch.tensor(bb[:3], device=dev)) / (size / 256)).floor().clamp(0, 255).to(torch.int64)
def spread(x):
    r = torch.zeros_like(x)
    for b in range(8): r |= ((x >> b) & 1) << (3 * b)
    return r
mkey = spread(fine[:, 0]) | (spread(fine[:, 1]) << 1) | (spread(fine[:, 2]) << 2)
pm = pts[torch.argsort(mkey)].contiguous()
morton = ms(q(pm))
res = torch.empty(n, dtype=torch.float32, device=dev)
def move():
    g = pts[perm]
    res[perm] = out
move_ms = ms(move)
print(f"C2 tree, {n} uniform points: query kernel {base:.4f} ms as they come, {binned:.4f} ms pre-sorted by start cell (9 bits), {morton:.4f} ms pre-sorted by depth-8 Morton cell")
print(f"gather of the points into sorted order + scatter of the results back (permutation given): {move_ms:.4f} ms -> binning pays only if {base:.4f} - {binned:.4f} = {base - binned:.4f} ms > {move_ms:.4f} ms + the histogram / scan / key pass")
