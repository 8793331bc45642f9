"""XCD-affine queries, measured without touching the kernel: workgroup b of a 1-D launch runs on XCD b % 8, so a point array
whose 256-point chunk b comes from bin b % 8 gives every XCD ONE eighth of the tree (bins: the 8 z-slabs of the start grid, or its 8 octants).
Times the C2 query kernel on 10 M uniform points (a) as they come, (b) binned + interleaved (XCD-affine), (c) binned, not interleaved, and the
data movement binning cannot avoid: the points gathered into bin order and the results scattered back (the permutation given for free)."""
import os, sys
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
bb = t.get_grid_bounding_box(); size = float(bb[3] - bb[0]); lo = torch.tensor(bb[:3], device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
pts = (lo + torch.rand((n, 3), generator=gen, device=dev) * (size * 0.999999)).contiguous()
out = torch.empty(n, dtype=torch.float32, device=dev)
def ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.mean(ts))
base = ms(lambda: t.get_distance(pts, out=out))
ref = out.clone()
print(f"C2 tree, {n} uniform points as they come: {base:.4f} ms")
half = ((pts - lo) / (size / 2)).floor().clamp(0, 1).to(torch.int64)
for name, key in (("8 z-slabs", ((pts[:, 2] - lo[2]) / (size / 8)).floor().clamp(0, 7).to(torch.int64)), ("8 octants", half[:, 0] + 2 * half[:, 1] + 4 * half[:, 2])):
    order = torch.sort(key, stable=True).indices
    counts = torch.bincount(key, minlength=8); starts = torch.cumsum(counts, 0) - counts
    mchunks = int(counts.min().item()) // 256
    idx = torch.stack([order[int(starts[x]):int(starts[x]) + 256 * mchunks].view(mchunks, 256) for x in range(8)], dim=1).reshape(-1)      # chunk b <- bin b % 8
    pa = pts[idx].contiguous(); na = pa.shape[0]
    oa = torch.empty(na, dtype=torch.float32, device=dev)
    affine = ms(lambda: t.get_distance(pa, out=oa))
    assert torch.equal(oa.view(torch.int32), ref[idx].view(torch.int32))
    ps = pts[order].contiguous()
    plain = ms(lambda: t.get_distance(ps, out=out))
    res = torch.empty(n, dtype=torch.float32, device=dev)
    def move():
        g = pts[idx]
        res[idx] = oa
    mv = ms(move)
    def keypass():            # the cheapest conceivable binning front end: read z, write an 8-bit key (the histogram / scan / scatter of indices come on top)
        k = ((pts[:, 2] - lo[2]) * (8.0 / size)).to(torch.uint8)
    kp = ms(keypass)
    print(f"{name}: XCD-affine {affine * n / na:.4f} ms per 10 M ({na} points: bins cut to equal length), binned but not affine {plain:.4f} ms; "
          f"gather points + scatter results {mv * n / na:.4f} ms, key pass alone {kp:.4f} ms -> total {affine * n / na + mv * n / na + kp:.4f} ms vs {base:.4f} ms as they come")
