"""10 M uniform-random getDistance() on the C2 tree (and the depth-9 HBM-resident tree with PROBE_DEEP=1): kernel time per launch.
SDFHIP_QUERY_LANE_LOADS=1 selects the kernel in which every lane fetches its own block."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdflib_amd as S
from sdflib_amd import meshgen
v, f = meshgen.bumpy_icosphere(7)
box = meshgen.box_with_margin(v)
ctx = S.Context(0, use_torch_stream=True)
mesh = S.Mesh(v, f, ctx)
deep = bool(os.environ.get("PROBE_DEEP"))
tree = S.OctreeSdf(mesh, box, 9 if deep else 8, 3, 2e-4 if deep else 1e-3, num_threads=2)
n = 12_000_000 if deep else 10_000_000
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
bb = tree.get_grid_bounding_box(); size = float(bb[3] - bb[0])
pts = (torch.tensor(bb[:3], device="cuda") + torch.rand((n, 3), generator=gen, device="cuda") * (size * 0.999999)).contiguous()
out = torch.empty(n, dtype=torch.float32, device="cuda"); outg = torch.empty((n, 3), dtype=torch.float32, device="cuda")
for name, fn in (("value exact", lambda: tree.get_distance(pts, eval_mode=S.EVAL_EXACT, out=out)),
                 ("value fast", lambda: tree.get_distance(pts, eval_mode=S.EVAL_FAST, out=out)),
                 ("value+grad fast", lambda: tree.get_distance(pts, gradient=True, eval_mode=S.EVAL_FAST, out=out, out_grad=outg)),
                 ("value+grad exact", lambda: tree.get_distance(pts, gradient=True, eval_mode=S.EVAL_EXACT, out=out, out_grad=outg))):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"{'deep' if deep else 'C2'} {name}: {ms:.4f} ms = {n / ms / 1e6:.2f} G q/s, crc {zlib.crc32(out.cpu().numpy().tobytes()):08x}", flush=True)
