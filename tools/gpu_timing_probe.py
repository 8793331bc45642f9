"""Dev probe: timings of mesh prep / BVH / build / query on the GPU for a given mesh size."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import sdflib_amd as S
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin

s = int(sys.argv[1]) if len(sys.argv) > 1 else 7
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
start = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nq = int(float(sys.argv[4])) if len(sys.argv) > 4 else 10_000_000
t = time.time(); v, f = bumpy_icosphere(s); box = box_with_margin(v); print(f"mesh s={s} T={len(f)} gen {time.time()-t:.2f}s")
ctx = S.Context(0, use_torch_stream=True)
t = time.time(); m = S.Mesh(v, f, ctx); print(f"mesh upload+triangle data {time.time()-t:.3f}s")
print(f"bvh build (host planner + upload) {m.build_bvh():.3f}s")
for it in range(2):
    t = time.time(); oc = S.OctreeSdf(m, box, depth, start, 1e-3, num_threads=2); dt = time.time() - t
    i = oc.info
    print(f"octree build {dt:.3f}s (samples {i.seconds_samples:.3f}s decide {i.seconds_decide:.3f}s) words={i.num_words} ({i.num_words*4/1e6:.1f} MB) leaves={i.num_leaves} nodes={i.num_nodes} samples={i.num_samples}")
    if it == 0: del oc
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
bb = oc.get_grid_bounding_box()
lo = torch.tensor(bb[:3], device=dev); size = float(bb[3] - bb[0])
pts = lo + torch.rand((nq, 3), generator=g, device=dev) * (size * 0.999999)
out = torch.empty(nq, device=dev); outg = torch.empty((nq, 3), device=dev)
for mode, name in ((S.EVAL_EXACT, "exact"), (S.EVAL_FAST, "fast")):
    for grad in (False, True):
        for _ in range(2): oc.get_distance(pts, gradient=grad, eval_mode=mode, out=out, out_grad=outg)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): oc.get_distance(pts, gradient=grad, eval_mode=mode, out=out, out_grad=outg)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"query {name} grad={grad}: {ms:.3f} ms / {nq/1e6:.0f}M = {nq/ms/1e3:.1f} Mq/s  ({nq*292/ms/1e6:.1f} GB/s @292B)")
t = time.time(); d = oc.get_distance_grid(bb[:3] + size / 512, np.full(3, size / 256, np.float32), (256, 256, 256), gradient=True, eval_mode=S.EVAL_FAST, device_out=True); torch.cuda.synchronize(); print(f"grid256 first {time.time()-t:.4f}s")
t = time.time(); d = oc.get_distance_grid(bb[:3] + size / 512, np.full(3, size / 256, np.float32), (256, 256, 256), gradient=True, eval_mode=S.EVAL_FAST, device_out=True); torch.cuda.synchronize(); print(f"grid256 second {time.time()-t:.4f}s")
