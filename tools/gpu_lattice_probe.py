"""256^3 lattice (value + gradient, EVAL_FAST) on the C2 tree: time per call; SDFHIP_LATTICE_POINTS=1 selects the point kernel."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdflib_amd as S
from sdflib_amd import meshgen

sub = int(os.environ.get("PROBE_SUBDIV", "7")); depth = int(os.environ.get("PROBE_DEPTH", "8"))
v, f = meshgen.bumpy_icosphere(sub)
mesh = S.Mesh(v, f)
box = meshgen.box_with_margin(v)
alg = S.ALG_CONTINUITY if os.environ.get("PROBE_CONT") else S.ALG_NO_CONTINUITY
tree = S.OctreeSdf(mesh, box, depth, 3, 1e-3, init_algorithm=alg, num_threads=2)
print("words", tree.info.num_words, flush=True)
bb = tree.get_grid_bounding_box(); size = float(bb[3] - bb[0])
for n in [int(t) for t in os.environ.get('PROBE_N', '256,200,512').split(',')]:
    step = np.full(3, size / n, dtype=np.float32); origin = (bb[:3] + 0.5 * step).astype(np.float32)
    for grad in (True, False):
        fn = lambda: tree.get_distance_grid(origin, step, (n, n, n), gradient=grad, eval_mode=(S.EVAL_EXACT if os.environ.get('PROBE_EXACT') else S.EVAL_FAST), device_out=True)
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): fn()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 100)
        print(f"n={n} grad={grad}: first call {first:.3f} ms, steady {best:.4f} ms = {n ** 3 / best / 1e6:.1f} G points/s, {(n ** 3 * (16 if grad else 4)) / best / 1e9:.2f} TB/s written", flush=True)

if os.environ.get('PROBE_DIST'):
    # distribution of single-launch times (events) and the write ceiling of the box (fill of the same number of bytes)
    n = 256
    step = np.full(3, size / n, dtype=np.float32); origin = (bb[:3] + 0.5 * step).astype(np.float32)
    fn = lambda: tree.get_distance_grid(origin, step, (n, n, n), gradient=True, eval_mode=S.EVAL_FAST, device_out=True)
    ts = []
    for _ in range(60):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts = np.sort(np.array(ts))
    print("single launches (events, us): min %.1f p25 %.1f median %.1f p75 %.1f max %.1f" % (ts[0], ts[15], ts[30], ts[45], ts[-1]), flush=True)
    for mb in (268, 2147):
        buf = torch.empty(mb * 1000 * 1000 // 4, dtype=torch.float32, device="cuda")
        for _ in range(3): buf.fill_(1.0)
        tt = []
        for _ in range(20):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); buf.fill_(2.0); e1.record(); torch.cuda.synchronize(); tt.append(e0.elapsed_time(e1) * 1e3)
        tt = np.sort(np.array(tt))
        print(f"fill {mb} MB: min {tt[0]:.1f} us median {tt[10]:.1f} us = {mb / tt[10] * 1e-6 * 1e6:.2f} TB/s", flush=True)
        del buf

