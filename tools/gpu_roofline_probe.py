"""GPU probe for the roofline denominators (run through gpurun):
 (a) the VALU issue ceiling (k_valu_peak, live), (b) the 256-B gather rate from a 200 MB (Infinity-Cache resident) array and from 2.56 GB (HBM),
 (c) depth-9 trees of growing size: words, build time, 12 M-query time -> which threshold gives an HBM-only working set below the 2^30-word format limit."""
import ctypes as C
import json
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdflib_amd as S
from sdflib_amd._lib import lib, check
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin

dev = torch.device("cuda", 0)
ctx = S.Context(0, use_torch_stream=True)
res = {}


def time_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), sum(ts) / len(ts)


# (a)
for blocks, iters in [(2048, 20000), (4096, 20000), (1024, 40000)]:
    out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    mn, av = time_ms(lambda: check(lib().sdfhip_test_valu_peak(ctx.h, blocks, iters, C.c_void_p(out.data_ptr()))))
    winst = blocks * 4 * 8 * iters
    res[f"valu_peak_{blocks}x{iters}"] = {"ms_min": mn, "ms_avg": av, "wave_fma_per_s": winst / mn * 1e3, "implied_clock_ghz_if_1024_simds_x4": winst / mn * 1e3 * 4 / 1024 / 1e9,
                                          "tflops": winst * 64 * 2 / mn * 1e3 / 1e12}
# (b)
n = 10_000_000
for name, nblocks in [("gather_200MB", 781_250), ("gather_128MB", 500_000), ("gather_64MB", 250_000), ("gather_2560MB", 10_000_000)]:
    data = torch.empty(64 * nblocks, dtype=torch.int32, device=dev).fill_(1)
    g = torch.Generator(device=dev); g.manual_seed(7)
    ids = torch.randint(0, nblocks, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32).contiguous()
    out = torch.empty(n, dtype=torch.float32, device=dev)
    mn, av = time_ms(lambda: check(lib().sdfhip_test_gather_blocks(ctx.h, C.c_void_p(data.data_ptr()), C.c_void_p(ids.data_ptr()), n, C.c_void_p(out.data_ptr()))), reps=10)
    res[name] = {"ms_min": mn, "ms_avg": av, "gb_s_avg": n * 264 / av / 1e6, "gb_s_min": n * 264 / mn / 1e6}
    del data, ids, out
print(json.dumps(res, indent=1), flush=True)
# (c)
v, f = bumpy_icosphere(7)
box = box_with_margin(v)
mesh = S.Mesh(v, f, ctx); mesh.build_bvh()
NQ = 12_000_000
for thr in [2e-4, 1e-4, 5e-5, 3e-5, 2e-5]:
    try:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t = S.OctreeSdf(mesh, box, 9, 3, thr, num_threads=2)
        torch.cuda.synchronize(); bs = time.perf_counter() - t0
    except Exception as e:
        res[f"d9_thr{thr}"] = {"error": str(e)[:300]}
        print(json.dumps({f"d9_thr{thr}": res[f"d9_thr{thr}"]}), flush=True)
        continue
    i = t.info
    g = torch.Generator(device=dev); g.manual_seed(4321)
    bb = t.get_grid_bounding_box(); size = float(bb[3] - bb[0])
    pts = (torch.tensor(bb[:3], device=dev) + torch.rand((NQ, 3), generator=g, device=dev) * (size * 0.999999)).contiguous()
    out = torch.empty(NQ, dtype=torch.float32, device=dev)
    mn, av = time_ms(lambda: t.get_distance(pts, eval_mode=S.EVAL_EXACT, out=out), reps=10)
    res[f"d9_thr{thr}"] = {"words": int(i.num_words), "gb": int(i.num_words) * 4 / 1e9, "leaves": int(i.num_leaves), "build_s": bs, "query_ms_avg": av, "query_ms_min": mn,
                           "leaves_per_depth": list(i.leaves_per_depth)[:10]}
    print(json.dumps({f"d9_thr{thr}": res[f"d9_thr{thr}"]}), flush=True)
    t.close(); del pts, out
json.dump(res, open("gpurun_out/roofline_probe.json", "w"), indent=1)
