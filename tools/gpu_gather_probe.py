"""Random 256-byte-block gather (the query kernel's leaf access): per-lane loads vs cooperative loads (SDFHIP_GATHER_COOP=1)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdflib_amd as S
from sdflib_amd._lib import lib, check
ctx = S.Context(0, use_torch_stream=True)
for blocks in (300_000, 10_000_000):
    n = 10_000_000
    data = torch.arange(64 * blocks, dtype=torch.int32, device="cuda").remainder_(7)
    ids = torch.randint(0, blocks, (n,), device="cuda", dtype=torch.int64).to(torch.int32).contiguous()
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    fn = lambda: check(lib().sdfhip_test_gather_blocks(ctx.h, C.c_void_p(data.data_ptr()), C.c_void_p(ids.data_ptr()), n, C.c_void_p(out.data_ptr())))
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{blocks} blocks ({blocks * 256 / 1e6:.0f} MB), {n} lanes: {ms:.4f} ms = {n * 264 / ms / 1e9:.2f} TB/s algorithmic, checksum {float(out.double().sum()):.0f}", flush=True)
