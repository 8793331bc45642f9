"""Per-call cost of the host scalar entry of both structures (pre-converted ctypes arguments, several points)."""
import ctypes as C, time, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdflib_amd as S
from sdflib_amd import meshgen
from sdflib_amd._lib import lib

sub = int(os.environ.get('PROBE_SUBDIV', '5')); depth = int(os.environ.get('PROBE_DEPTH', '6'))
v, f = meshgen.bumpy_icosphere(sub)
mesh = S.Mesh(v, f)
box = meshgen.box_with_margin(v)
tree = S.OctreeSdf(mesh, box, depth, 3, 1e-3, num_threads=2)
ex = S.ExactOctreeSdf(mesh, box, 5, 3, 32)
L = lib()
rng = np.random.default_rng(1)
for label, pts in (("inside box", rng.uniform(-0.5, 0.5, (64, 3)).astype(np.float32)), ("outside box", rng.uniform(3, 4, (64, 3)).astype(np.float32))):
    d = np.zeros(1, np.float32)
    dp = C.c_void_p(d.ctypes.data)
    ptrs = [C.c_void_p(pts[i].ctypes.data) for i in range(len(pts))]
    for name, call in (("octree exact-eval", lambda p: L.sdfhip_octree_query(tree.h, p, 1, dp, None, 0, S.EVAL_EXACT)),
                       ("octree fast-eval", lambda p: L.sdfhip_octree_query(tree.h, p, 1, dp, None, 0, S.EVAL_FAST)),
                       ("exact", lambda p: L.sdfhip_exact_query(ex.h, p, 1, dp, None, None, 0))):
        call(ptrs[0])
        t0 = time.perf_counter()
        for r in range(200):
            for p in ptrs: call(p)
        print(f"{label:12s} {name:18s} {(time.perf_counter() - t0) / (200 * len(ptrs)) * 1e6:7.2f} us/call", flush=True)

# one fixed point, the bench's way (arguments converted on every call)
p1 = pts[:1].copy(); d1 = np.empty(1, np.float32)
for name, call in (("octree", lambda: L.sdfhip_octree_query(tree.h, p1.ctypes.data_as(C.c_void_p), 1, d1.ctypes.data_as(C.c_void_p), None, 0, S.EVAL_EXACT)),
                   ("exact", lambda: L.sdfhip_exact_query(ex.h, p1.ctypes.data_as(C.c_void_p), 1, d1.ctypes.data_as(C.c_void_p), None, None, 0))):
    for rep in range(3):
        call(); t0 = time.perf_counter()
        for _ in range(3000): call()
        print(f"fixed point, {name} (round {rep}): {(time.perf_counter() - t0) / 3000 * 1e6:.2f} us/call  point {p1[0]} -> {d1[0]}", flush=True)
