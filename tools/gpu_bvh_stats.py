"""Dev probe: BVH traversal statistics for sample-like points (distance to the surface ~ an octree leaf)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import sdflib_amd as S
from sdflib_amd._lib import lib, check
from sdflib_amd.meshgen import bumpy_icosphere

s = int(sys.argv[1]) if len(sys.argv) > 1 else 7
v, f = bumpy_icosphere(s)
m = S.Mesh(v, f)
rng = np.random.default_rng(0)
n = 200000
base = v[rng.integers(0, len(v), n)]
for scale in (0.005, 0.02, 0.08, 0.3):
    pts = (base * (1.0 + rng.normal(0, scale, (n, 1)))).astype(np.float32)
    out = np.zeros((n, 4), np.uint32)
    check(lib().sdfhip_mesh_nearest_stats(m.h, pts.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)))
    print(f"T={len(f)} offset~{scale}: inner {out[:,1].mean():.1f} (max {out[:,1].max()}) wave iterations {out[:,2].mean():.1f} tris {out[:,3].mean():.1f} (max {out[:,3].max()})")
