import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
src = open("/root/repo/tools/gpu_fuzz.py").read().split("fails = 0; t00")[0]
exec(src)
seed = int(sys.argv[1])
# replay one() but keep the objects: patch the function to return locals
import types
code = src.split("def one(seed):")[1]
body = "def one_dbg(seed):" + code.replace('    return f"T={len(f)} cont', '    return locals()\n    return f"T={len(f)} cont')
body = body.replace('assert np.array_equal(bits(e0), bits(e1)) and np.array_equal(t0, t1.astype(np.uint32)), "exact queries"', 'return locals()')
exec(body)
L = one_dbg(seed)
e0, t0, e1, t1, pts, f, v = L["e0"], L["t0"], L["e1"], L["t1"], L["pts"], L["f"], L["v"]
t1 = t1.astype(np.uint32)
bad = np.flatnonzero((bits(e0) != bits(e1)) | (t0 != t1))
print("T", len(f), "exact depth/start/min", L["edepth"], L["estart"], L["mint"], "mismatches", len(bad), "of", len(pts))
print("idx", bad[:10]); print("oracle d", e0[bad[:10]], "tri", t0[bad[:10]]); print("gpu d", e1[bad[:10]], "tri", t1[bad[:10]])
print("faces oracle", f[t0[bad[:5]]], "faces gpu", f[t1[bad[:5]]])
# brute force over all triangles with the oracle mesh
om = L["om"]
print("oracle nearest (bvh)", om.nearest(pts[bad[:10]]))
# small-batch path (per-lane kernel) vs sorted path
ge = L["ge"]
d_small, t_small = ge.get_distance(pts[bad[:10]], triangle=True)
print("gpu per-lane kernel d", d_small, "tri", t_small)
gm = L["gm"]
tdm = gm.triangle_data(); tdo = om.triangle_data()
print("mesh td equal oracle:", np.array_equal(bits(tdm), bits(tdo)) or np.array_equal(tdm, tdo, equal_nan=True))
import ctypes as C
from sdflib_amd._lib import lib, check
tde = np.zeros_like(tdm)
check(lib().sdfhip_exact_triangle_data(ge.h, tde.ctypes.data_as(C.c_void_p)))
print("exact td equal mesh td:", np.array_equal(bits(tde), bits(tdm)) or np.array_equal(tde, tdm, equal_nan=True))
print("td row 3 mesh", tdm[3]); print("td row 3 exact", tde[3])
print("v", v, "f", f, "box", L["box"])
pv = gm.point_values(pts[bad[:10]], t0[bad[:10]]); print("gpu point_values d", pv[:, 0])
pvo = om.point_values(pts[bad[:10]], t0[bad[:10]]); print("oracle point_values d", pvo[:, 0])
print("pts", pts[bad[:4]])
