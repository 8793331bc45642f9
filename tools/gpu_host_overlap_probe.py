import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/sdflib_amd') else '.')
import numpy as np, torch
import sdflib_amd as S
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
v, f = bumpy_icosphere(7); box = box_with_margin(v)
ctx = S.Context(0); m = S.Mesh(v, f, ctx); m.build_bvh()
t = S.OctreeSdf(m, box, 8, 3, 1e-3, num_threads=2)
n = 10_000_000
pts = random_points_in_box(box, n, seed=3); hd = np.empty(n, np.float32); hg = np.empty((n, 3), np.float32)
def best(fn, reps=30):
    fn(); fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, float(np.median(ts)) * 1e3
print("overlap", os.environ.get("SDFHIP_HOST_OVERLAP", "on"), "uploaders", os.environ.get("SDFHIP_HOST_UPLOADERS", "default"),
      "value %.3f (median %.3f) ms" % best(lambda: t.get_distance(pts, out=hd)), "value+grad %.3f (median %.3f) ms" % best(lambda: t.get_distance(pts, gradient=True, out=hd, out_grad=hg)))
