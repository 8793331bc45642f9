"""Dev probe: host-pointer (numpy) query path timing, PCIe-inclusive."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import sdflib_amd as S
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
v, f = bumpy_icosphere(6); box = box_with_margin(v)
m = S.Mesh(v, f); oc = S.OctreeSdf(m, box, 7, 3, 1e-3, num_threads=2)
for n in (1_000_000, 10_000_000):
    pts = random_points_in_box(oc.get_grid_bounding_box(), n, seed=3)
    oc.get_distance(pts[:1000])
    for grad in (False, True):
        t = time.time(); d = oc.get_distance(pts, gradient=grad); dt = time.time() - t
        t = time.time(); d = oc.get_distance(pts, gradient=grad); dt2 = time.time() - t
        byts = n * (16 + (12 if grad else 0))
        print(f"host query n={n} grad={grad}: {dt*1e3:.1f} / {dt2*1e3:.1f} ms  -> {n/dt2/1e6:.1f} Mq/s, {byts/dt2/1e9:.1f} GB/s over PCIe")
ex = S.ExactOctreeSdf(m, box, 6, 3, 64)
pts = random_points_in_box(oc.get_grid_bounding_box(), 2_000_000, seed=4)
ex.get_distance(pts[:1000])
t = time.time(); ex.get_distance(pts); print(f"exact host query 2M: {(time.time()-t)*1e3:.1f} ms")
