"""One-off check (too slow for the suite: the oracle's ExactOctreeSdf build is single-threaded like the reference's): BASELINE
configs[2] at full size — GPU arrays vs the CPU oracle's, bit for bit."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import sdflib_amd as S
from oracle import pyoracle as O
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
s = int(sys.argv[1]) if len(sys.argv) > 1 else 7
v, f = bumpy_icosphere(s); box = box_with_margin(v)
gm = S.Mesh(v, f)
t = time.time(); ge = S.ExactOctreeSdf(gm, box, 7, 3, 128); print(f"gpu build {time.time()-t:.2f} s, max tris in leafs {ge.info.max_triangles_in_leafs}, nodes {ge.info.num_nodes}", flush=True)
om = O.Mesh(v, f)
t = time.time(); oe = O.Exact(om, box, 7, 3, 128); print(f"oracle build {time.time()-t:.1f} s", flush=True)
names = ("nodes", "has", "sets", "masks")
for name, a, b in zip(names, oe.data(), ge.download()):
    print(name, a.shape, b.shape, "EQUAL" if a.shape == b.shape and np.array_equal(a, b) else "DIFFERENT")
pts = random_points_in_box(box, 200000, seed=11)
d0 = oe.query(pts, threads=0); d1 = ge.get_distance(pts)
print("queries bit-equal:", np.array_equal(d0.view(np.uint32), d1.view(np.uint32)))
