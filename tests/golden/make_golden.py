"""Generates tests/golden/golden_small.npz with the ORACLE (oracle/), i.e. the CPU restatement of the reference.

The reference itself cannot be run here (its glm/cereal/spdlog dependencies are neither vendored under
/root/reference nor installed, and no stand-in headers are written), so these vectors pin the oracle's behaviour
across rounds (regression) and give the GPU tests a checker-independent fixture; they are NOT reference outputs.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box  # noqa: E402

SUBDIV, DEPTH, START, EXACT_DEPTH, EXACT_MIN = 2, 4, 2, 5, 16
v, f = bumpy_icosphere(SUBDIV)
box = box_with_margin(v)
m = O.Mesh(v, f)
pts = random_points_in_box(box, 4096, seed=99)
pts[:64] *= 2.5                                 # some points outside the grid
oc = O.Octree(m, box, DEPTH, START, 1e-3, vertex_cache=False, layout=O.LAYOUT_SUBTREES)
d, g = oc.query(pts, grad=True)
ex = O.Exact(m, box, EXACT_DEPTH, 1, EXACT_MIN)
nodes, has, sets, masks = ex.data()
ed, eg, et = ex.query(pts, grad=True, tri=True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_small.npz"),
                    subdiv=SUBDIV, depth=DEPTH, start_depth=START, exact_depth=EXACT_DEPTH, exact_min_tri=EXACT_MIN,
                    vertices=v, triangles=f, box=box, points=pts, triangle_data=m.triangle_data(), nearest_ids=m.nearest(pts),
                    octree_words=oc.data(), octree_value_range=np.float32(oc.value_range), octree_min_border=np.float32(oc.min_border),
                    octree_dist=d, octree_grad=g, exact_nodes=nodes, exact_has=has, exact_sets=sets, exact_masks=masks,
                    exact_dist=ed, exact_grad=eg, exact_tri=et)
print("written", os.path.getsize(os.path.join(ROOT, "tests", "golden", "golden_small.npz")), "bytes")
