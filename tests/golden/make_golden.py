"""Generates tests/golden/golden_small.npz.

Two kinds of content, and the file says which is which:
 * REFERENCE OUTPUTS (`nearest_ids_source` = "reference:TriangleMeshDistance.h"): `nearest_ids`, `tie_points` /
   `tie_nearest_ids` / `tie_distances`, `bvh_spheres` / `bvh_children` come from oracle/_ref/libtmd_ref.so, i.e. the
   reference's own tmd::TriangleMeshDistance compiled here from /root/reference with no stand-ins (oracle/ref_tmd.cpp) and
   used as SdfLib's ICG wrapper uses it (include/SdfLib/TrianglesInfluence.h:884-905).  They pin SURVEY.md §8 row a5 —
   "which triangle is nearest" — also on the GPU box, where /root/reference does not exist.
 * ORACLE OUTPUTS (everything else): the rest of the reference's path includes glm / cereal / spdlog, which are neither
   vendored under /root/reference nor installed, and no stand-in headers are written; those vectors come from oracle/
   (the CPU restatement) and are a regression pin and a checker-independent fixture, NOT reference outputs.
Run from the repo root (needs /root/reference):  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as O  # noqa: E402
from oracle import pyref as R  # noqa: E402
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

# --- the real reference (oracle/_ref): nearest ids, tie points, BVH ---
ref = R.RefMesh(v, f)
ref_ids = ref.nearest(pts)
assert np.array_equal(ref_ids, m.nearest(pts)), "oracle/orc_bvh.h diverges from the reference: fix it before regenerating"
tri = v[f]
tie = np.concatenate([v,                                                     # exactly on every vertex
                      (tri[:, 0] + tri[:, 1]) * np.float32(0.5), (tri[:, 1] + tri[:, 2]) * np.float32(0.5),
                      (tri[:, 2] + tri[:, 0]) * np.float32(0.5),             # on every edge
                      (tri[:, 0] + tri[:, 1] + tri[:, 2]) / np.float32(3),   # on every face
                      np.zeros((1, 3), np.float32)]).astype(np.float32)      # the centre
tie_ids, tie_d = ref.nearest(tie, with_dist=True)
bvh_sph, bvh_lr, _ = ref.bvh_export()
bvh_sph[bvh_lr[:, 0] == -1] = 0.0                                            # a leaf's spheres are never written: no garbage in the file
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_small.npz"),
                    subdiv=SUBDIV, depth=DEPTH, start_depth=START, exact_depth=EXACT_DEPTH, exact_min_tri=EXACT_MIN,
                    vertices=v, triangles=f, box=box, points=pts, triangle_data=m.triangle_data(), nearest_ids=ref_ids,
                    nearest_ids_source="reference:TriangleMeshDistance.h", tie_points=tie, tie_nearest_ids=tie_ids, tie_distances=tie_d,
                    bvh_spheres=bvh_sph, bvh_children=bvh_lr,
                    octree_words=oc.data(), octree_value_range=np.float32(oc.value_range), octree_min_border=np.float32(oc.min_border),
                    octree_dist=d, octree_grad=g, exact_nodes=nodes, exact_has=has, exact_sets=sets, exact_masks=masks,
                    exact_dist=ed, exact_grad=eg, exact_tri=et)
print("written", os.path.getsize(os.path.join(ROOT, "tests", "golden", "golden_small.npz")), "bytes")

# --- reference ids at the BASELINE configs' full sizes (ids only: the points are regenerated from seeds, refpoints.py) ---
from refpoints import ref_fixture_cases, points_digest  # noqa: E402
big = {"source": "reference:TriangleMeshDistance.h"}
for name, bv, bf, bp in ref_fixture_cases():
    ids = R.RefMesh(bv, bf).nearest(bp)
    assert np.array_equal(ids, O.Mesh(bv, bf).nearest(bp)), name
    big[name + "_ids"] = ids
    big[name + "_digest"] = np.uint64(points_digest(bp))
    print(name, len(bf), "triangles,", len(bp), "points")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_nearest_large.npz"), **big)
print("written", os.path.getsize(os.path.join(ROOT, "tests", "golden", "ref_nearest_large.npz")), "bytes")
