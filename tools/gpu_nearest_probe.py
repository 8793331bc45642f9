"""GPU probe: the two-phase nearest-triangle search (dev_bvh_fast.h) against the order-exact traversal and the oracle.
Usage: python tools/gpu_nearest_probe.py [subdiv]   (SDFHIP_NEAREST=exact selects the old path for the whole process)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import sdflib_amd as S
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin, random_points_in_box
s = int(sys.argv[1]) if len(sys.argv) > 1 else 7
v, f = bumpy_icosphere(s); box = box_with_margin(v)
gm = S.Mesh(v, f); gm.build_bvh()
mode = os.environ.get("SDFHIP_NEAREST", "two-phase")
pts = random_points_in_box(box, 2_000_000, seed=3)
near = (v[np.random.default_rng(1).integers(0, len(v), 1_000_000)] + np.random.default_rng(2).normal(0, 0.01, (1_000_000, 3))).astype(np.float32)
for name, p in (("uniform", pts), ("near-surface", near)):
    gm.nearest_triangle(p[:1000])
    t = time.time(); ids = gm.nearest_triangle(p); dt = time.time() - t
    print(f"[{mode}] {name}: {len(p)} points in {dt*1e3:.1f} ms (incl. PCIe)", flush=True)
    if os.environ.get("PROBE_ORACLE", "1") == "1":
        from oracle import pyoracle as O
        om = O.Mesh(v, f)
        want = om.nearest(p[:300000])
        bad = int((want != ids[:300000]).sum())
        print(f"    vs oracle on 300000: {bad} mismatches", flush=True)
for alg, nm in ((S.ALG_NO_CONTINUITY, "NO_CONTINUITY"), (S.ALG_CONTINUITY, "CONTINUITY")):
    S.OctreeSdf(gm, box, 6, 3, 1e-3, init_algorithm=alg, num_threads=2)
    t = time.time(); tr = S.OctreeSdf(gm, box, 8, 3, 1e-3, init_algorithm=alg, num_threads=2); dt = time.time() - t
    i = tr.info
    print(f"[{mode}] {nm} depth 8 build {dt*1e3:.1f} ms, traversals {i.num_traversals}, fallbacks {i.num_nearest_fallbacks} ({100.0*i.num_nearest_fallbacks/max(i.num_traversals,1):.3f} %), words {i.num_words}", flush=True)
