"""Mesh creation timing (host arrays in -> TriangleData on the device): PROBE_SUBDIV 7 / 8."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdflib_amd as S
from sdflib_amd import meshgen
for sub in (7, 8):
    v, f = meshgen.bumpy_icosphere(sub)
    ctx = S.Context(0)
    ts = []
    for rep in range(6):
        t0 = time.perf_counter(); m = S.Mesh(v, f, ctx); ts.append(time.perf_counter() - t0); m.close()
    print(f"{len(f)} triangles: mesh creation " + " ".join(f"{1e3 * t:.2f}" for t in ts) + " ms", flush=True)
