"""BVH planner timing (SDFHIP_TIMING=1 prints the phases): icosphere subdivision PROBE_SUBDIV (8 = 1.31 M triangles)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SDFHIP_TIMING", "1")
import sdflib_amd as S
from sdflib_amd import meshgen
sub = int(os.environ.get("PROBE_SUBDIV", "8"))
if os.environ.get("PROBE_KNOT"):                     # PROBE_KNOT=nu:nv -> torus knot of 2 * nu * nv triangles
    nu, nv = (int(x) for x in os.environ["PROBE_KNOT"].split(":"))
    v, f = meshgen.torus_knot(nu=nu, nv=nv)
else:
    v, f = meshgen.bumpy_icosphere(sub)
print("triangles", len(f), "hardware threads", os.cpu_count(), flush=True)
for rep in range(int(os.environ.get('PROBE_REPS', '3'))):
    mesh = S.Mesh(v, f)
    time.sleep(0.4)                      # let the container's CPU quota recover: back-to-back builds throttle each other
    import resource
    r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter(); s = mesh.build_bvh(); dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
    print(f"build_bvh: {dt:.4f} s (reported {s}), cpu {r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime:.3f} s", flush=True)
    del mesh
