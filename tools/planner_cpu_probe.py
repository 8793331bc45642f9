"""Wall and CPU time of the BVH planner alone (no GPU): PROBE_SUBDIV icosphere subdivisions (8 = 1.31 M triangles)."""
import ctypes as C, os, sys, time, resource
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdflib_amd import meshgen
from sdflib_amd._lib import lib
sub = int(os.environ.get("PROBE_SUBDIV", "8"))
v, f = meshgen.bumpy_icosphere(sub)
v = np.ascontiguousarray(v, np.float32); f = np.ascontiguousarray(f, np.uint32)
n = len(f) - 1
sph = np.empty(8 * n, np.float64); kids = np.empty(2 * n, np.int32)
for rep in range(int(os.environ.get("PROBE_REPS", "3"))):
    r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
    lib().sdfhip_test_plan_bvh(v.ctypes.data_as(C.c_void_p), len(v), f.ctypes.data_as(C.c_void_p), len(f), sph.ctypes.data_as(C.c_void_p), kids.ctypes.data_as(C.c_void_p), None)
    dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
    print(f"{len(f)} triangles: wall {dt:.3f} s, cpu user {r1.ru_utime - r0.ru_utime:.3f} s sys {r1.ru_stime - r0.ru_stime:.3f} s = {(r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime) / len(f) * 1e6:.2f} us per triangle", flush=True)
