"""Dev probe: CONTINUITY build timing."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sdflib_amd as S
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
s = int(sys.argv[1]) if len(sys.argv) > 1 else 7
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
start = int(sys.argv[3]) if len(sys.argv) > 3 else 3
v, f = bumpy_icosphere(s); box = box_with_margin(v)
m = S.Mesh(v, f); m.build_bvh()
for it in range(2):
    t = time.time(); oc = S.OctreeSdf(m, box, depth, start, 1e-3, init_algorithm=S.ALG_CONTINUITY, num_threads=2); dt = time.time() - t
    i = oc.info
    print(f"continuity build {dt:.3f}s words={i.num_words} leaves={i.num_leaves} samples={i.num_samples} rescheduled={i.post_pass_scheduled}")
    del oc
