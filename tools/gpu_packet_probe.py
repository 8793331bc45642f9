"""Dev probe: what a PACKET of g Morton-consecutive level-7 sample points would visit if it shared one traversal: the search run at the
packet's centre with its bound widened by PROBE_SLACK (set to about twice the packets' radius), against the members' own searches."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_near_hist.py")).read().split("for level in levels:")[0])
g = int(os.environ.get("PROBE_PACKET", "4"))
nodes, pts = level_points(7)
n = len(pts) // g * g
P = pts[:n].reshape(-1, g, 3)
pc = 0.5 * (P.min(1) + P.max(1)); rp = np.sqrt(((P - pc[:, None, :]) ** 2).sum(2)).max(1)
cell = size / 2 ** 7
print(f"{n//g} packets of {g}: radius mean {rp.mean()/cell:.2f} cells, p50 {np.median(rp)/cell:.2f}, p90 {np.percentile(rp,90)/cell:.2f}, max {rp.max()/cell:.2f}; PROBE_SLACK = {os.environ.get('PROBE_SLACK')}")
out = run(np.ascontiguousarray(pc.astype(np.float32)))
ex, tr = out[:, 1].astype(np.int64), out[:, 3].astype(np.int64)
tight = rp <= float(os.environ.get("PROBE_SLACK", "0")) / 2 * 1.0001
print(f"packet-centre searches: expansions {ex.mean():.1f} per packet = {ex.mean()/g:.1f} per member, triangles {tr.mean():.1f} per packet; packets whose radius is within the slack: {tight.mean()*100:.1f} % (expansions {ex[tight].mean():.1f}, triangles {tr[tight].mean():.1f})")
