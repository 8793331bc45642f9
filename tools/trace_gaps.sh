#!/bin/bash
# Where a build's WALL time goes beyond its GPU time: kernel trace of a probe script (PROBE_SCRIPT, default gpu_continuity_probe.py), the last
# build's launches in start order, and the idle gaps between consecutive kernels aggregated by (kernel before -> kernel after).
# Usage: tools/trace_gaps.sh [ENV=V ...]
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/trace_gaps; mkdir -p $OUT; cd /tmp
env -u SDFHIP_TIMING PROBE_QUIET=1 "$@" timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o p -- python $REPO/tools/${PROBE_SCRIPT:-gpu_continuity_probe.py} > $OUT/t.log 2>&1
grep -E "^build|build_bvh" $OUT/t.log | tail -4
python - <<PY
import sqlite3,glob,collections
f=glob.glob("$OUT/t/**/*.db",recursive=True)
db=sqlite3.connect(f[0])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
k=[t for t in tabs if t.startswith('kernels')][0]
cols=[r[1] for r in db.execute(f"pragma table_info({k})")]
rows=list(db.execute(f"select name,start,end from {k} order by start"))
try:
    grids={(n,s):g for n,s,g in db.execute(f"select name,start,grid_x from {k}")}
except Exception:
    grids={}
short=lambda n: n.replace('(anonymous namespace)::','').replace('void ','').split('(')[0][-40:]
# the last build = the launches after the last idle gap longer than 2 ms
cut=0
for i in range(1,len(rows)):
    if rows[i][1]-rows[i-1][2] > 2_000_000: cut=i
R=rows[cut:]
busy=sum(e-s for _,s,e in R); span=R[-1][2]-R[0][1]
print(f"last burst: {len(R)} launches, span {span/1e6:.2f} ms, GPU busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms")
per=collections.defaultdict(lambda:[0,0])
for n,s,e in R:
    per[short(n)][0]+=e-s; per[short(n)][1]+=1
print("  GPU time by kernel:")
for key,(g,c) in sorted(per.items(), key=lambda kv:-kv[1][0])[:24]:
    print(f"  {g/1e6:7.3f} ms in {c:3d} launches  {key}")
print("  longest launches (offset in the burst, duration, work-items):")
for n,s,e in sorted(R, key=lambda r: r[1]-r[2])[:10]:
    print(f"   +{(s-R[0][1])/1e6:7.3f} ms  {(e-s)/1e6:7.3f} ms  {grids.get((n,s),0):>9}  {short(n)}")
print("  idle gaps:")
gaps=collections.defaultdict(lambda:[0,0])
for a,b in zip(R[:-1],R[1:]):
    g=b[1]-a[2]
    if g>3000:
        key=short(a[0])+" -> "+short(b[0]); gaps[key][0]+=g; gaps[key][1]+=1
for key,(g,c) in sorted(gaps.items(), key=lambda kv:-kv[1][0])[:28]:
    print(f"  {g/1e6:7.3f} ms in {c:3d} gaps  {key}")
PY
rm -rf $OUT/t
