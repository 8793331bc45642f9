#!/bin/bash
# as tools/trace_build.sh, every kernel listed (top 40) — where a build's GPU time goes.  Usage: tools/trace_build_full.sh [ENV=V ...]
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/trace_build; mkdir -p $OUT; cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o p -- python $REPO/tools/${PROBE_SCRIPT:-gpu_build_probe.py} > $OUT/t.log 2>&1
grep build $OUT/t.log
python - <<PY
import sqlite3,glob
f=glob.glob("$OUT/t/**/*.db",recursive=True)
db=sqlite3.connect(f[0])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
k=[t for t in tabs if t.startswith('kernels')][0]
rows=list(db.execute(f"select name,count(*),sum(duration),max(duration) from {k} group by name order by sum(duration) desc"))
tot=sum(r[2] for r in rows)
print(f"GPU busy per build: {tot/4e6:.2f} ms in {sum(r[1] for r in rows)//4} launches")
for n,c,s,mx in rows[:40]:
    nm = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-44:]
    print(f"{nm:46s} launches/build {c/4:6.1f}  ms/build {s/4e6:7.3f}  max {mx/1e6:6.3f} ms")
PY
rm -rf $OUT/t
