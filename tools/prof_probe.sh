#!/bin/bash
# kernel-trace stats of the nearest probe in both modes
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/prof_probe; mkdir -p $OUT; cd /tmp
for mode in two-phase exact; do
  PROBE_ORACLE=0 SDFHIP_NEAREST=$mode rocprofv3 --kernel-trace --stats -d $OUT/$mode -o p -- python $REPO/tools/gpu_nearest_probe.py 7 > $OUT/$mode.log 2>&1
  python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("$OUT/$mode/*.db")[0] if glob.glob("$OUT/$mode/*.db") else glob.glob("$OUT/$mode/*/*.db")[0])
rows=db.execute("select name,count(*),sum(duration),max(duration) from kernels group by name order by sum(duration) desc limit 12").fetchall()
print("== $mode")
for n,c,s,m in rows: print(f"{n[:70]:70s} {c:5d} {s/1e6:9.3f} ms  max {m/1e6:8.3f}")
PY
  rm -rf $OUT/$mode
done
