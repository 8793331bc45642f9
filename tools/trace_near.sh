#!/bin/bash
# per-kernel times of tools/gpu_nearest_bench.py (args: env assignments)
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/trace_near; mkdir -p $OUT; cd /tmp
env "$@" timeout 240 rocprofv3 --kernel-trace -d $OUT/t -o p -- python $REPO/tools/gpu_nearest_bench.py > $OUT/t.log 2>&1
python - <<PY
import sqlite3,glob
f=glob.glob("$OUT/t/**/*.db",recursive=True)
db=sqlite3.connect(f[0])
print("== $@")
for n,g,c,s,mx,mn in db.execute("select name,grid_x,count(*),sum(duration),max(duration),min(duration) from kernels where name like '%near%' group by name,grid_x order by grid_x"):
    print(f"{n.split('(')[0][-34:]:34s} grid {g:8d} n={c:3d} avg {s/c/1e6:8.3f} ms  min {mn/1e6:8.3f} max {mx/1e6:8.3f}")
PY
tail -2 $OUT/t.log
rm -rf $OUT/t
