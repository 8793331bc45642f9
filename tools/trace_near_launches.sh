#!/bin/bash
# every k_near_* dispatch of tools/gpu_build_probe.py's LAST build, in order (args: env assignments)
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/trace_nl; mkdir -p $OUT; cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o p -- python $REPO/tools/gpu_build_probe.py > $OUT/t.log 2>&1
grep build $OUT/t.log | tail -2
python - <<PY
import sqlite3,glob
f=glob.glob("$OUT/t/**/*.db",recursive=True)
db=sqlite3.connect(f[0])
rows=list(db.execute("select name,grid_x,start,duration from kernels where name like '%k_near%' order by start"))
per=len(rows)//4
print("== $@")
for n,g,s,d in rows[-per:]:
    print(f"{n.split('(')[0].split('::')[-1][:28]:28s} grid {g:8d} {d/1e6:8.3f} ms")
PY
rm -rf $OUT/t
