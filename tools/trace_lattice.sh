#!/bin/bash
# kernel trace of tools/gpu_lattice_probe.py: per (kernel, grid) average duration of the lattice kernels
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/trace_lattice; mkdir -p $OUT; cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o p -- python $REPO/tools/gpu_lattice_probe.py > $OUT/t.log 2>&1
python - <<PY
import sqlite3,glob
f=glob.glob("$OUT/t/**/*.db",recursive=True)
db=sqlite3.connect(f[0])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
k=[t for t in tabs if t.startswith('kernels')][0]
for n,g,c,a,mn in db.execute(f"select name,grid_x,count(*),avg(duration),min(duration) from {k} where name like '%lattice%' or name like '%query_grid%' group by name,grid_x order by name,grid_x"):
    print(f"{n.split('(')[0][-40:]:42s} grid {g:>10} calls {c:4d} avg {a/1e3:8.1f} us min {mn/1e3:8.1f} us")
PY
rm -rf $OUT/t
