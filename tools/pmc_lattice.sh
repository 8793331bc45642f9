#!/bin/bash
# memory counters of the lattice kernels on tools/gpu_lattice_probe.py (hard timeouts)
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/pmc_lattice; mkdir -p $OUT; cd /tmp
for pass in "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  env "$@" timeout 240 rocprofv3 --kernel-trace --pmc $pass -d $OUT/t -o p -- python $REPO/tools/gpu_lattice_probe.py > $OUT/t.log 2>&1
  python - <<PY
import sqlite3,glob
f=glob.glob("$OUT/t/**/*.db",recursive=True)
if f:
    db=sqlite3.connect(f[0]); print("== $pass")
    d={}
    for n,g,c,cnt,avg in db.execute("select kernel_name,grid_size_x,counter_name,count(*),avg(value) from counters_collection where kernel_name like '%lattice%' or kernel_name like '%query_grid%' group by kernel_name,grid_size_x,counter_name"):
        d.setdefault((n.split('(')[0][-28:],g,cnt),{})[c]=avg
    for k,v in sorted(d.items(), key=lambda kv: (kv[0][0],kv[0][1])):
        print(k, " ".join(f"{c}={x:.5g}" for c,x in sorted(v.items())))
else:
    print("no db for $pass"); print(open("$OUT/t.log").read()[-1500:])
PY
  rm -rf $OUT/t
done
