#!/bin/bash
# dynamic instruction mix of the nearest kernels on tools/gpu_nearest_bench.py (one SQ pass per mode, hard timeouts)
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/pmc_near; mkdir -p $OUT; cd /tmp
for mode in two-phase exact; do
  SDFHIP_NEAREST=$mode timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY -d $OUT/$mode -o p -- python $REPO/tools/gpu_nearest_bench.py > $OUT/$mode.log 2>&1
  python - <<PY
import sqlite3,glob
f=glob.glob("$OUT/$mode/**/*.db",recursive=True)
if f:
    db=sqlite3.connect(f[0]); print("== $mode")
    rows=db.execute("select kernel_name,grid_size_x,counter_name,count(*),avg(value) from counters_collection where kernel_name like '%near%' group by kernel_name,grid_size_x,counter_name").fetchall()
    d={}
    for n,g,c,cnt,avg in rows: d.setdefault((n.split('(')[0][-40:],g),{})[c]=avg
    for k,v in d.items():
        w=v.get('SQ_WAVES',1)
        print(k, f"waves {w:.0f} | per wave: VALU {v.get('SQ_INSTS_VALU',0)/w:.0f} SALU {v.get('SQ_INSTS_SALU',0)/w:.0f} VMEM {v.get('SQ_INSTS_VMEM_RD',0)/w:.0f} LDS {v.get('SQ_INSTS_LDS',0)/w:.0f} | lanes/VALU {v.get('SQ_THREAD_CYCLES_VALU',0)/max(v.get('SQ_ACTIVE_INST_VALU',1),1)/64*100:.1f}% ")
PY
  rm -rf $OUT/$mode
done
