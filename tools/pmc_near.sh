#!/bin/bash
# dynamic instruction mix of the nearest kernels on tools/gpu_nearest_bench.py (hard timeouts); args: env assignments
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/pmc_near; mkdir -p $OUT; cd /tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  env "$@" timeout 240 rocprofv3 --kernel-trace --pmc $pass -d $OUT/t -o p -- python $REPO/tools/gpu_nearest_bench.py > $OUT/t.log 2>&1
  python - <<PY
import sqlite3,glob
f=glob.glob("$OUT/t/**/*.db",recursive=True)
if f:
    db=sqlite3.connect(f[0]); print("== $@ | $pass")
    d={}
    for n,g,c,cnt,avg in db.execute("select kernel_name,grid_size_x,counter_name,count(*),avg(value) from counters_collection where kernel_name like '%near%' group by kernel_name,grid_size_x,counter_name"):
        d.setdefault((n.split('<')[0].split('::')[-1][:20],g,cnt),{})[c]=avg
    for k,v in sorted(d.items(), key=lambda kv: kv[0][1]):
        if max(v.values()) < 1e7: continue
        print(k, " ".join(f"{c[3:]}={x:.4g}" for c,x in sorted(v.items())))
PY
  rm -rf $OUT/t
done
