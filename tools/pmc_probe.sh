#!/bin/bash
# PMC passes over the nearest probe (two-phase mode): is k_near_candidates bound by the texture-address / L1 path?
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/pmc_probe; mkdir -p $OUT; cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -oE "^\s*(Name|Counter_Name)\s*:\s*\S+|\b(TA|TCP|TD|SQ|TCC)_[A-Z0-9_]+\b" $OUT/counters.txt | sort -u | head -400 > $OUT/counter_names.txt
run() { tag=$1; shift
  PROBE_ORACLE=0 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$tag -o p -- python $REPO/tools/gpu_nearest_probe.py 7 > $OUT/$tag.log 2>&1
  python - <<PY
import sqlite3,glob
f=glob.glob("$OUT/$tag/**/*.db",recursive=True)
if not f: print("no db for $tag"); raise SystemExit
db=sqlite3.connect(f[0])
print("== $tag")
for n,c,cnt,avg,mx in db.execute("select kernel_name,counter_name,count(*),avg(value),max(value) from counters_collection where kernel_name like '%k_near_candidates%' or kernel_name like '%k_near_resolve%' group by kernel_name,counter_name"):
    print(f"{n[:60]:60s} {c:34s} n={cnt:3d} avg={avg:.4g} max={mx:.4g}")
PY
  rm -rf $OUT/$tag
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAVES
run ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
