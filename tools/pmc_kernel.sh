#!/bin/bash
# Dev probe (GPU box): SQ counter passes over any command, per kernel matching a SQL LIKE pattern.
# Usage: tools/pmc_kernel.sh <tag> <like-pattern> -- <command>      (counters in their own runs with --kernel-trace only)
TAG=$1; LIKE=$2; shift 3
export TMPDIR=/tmp; REPO=$PWD; OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp
run() { tag=$1; shift
  rocprofv3 --kernel-trace --pmc $PMCS -d $OUT/$tag -o p -- "$@" > $OUT/$tag.log 2>&1
  python - <<PY
import sqlite3,glob
f=glob.glob("$OUT/$tag/**/*.db",recursive=True)
if not f: print("no db for $tag"); raise SystemExit
db=sqlite3.connect(f[0])
print("== $tag")
for n,c,cnt,avg,mx in db.execute("select kernel_name,counter_name,count(*),avg(value),max(value) from counters_collection where kernel_name like '$LIKE' group by kernel_name,counter_name"):
    n=n.replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    print(f"{n[:44]:44s} {c:30s} n={cnt:3d} avg={avg:.5g} max={mx:.5g}")
PY
  rm -rf $OUT/$tag
}
PMCS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" run sq1 "$@"
PMCS="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT" run sq2 "$@"
cd $REPO
