#!/bin/bash
# Dev probe (run on the GPU box through gpurun): SQ counters of one kernel family for a python command.
# Usage: tools/pmc_probe.sh <out-tag> <kernel-substring> -- <command...>
TAG=$1; KSUB=$2; shift 3
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES -d $OUT/p1 -o r -- "$@" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 -d $OUT/p2 -o r -- "$@" > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/p3 -o r -- "$@" > $OUT/p3.log 2>&1
cd $REPO
python - "$OUT" "$KSUB" <<'PY'
import sqlite3, sys, os, glob
out, ksub = sys.argv[1], sys.argv[2]
for sub in ("p1", "p2", "p3"):
    for p in glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True):
        d = sqlite3.connect(p)
        try:
            rows = d.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
        except Exception as e:
            print(sub, "error", e); continue
        for name, cname, n, tot in rows:
            if ksub in name:
                print(f"{sub} {name[:60]:60s} {cname:28s} dispatches={n:5d} sum={tot:.6g}")
        os.remove(p)
PY
tail -3 $OUT/p1.log
