#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of the bench command + separate PMC passes.
# Usage: tools/profile_bench.sh <round-tag>
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-forecast ${BENCH_EXTRA:-}"
cd /tmp
timeout 700 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/bench_trace.log 2>&1
# PMC passes: counters only, with kernel-trace only (no sys/hip/hsa traces)
timeout 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/bench_pmc_fetch.log 2>&1
timeout 700 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $CMD > $OUT/bench_pmc_write.log 2>&1
timeout 700 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_l2 -o bench -- $CMD > $OUT/bench_pmc_l2.log 2>&1
timeout 700 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o bench -- $CMD > $OUT/bench_pmc_sq.log 2>&1
# the L2 -> fabric read requests by size and by destination (is there a DRAM-side count that excludes Infinity-Cache hits?)
# (five TCC counters in one pass exceed the hardware's counter slots: two passes)
timeout 700 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d $OUT/pmc_ea -o bench -- $CMD > $OUT/bench_pmc_ea.log 2>&1
timeout 700 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $OUT/pmc_ea2 -o bench -- $CMD > $OUT/bench_pmc_ea2.log 2>&1
cd $REPO
python tools/summarize_rocpd.py gpurun_out/prof_$TAG gpurun_out/prof_$TAG/summary > gpurun_out/prof_$TAG/summary.txt 2>&1
# what ties the counters to the code: hashes of the sources the profiled library was built from (bench.py refuses a profile whose hashes
# differ from the files it is run with; the GPU box has no .git, so the commit id is added when the summaries are copied to profiles/)
python - <<PY > gpurun_out/prof_$TAG/summary_meta.json
import json, sys, time
sys.path.insert(0, "$REPO")
import bench
print(json.dumps({"tag": "$TAG", "command": "$CMD", "recorded_unix": int(time.time()), "source_hashes": bench.source_hashes()}, indent=1))
PY
rm -f gpurun_out/prof_$TAG/*/bench_results.db          # keep only the text summaries (gpurun_out is capped at 64 MiB)
ls gpurun_out/prof_$TAG
