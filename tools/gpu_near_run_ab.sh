#!/bin/bash
# A/B of the candidate search's run length (SDFHIP_NEAR_RUN): per-query counters at level 7 and the kernel trace of four C2 builds
mkdir -p gpurun_out; OUT=gpurun_out/near_run_ab.txt; : > $OUT
for RUN in ${RUNS:-1 4 8 16 32}; do
  echo "#### SDFHIP_NEAR_RUN=$RUN" >> $OUT
  SDFHIP_NEAR_RUN=$RUN python tools/gpu_near_hist.py 7 7 2>&1 | grep -E "==|\[" >> $OUT
  bash tools/trace_build.sh SDFHIP_NEAR_RUN=$RUN 2>&1 | head -8 >> $OUT
  PROBE_KNOT=1 bash tools/trace_build.sh SDFHIP_NEAR_RUN=$RUN PROBE_KNOT=1 2>&1 | head -7 >> $OUT
done
cat $OUT
