#!/bin/bash
# GPU-box soak of the differential fuzzers (tools/gpu_fuzz.py in its four modes, tools/gpu_fuzz_ranks.py): totals to gpurun_out/soak_<tag>.txt
TAG=${1:-soak}; N=${2:-1500}; SEED=${3:-3000000}
OUT=gpurun_out/soak_$TAG.txt; : > $OUT
i=0
for mode in "" continuity exact big; do
  FUZZ_MODE=$mode timeout 1500 python tools/gpu_fuzz.py $N $((SEED + 100000 * i)) > gpurun_out/soak_${TAG}_$i.log 2>&1
  echo "mode '${mode:-default}': $(tail -1 gpurun_out/soak_${TAG}_$i.log); mismatches: $(grep -c -E 'MISMATCH|ERROR' gpurun_out/soak_${TAG}_$i.log)" >> $OUT
  grep -E "MISMATCH|ERROR" gpurun_out/soak_${TAG}_$i.log | head -5 >> $OUT
  i=$((i + 1))
done
FUZZ_WORLDS=2,3,5,8 timeout 1500 python tools/gpu_fuzz_ranks.py 40 $((SEED + 900000)) > gpurun_out/soak_${TAG}_ranks.log 2>&1
echo "ranks: $(tail -1 gpurun_out/soak_${TAG}_ranks.log)" >> $OUT
cat $OUT
