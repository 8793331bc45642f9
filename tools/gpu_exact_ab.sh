#!/bin/bash
# Dev probe (GPU box): ExactOctreeSdf query, round-2 kernel vs the current one (kernel trace of both), then the exact tests.
mkdir -p gpurun_out
for k in tiles; do
  echo "== SDFHIP_EXACT_KERNEL=$k"
  SDFHIP_EXACT_KERNEL=$k TRACE_AGG=1 bash tools/trace_probe.sh exact_$k "k_exact" -- python $PWD/tools/gpu_exact_probe.py 7 7 3 1e7 2>&1 | tail -8
  grep "exact query" gpurun_out/trace_exact_$k/run.log
done
python -m pytest tests -m gpu -x -q -k "exact or golden or c3" 2>&1 | grep -E "passed|failed|Error|error" | tail -8
