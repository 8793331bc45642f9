#!/bin/bash
# A/B of the candidate search's knobs.  Usage: tools/gpu_near_ab.sh TAG "A=1,B=2" "A=3" ...   (per configuration: per-query counters at
# level 7, kernel traces of four C2 builds, four torus-knot builds and four 1.31 M-triangle builds)
TAG=$1; shift
mkdir -p gpurun_out; OUT=gpurun_out/near_ab_$TAG.txt; : > $OUT
for V in "$@"; do
  E=${V//,/ }
  echo "#### $V" >> $OUT
  [ -n "$NO_HIST" ] || env $E python tools/gpu_near_hist.py 7 7 2>&1 | grep -E "\[default" >> $OUT
  bash tools/trace_build.sh $E 2>&1 | grep -E "^build|GPU busy|k_near" >> $OUT
  bash tools/trace_build.sh $E PROBE_KNOT=1 2>&1 | grep -E "^build|GPU busy|k_near" >> $OUT
  bash tools/trace_build.sh $E PROBE_SUBDIV=8 2>&1 | grep -E "^build|GPU busy|k_near" >> $OUT
done
cat $OUT
