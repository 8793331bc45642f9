#!/bin/bash
# A/B of the BVH build on the GPU box: host planner alone (mode 0) vs host planner + device subtrees of N triangles (SDFHIP_BVH_BUILD=host,
# SDFHIP_BVH_DEVICE_SUBTREES=N) vs the whole tree on the device (mode device, the library's default), on 327 680 / 655 360 / 1.31 M
# triangles; every mode runs in ROUNDS fresh processes, interleaved (the planner's wall time drifts by several ms from process to process
# under the box's CPU quota); then the hybrid tests.  Usage (through gpurun): tools/gpu_bvh_ab.sh <tag> [rounds]
TAG=${1:-r03}; ROUNDS=${2:-3}
mkdir -p gpurun_out
OUT=gpurun_out/bvh_ab_$TAG.txt
: > $OUT
for mesh in "PROBE_SUBDIV=7" "PROBE_KNOT=2048:160" "PROBE_SUBDIV=8"; do
  for r in $(seq $ROUNDS); do
    for mode in 0 4096 device; do
      echo "== $mesh, device subtrees $mode, round $r" >> $OUT
      if [ $mode = device ]; then B=device; M=4096; else B=host; M=$mode; fi
      env $mesh SDFHIP_BVH_BUILD=$B SDFHIP_BVH_DEVICE_SUBTREES=$M PROBE_REPS=5 SDFHIP_TIMING=1 timeout 300 python tools/gpu_bvh_probe.py 2>&1 | grep -E "build_bvh|bvh:|device subtrees|triangles" >> $OUT
    done
  done
done
python - $OUT <<'PY' | tee -a $OUT
import re, sys, statistics
cur, res = None, {}
for line in open(sys.argv[1]):
    m = re.match(r"== (\S+), device subtrees (\w+)", line)
    if m: cur = (m.group(1), m.group(2)); continue
    m = re.match(r"build_bvh: ([0-9.]+) s", line)
    if m and cur: res.setdefault(cur, []).append(float(m.group(1)))
for k in sorted(res): v = sorted(res[k]); print(f"SUMMARY {k[0]:22s} subtrees {k[1]:>6s}: median {statistics.median(v)*1e3:6.1f} ms, min {v[0]*1e3:6.1f}, max {v[-1]*1e3:6.1f} ({len(v)} builds)")
PY
unset SDFHIP_BVH_DEVICE_SUBTREES
timeout 900 python -m pytest tests/test_gpu_octree.py -m gpu -x -q -k "hybrid_bvh or imported_bvh" > gpurun_out/pytest_bvh_$TAG.log 2>&1; echo "pytest rc=$?" >> $OUT
tail -3 gpurun_out/pytest_bvh_$TAG.log >> $OUT
grep -E "SUMMARY|pytest|passed|failed" $OUT
