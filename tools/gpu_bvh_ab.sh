#!/bin/bash
# A/B of the BVH build on the GPU box: host planner alone vs device subtrees of 4096 / 16384 (SDFHIP_BVH_DEVICE_SUBTREES), at 327 680 and
# 1.31 M triangles; then the hybrid tests.  Usage (through gpurun): tools/gpu_bvh_ab.sh <tag>
TAG=${1:-r03}
mkdir -p gpurun_out
OUT=gpurun_out/bvh_ab_$TAG.txt
: > $OUT
for sub in 7 8; do
  for mode in host 4096 16384; do
    echo "== subdiv $sub, $mode" >> $OUT
    if [ $mode = host ]; then unset SDFHIP_BVH_DEVICE_SUBTREES; else export SDFHIP_BVH_DEVICE_SUBTREES=$mode; fi
    PROBE_SUBDIV=$sub PROBE_REPS=5 SDFHIP_TIMING=1 timeout 300 python tools/gpu_bvh_probe.py 2>&1 | grep -E "build_bvh|bvh plan|bvh:|device subtrees|triangles" >> $OUT
  done
done
unset SDFHIP_BVH_DEVICE_SUBTREES
timeout 900 python -m pytest tests/test_gpu_octree.py -m gpu -x -q -k "hybrid_bvh or imported_bvh" > gpurun_out/pytest_bvh_$TAG.log 2>&1; echo "pytest rc=$?" >> $OUT
tail -3 gpurun_out/pytest_bvh_$TAG.log >> $OUT
cat $OUT
