#!/bin/bash
# A/B on the GPU box: k_near_resolve for different (tie slots, waves-per-SIMD target) pairs; octree_build.o / octree_continuity.o rebuilt per setting
cd sdflib_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-result -Wno-unused-function -pthread"
for S in "9 3" "8 5" "8 6" "6 6"; do
  set -- $S
  for f in octree_build octree_continuity bvh; do
    X=""; [ $f = bvh ] && X="-mllvm -disable-promote-alloca-to-lds"
    /opt/rocm/bin/hipcc $F $X -DRESOLVE_MAX_TIES=$1 -DRESOLVE_WAVES=$2 -c $f.hip -o $f.o 2>/dev/null &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libsdfhip.so ctx_mesh.o bvh.o octree_build.o octree_continuity.o octree_query.o octree_lattice.o blocks.o exact_build.o exact_query.o multi.o -ldl
  echo "#### RESOLVE_MAX_TIES=$1 RESOLVE_WAVES=$2"
  (cd ../..; bash tools/trace_build.sh 2>&1 | grep -E "^build 0.01|k_near_res|k_near_fall"; bash tools/trace_build.sh PROBE_KNOT=1 2>&1 | grep -E "^build 0.01|k_near_res|k_near_fall"; bash tools/trace_build.sh PROBE_SUBDIV=8 2>&1 | grep -E "^build 0.01|k_near_res|k_near_fall")
done
