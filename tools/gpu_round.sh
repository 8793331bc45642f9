#!/bin/bash
# One GPU-box visit: the -m gpu suite, a bench run, and the profile passes.  Usage (through gpurun): tools/gpu_round.sh <tag> [tests|bench|prof ...]
TAG=${1:-r02}; shift
WHAT=${@:-tests bench prof}
mkdir -p gpurun_out
for w in $WHAT; do
  case $w in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_$TAG.log ;;
    bench) timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/bench_$TAG.json ;;
    prof) timeout 1500 bash tools/profile_bench.sh $TAG > gpurun_out/prof_$TAG.log 2>&1; echo "prof rc=$?"; tail -30 gpurun_out/prof_$TAG/summary.txt ;;
  esac
done
