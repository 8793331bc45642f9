#!/bin/bash
# Copy the summaries of gpurun_out/prof_<tag> (tools/profile_bench.sh) into profiles/<tag>_bench_* and stamp the commit they belong to.
# Usage: tools/keep_profile.sh <tag> [bench-json]
TAG=$1
SRC=gpurun_out/prof_$TAG
for k in kernel_stats pmc pmc_sq pmc_ea; do [ -f $SRC/summary_$k.csv ] && cp $SRC/summary_$k.csv profiles/${TAG}_bench_$k.csv; done
[ -f $SRC/summary_copies.txt ] && cp $SRC/summary_copies.txt profiles/${TAG}_bench_copies.txt
python - <<PY
import json, subprocess
m = json.load(open("$SRC/summary_meta.json"))
m["git_head_when_kept"] = subprocess.check_output(["git", "rev-parse", "HEAD"]).decode().strip()
m["git_dirty_sources"] = subprocess.check_output(["git", "status", "--porcelain", "sdflib_amd/csrc", "include"]).decode().split("\n")[:-1]
json.dump(m, open("profiles/${TAG}_bench_meta.json", "w"), indent=1)
PY
[ -n "$2" ] && cp $2 profiles/${TAG}_bench_n1.json
ls -la profiles/${TAG}_*
