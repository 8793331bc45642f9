#!/bin/bash
# Multi-GPU smoke for a node with >= 2 GPUs (the round's GPU box has one: this script is what a maintainer / the driver runs where more
# are visible, so that the first N > 1 bench run is not also the first execution of the RCCL branches).
#   1. multi.hip with DISTINCT devices: ncclCommInitAll, the grouped ncclBroadcast all-gather-v of sharded OctreeSdf / ExactOctreeSdf
#      builds, ncclAllReduce of the shared CONTINUITY traversals, every device building its own BVH - arrays compared with the
#      single-device build by tests/cpp/test_cpp_multi.cpp ("transport rccl", "mismatches 0");
#   2. sdflib_amd/distributed.py over torch.distributed "nccl" (= RCCL), one process per GPU: bench.py --gpus N on a small
#      configuration with the 1.31 M-triangle build and its serial / sharded / exchange split.
# Usage: tools/gpu_multi_smoke.sh [N]      (N defaults to the number of visible GPUs, at most 8).  Exit code 0 = all good, 3 = fewer than 2 GPUs.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
VISIBLE=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
N=${1:-$VISIBLE}; [ "$N" -gt 8 ] && N=8
if [ "$VISIBLE" -lt 2 ] || [ "$N" -lt 2 ]; then echo "gpu_multi_smoke: $VISIBLE GPU(s) visible - nothing to do (needs >= 2)"; exit 3; fi
OUT=${OUT:-gpurun_out}; mkdir -p "$OUT"
DEVS=$(seq -s, 0 $((N - 1)))
rc=0

echo "== 1. sdfhip_multi_* on devices $DEVS (RCCL in one process)"
g++ -std=c++17 -O1 -I include tests/cpp/test_cpp_multi.cpp -L sdflib_amd -lsdfhip -Wl,-rpath,"$PWD/sdflib_amd" -o /tmp/sdflib_amd_multi_smoke || exit 1
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from sdflib_amd.meshgen import bumpy_icosphere, box_with_margin
v, f = bumpy_icosphere(5)
v.tofile("/tmp/ms_v.bin"); f.tofile("/tmp/ms_f.bin"); np.asarray(box_with_margin(v), dtype=np.float32).tofile("/tmp/ms_box.bin")
PY
timeout 600 /tmp/sdflib_amd_multi_smoke /tmp/ms_v.bin /tmp/ms_f.bin /tmp/ms_box.bin "$DEVS" > "$OUT/multi_smoke_cpp.log" 2>&1 || rc=1
cat "$OUT/multi_smoke_cpp.log"
grep -q "transport rccl" "$OUT/multi_smoke_cpp.log" || { echo "FAIL: the RCCL transport was not used"; rc=1; }
[ "$(grep -c 'mismatches 0 scalars_equal 1' "$OUT/multi_smoke_cpp.log")" -eq 3 ] || { echo "FAIL: arrays differ from the single-device build"; rc=1; }

echo "== 1b. the same over staged device-to-device copies (SDFHIP_MULTI_TRANSPORT=copy between DISTINCT devices)"
SDFHIP_MULTI_TRANSPORT=copy timeout 600 /tmp/sdflib_amd_multi_smoke /tmp/ms_v.bin /tmp/ms_f.bin /tmp/ms_box.bin "$DEVS" > "$OUT/multi_smoke_cpp_copy.log" 2>&1 || rc=1
grep -q "transport copy" "$OUT/multi_smoke_cpp_copy.log" || { echo "FAIL: the copy transport was not used"; rc=1; }
[ "$(grep -c 'mismatches 0 scalars_equal 1' "$OUT/multi_smoke_cpp_copy.log")" -eq 3 ] || { echo "FAIL (copy transport): arrays differ from the single-device build"; rc=1; }

echo "== 2. bench.py --gpus $N (torch.distributed nccl, one process per GPU)"
timeout 900 python bench.py --gpus "$N" --steps 3 --warmup 1 --subdiv 5 --depth 6 --queries 1000000 --no-cpu-baseline > "$OUT/multi_smoke_bench.json" 2> "$OUT/multi_smoke_bench.err" || rc=1
python - "$OUT/multi_smoke_bench.json" "$N" <<'PY' || rc=1
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
assert len(lines) == 1, "bench.py did not print exactly one JSON line"
d, n = json.loads(lines[0]), int(sys.argv[2])
c, b = d["collectives"], d["build_1m"]
assert d["n_gpus"] == n and c["backend"] == "nccl" and c["ranks_seen"] == n and c["rank_sum_ok"], c
assert b["n_gpus"] == n and b["words"] == 20058064, b
print(f"value {d['value']} Mq/s on {n} GPUs ({d['per_gpu_mqueries_s']} per GPU, roofline frac {d['roofline']['frac']}); 1.31 M build {b['octree_build_s']} s, split {b['split']}")
PY
[ $rc -eq 0 ] && echo "gpu_multi_smoke: ok" || { echo "gpu_multi_smoke: FAILED"; tail -20 "$OUT/multi_smoke_bench.err"; }
exit $rc
