#!/bin/bash
# A/B of the builders' read-backs: mailbox kernel + spin (default) against hipMemcpyAsync + hipStreamSynchronize (SDFHIP_READBACK=copy)
for M in copy mailbox; do
  echo "#### SDFHIP_READBACK=$M"
  for P in gpu_build_probe.py gpu_continuity_probe.py gpu_exact_build_probe.py; do
    SDFHIP_READBACK=$M SDFHIP_TIMING= python tools/$P 2>&1 | grep -E "^build" | tail -2 | sed "s/^/$P: /"
  done
  SDFHIP_READBACK=$M PROBE_SUBDIV=7 PROBE_REPS=4 python tools/gpu_bvh_probe.py 2>&1 | grep -E "build_bvh" | tail -2
  SDFHIP_READBACK=$M PROBE_SUBDIV=8 PROBE_REPS=4 python tools/gpu_bvh_probe.py 2>&1 | grep -E "build_bvh" | tail -2
done
