mkdir -p gpurun_out
for s in 1536 768 384 192; do
  echo "== SDFHIP_NEAR_SMALL_STEPS=$s"
  SDFHIP_NEAR_SMALL_STEPS=$s python tools/gpu_shard_probe.py 2>&1 | grep triangles
  SDFHIP_NEAR_SMALL_STEPS=$s PROBE_SUBDIV=8 python tools/gpu_shard_probe.py 2>&1 | grep triangles
  SDFHIP_NEAR_SMALL_STEPS=$s PROBE_QUIET=1 python tools/gpu_continuity_probe.py 2>&1 | grep build | tail -2
done
