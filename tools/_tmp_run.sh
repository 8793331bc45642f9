for d in 0 4; do echo "debug $d"; PROBE_N=256,512 SDFHIP_LAT_DEBUG=$d bash tools/trace_lattice.sh 2>&1 | grep "lattice"; done
