#!/bin/bash
# Dev probe (GPU box): per-dispatch durations of kernels matching a substring.  Usage: tools/trace_probe.sh <tag> <kernel-substr> -- <command>
TAG=$1; KSUB=$2; shift 3
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace -d $OUT/t -o r -- "$@" > $OUT/run.log 2>&1
cd $REPO
python - "$OUT" "$KSUB" <<'PY'
import sqlite3, sys, os, glob
out, ksub = sys.argv[1], sys.argv[2]
for p in glob.glob(os.path.join(out, "t", "**", "*.db"), recursive=True):
    d = sqlite3.connect(p)
    cols = [r[1] for r in d.execute("pragma table_info(kernels)")]
    print(cols)
    for row in d.execute("select name, duration, grid_x, workgroup_x from kernels order by start").fetchall():
        if ksub in row[0]:
            print(f"{row[0][:40]:40s} {row[1]/1e6:9.3f} ms grid={row[2]} wg={row[3]}")
    os.remove(p)
PY
