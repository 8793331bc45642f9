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
    agg = {}
    for row in d.execute("select name, duration, grid_x, workgroup_x from kernels order by start").fetchall():
        row = (row[0].replace("(anonymous namespace)::", "").replace("void ", ""),) + tuple(row[1:])
        if ksub in row[0]:
            if os.environ.get("TRACE_AGG"):
                k = row[0].split("(")[0]; a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += row[1] / 1e6; a[2] = max(a[2], row[1] / 1e6)
            else:
                print(f"{row[0][:40]:40s} {row[1]/1e6:9.3f} ms grid={row[2]} wg={row[3]}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:60]:60s} calls={a[0]:4d} total={a[1]:9.3f} ms max={a[2]:8.3f} ms")
    os.remove(p)
PY
