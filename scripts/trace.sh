#!/bin/bash
# quick kernel-trace of a command on the GPU box: scripts/trace.sh <tag> <cmd...>; prints per-kernel stats
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- "$@" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
cd "$REPO"
python - "$OUT" <<'PY'
import csv, sys, glob
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("psk::", "").replace("void ", "")
        print(f"{n[:120]:120s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} min_us={float(r['MinNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
