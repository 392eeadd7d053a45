#!/bin/bash
# kernel-trace stats of a bench.py run (on the GPU box): scripts/trace_bench.sh <tag> [bench args] -> gpurun_out/trace_bench_<tag>.txt
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_bench_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY' > "$OUT.txt"
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("psk::", "").replace("void ", "")
        print(f"{n[:150]:150s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:10.2f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
rm -rf "$OUT"
tail -2 "$OUT.log" | cut -c1-300; head -14 "$OUT.txt"
