#!/bin/bash
# kernel-trace stats of one operation (run on the GPU box): scripts/trace_op.sh <op> [n] [iters] -> gpurun_out/trace_<op>.txt
set -u
OP=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_$OP
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python "$REPO/scripts/prof_ops.py" $OP "$@" > "$OUT.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY' > "$OUT.txt"
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("psk::", "").replace("void ", "")
        print(f"{n[:140]:140s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
rm -rf "$OUT"
tail -3 "$OUT.log"; head -12 "$OUT.txt"
