#!/bin/bash
# kernel-trace stats of the per-key loop (scripts/latency_probe.py) on the GPU box -> gpurun_out/trace_per_key.txt: which kernels a
# value-returning single-key call runs, and how long they take
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_per_key
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python "$REPO/scripts/latency_probe.py" 0123456789abcdef > "$OUT.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY' > "$OUT.txt"
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("psk::", "").replace("void ", "")
        print(f"{n[:140]:140s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f}")
PY
rm -rf "$OUT"
head -20 "$OUT.txt"
