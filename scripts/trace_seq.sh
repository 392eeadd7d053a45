#!/bin/bash
# per-dispatch kernel durations, in launch order, of the LAST iteration of one operation (run on the GPU box):
#   scripts/trace_seq.sh <op> [n] [iters]  (PSK_OPTS=name=value,... sets engine options) -> gpurun_out/seq_<op>.txt
set -u
OP=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/seq_$OP
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python "$REPO/scripts/prof_ops.py" $OP "$@" > "$OUT.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY' > "$OUT.txt"
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-40:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    n = r["Kernel_Name"].replace("psk::", "").replace("void ", "")
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?'):>8s}  {n[:110]}")
PY
rm -rf "$OUT"
cat "$OUT.txt"
