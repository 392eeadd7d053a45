#!/bin/bash
# SQ counters of the kernels of ONE operation (scripts/prof_ops.py <op>): scripts/profile_sq_op.sh <op> [n] [iters] -> gpurun_out/sq_<op>.txt
# (one rocprofv3 --pmc pass per counter group, kernel-trace only; run on the GPU box via gpurun)
set -u
OP=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sq_$OP
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-60)
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$N" -o pmc -- python "$REPO/scripts/prof_ops.py" $OP "$@" > /dev/null 2> "$OUT/$N.err" || echo "pmc $C failed" >> "$OUT/errors.txt"
done
cd "$REPO"
python - "$OUT" <<'PY' > "$OUT.txt"
import csv, sys, glob, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").replace("psk::", "").replace("void ", "")
        if not k.startswith("k_"):
            continue
        short = re.sub(r"\(.*", "", k)[:90]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("==", k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
rm -rf "$OUT"
cat "$OUT.txt"
