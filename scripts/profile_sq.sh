#!/bin/bash
# SQ (shader sequencer) counters of the bench kernels: VALU / LDS instruction counts, busy and wait cycles, LDS bank
# conflicts.  One rocprofv3 --pmc pass per counter group (run on the GPU box via gpurun).
set -u
TAG=${1:-sq}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --spinup 0 --no-cpu-baseline --no-detail --no-extra-configs"
for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-60)
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$N" -o pmc -- python "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/$N.err" || echo "pmc $C failed" >> "$OUT/errors.txt"
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "k_part_scatter" in k or "k_bloom_apply" in k or "k_bloom_test" in k:
            short = ("scatter keyed" if "PayKeyId" in k else "scatter insert") if "scatter" in k else ("apply" if "apply" in k else "test")
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("==", k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
