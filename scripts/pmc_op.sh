#!/bin/bash
# HBM-side traffic and L2 hit rate of one operation from the PMC counters (run on the GPU box): FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (MI355X_MICROARCH.md: the TCC block cannot hold both), TCC_HIT_sum + TCC_MISS_sum in a third, kernel-trace
# only.   scripts/pmc_op.sh <op> [n] [iters]
# -> gpurun_out/pmc_<op>.json   (units: the counters are in KiB; FETCH_SIZE is doubled -- on gfx950 it reports half of a
# wide coalesced streaming read, calibrated on k_gen_keys16 / the key stream)
set -u
OP=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$OP
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$D" -o pmc -- python "$REPO/scripts/prof_ops.py" $OP "$@" > "$OUT/$D.log" 2>&1 || echo "pmc $C failed" >> "$OUT/errors.txt"
done
cd "$REPO"
python - "$OUT" "$OP" <<'PY' > "$OUT.json"
import csv, glob, json, re, sys
out, op = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
launches, n = None, None
for line in open(out + "/FETCH_SIZE.log"):
    m = re.search(r"launches=(\d+) n=(\d+)", line)
    if m:
        launches, n = int(m.group(1)), int(m.group(2))
res = {"op": op, "keys": n, "launches_counted": launches, "kernels": {}}
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
for c in tot:
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("psk::", "").replace("void ", "")
            if not k.startswith(("k_part", "k_bloom", "k_counter", "k_lookup", "k_weight_sum", "k_apply", "k_cbf", "k_nib", "k_tally", "k_win", "k_pack", "__amd_rocclr")):
                continue
            short = re.sub(r"\(.*", "", k)
            short = re.sub(r"KeysFixed16, |Spill\w+(<\w+>)?, ", "", short)
            d = res["kernels"].setdefault(short, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "dispatches": 0, "TCC_HIT_sum": 0.0, "TCC_MISS_sum": 0.0})
            d[c] += float(r["Counter_Value"] or 0)
            if c == "FETCH_SIZE":
                d["dispatches"] += 1
            tot[c] += float(r["Counter_Value"] or 0)
hit = miss = 0.0
for f in glob.glob(f"{out}/TCC_HIT_sum_TCC_MISS_sum/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("psk::", "").replace("void ", "")
        short = re.sub(r"KeysFixed16, |Spill\w+(<\w+>)?, ", "", re.sub(r"\(.*", "", k))
        if short in res["kernels"] and r["Counter_Name"] in ("TCC_HIT_sum", "TCC_MISS_sum"):
            res["kernels"][short][r["Counter_Name"]] += float(r["Counter_Value"] or 0)
for d in res["kernels"].values():
    h, m = d.pop("TCC_HIT_sum"), d.pop("TCC_MISS_sum")
    hit, miss = hit + h, miss + m
    d["l2_hit"] = round(h / (h + m), 4) if h + m else None
    d["l2_requests_per_launch"] = round((h + m) / launches, 1)
res["l2_hit"] = round(hit / (hit + miss), 4) if hit + miss else None
for d in res["kernels"].values():
    d["fetch_KiB_per_launch"] = round(d.pop("FETCH_SIZE") / launches, 1)
    d["write_KiB_per_launch"] = round(d.pop("WRITE_SIZE") / launches, 1)
    d["dispatches_per_launch"] = round(d.pop("dispatches") / launches, 2)
res["hbm_bytes_per_launch"] = int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / launches)
res["bytes_per_key"] = round(res["hbm_bytes_per_launch"] / n, 1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT"
cat "$OUT.json" | head -40
