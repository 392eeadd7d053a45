#!/bin/bash
# kernel-trace of scripts/ab_cms_check_lib.py under one package path: per-kernel avg time + launch shape (grid, workgroup, LDS, VGPRs)
P=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/abtrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $REPO/scripts/ab_cms_check_lib.py $REPO/$P > $OUT.log 2>&1
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0, None])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("psk::", "").replace("void ", "")[:60]
        a = agg[n]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a[2] = {k: r[k] for k in r if k in ("Workgroup_Size_X", "Grid_Size_X", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]:
    print(f"{n:60s} calls={a[0]:4d} avg_us={a[1]/a[0]:8.1f} {a[2]}")
PY
