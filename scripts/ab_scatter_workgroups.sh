#!/bin/bash
# Round 4: pass 1 under fewer workgroups (option scatter_workgroups), three geometries (1024, 2048, 256 slices).
# Hypothesis tested and REFUTED: at 1024 / 2048 slices every (slice, workgroup) segment keeps one partially written 128-byte line open
# between two tiles of its workgroup -- 32 workgroups per XCD x B slices x 128 B = 4 MiB (B = 1024: the whole L2) or 8 MiB (B = 2048) -- so
# fewer workgroups might let the lines merge in L2.  Measured: pass 1 slows roughly in proportion to the workgroups taken away at every
# geometry (profiles/r04_ab_scatter_workgroups.txt); per-workgroup work, not the L2, is what 1024+ slices cost.  The same table bounds what a
# fused scatter + apply kernel could gain: the Bloom insert's pass 1 with ONE 512-thread workgroup per CU.
#   scripts/ab_scatter_workgroups.sh -> gpurun_out/ab_scatter_workgroups.txt   (run on the GPU box)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/ab_scatter_workgroups.txt
: > $OUT
for spec in "cbf_add 10000000" "bloom31_add 33554432" "bloom_add 10000000"; do
  set -- $spec
  for w in 0 96 128 160 192 224 256; do
    PSK_OPTIONS=scatter_workgroups=$w scripts/trace_op.sh $1 $2 4 > /dev/null 2>&1
    echo "== $1 scatter_workgroups=$w" >> $OUT
    grep -E "k_part_scatter|k_nib_apply|k_bloom_apply" gpurun_out/trace_$1.txt | cut -c1-60,140-200 >> $OUT
  done
done
cat $OUT
