#!/bin/bash
# Round 4: does pass 1 at 1024 / 2048 slices thrash the XCD L2 with half-written lines?  Every (slice, workgroup) segment keeps one
# partially written 128-byte line open between two tiles of its workgroup: 32 workgroups per XCD x B slices x 128 B = 4 MiB (B = 1024:
# the whole L2) or 8 MiB (B = 2048).  Fewer pass-1 workgroups (option scatter_workgroups) shrink that footprint at the price of idle CUs.
#   scripts/ab_l2_open_lines.sh -> gpurun_out/ab_l2_open_lines.txt   (run on the GPU box)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/ab_l2_open_lines.txt
: > $OUT
for spec in "cbf_add 10000000" "bloom31_add 33554432" "bloom_add 10000000"; do
  set -- $spec
  for w in 0 96 128 160 192 224; do
    PSK_OPTIONS=scatter_workgroups=$w scripts/trace_op.sh $1 $2 4 > /dev/null 2>&1
    echo "== $1 scatter_workgroups=$w" >> $OUT
    grep -E "k_part_scatter|k_nib_apply|k_bloom_apply" gpurun_out/trace_$1.txt | cut -c1-60,140-200 >> $OUT
  done
done
cat $OUT
