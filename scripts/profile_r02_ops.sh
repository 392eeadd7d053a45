#!/bin/bash
# Round-2 side evidence (run on the GPU box via gpurun): kernel-trace stats per operation, the SQ counters of the bench
# kernels and the pass-1 ablation / phase profile (knobs build) -> gpurun_out/r02/; copy what is cited into profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02
mkdir -p "$OUT"
cd "$REPO"
for spec in "bloom31_add 33554432 5" "bloom31_check 33554432 5" "bloom_check_fresh 10000000 10" "cbf25_check 10000000 10" "cms_add 10000000 10" "cms_check 10000000 10"; do
  set -- $spec
  scripts/trace_op.sh $1 $2 $3 > /dev/null 2>&1
  cp gpurun_out/trace_$1.txt "$OUT/rocprofv3_kernel_stats_$1.txt"
done
scripts/profile_sq.sh r02 > "$OUT/sq_counters.txt" 2>&1
python scripts/ablate.py > "$OUT/ablation_phase_profile.txt" 2>&1
ls -la "$OUT"
