#!/bin/bash
# Round-6 evidence (run on the GPU box via gpurun): kernel-trace stats of the default bench's own kernels and of every configuration, of the
# single operations, the PMC traffic / L2 passes (scripts/profile_r06_pmc.sh) and the bench lines -> gpurun_out/r06/
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
# (the headline's own kernels: the extra configurations of the default line would mix their launches into the averages)
scripts/trace_bench.sh r06_cfg2 --steps 20 --warmup 5 --no-extra-configs > /dev/null 2>&1
cp gpurun_out/trace_bench_r06_cfg2.txt "$OUT/rocprofv3_kernel_stats_cfg2.txt"
for c in cfg3 cfg4 cfg5; do
  scripts/trace_bench.sh r06_$c --config $c --steps 3 --warmup 1 --spinup 0.2 > /dev/null 2>&1
  cp gpurun_out/trace_bench_r06_$c.txt "$OUT/rocprofv3_kernel_stats_$c.txt"
done
for op in cbf_check cbf_check_kept cbf_add cbf_remove cms_check cms_add bloom_check_fresh bloom_check_half; do
  scripts/trace_op.sh $op 10000000 $([ $op = cbf_check_kept ] && echo 40 || echo 10) > /dev/null 2>&1
  cp gpurun_out/trace_$op.txt "$OUT/rocprofv3_kernel_stats_$op.txt"
done
scripts/profile_r06_pmc.sh > "$OUT/pmc.log" 2>&1
python bench.py --steps 20 --warmup 5 > "$OUT/bench_cfg2_default.json" 2> /dev/null
cp gpurun_out/bench_detail.json "$OUT/bench_cfg2_default_detail.json"
for c in cfg3 cfg4 cfg5; do python bench.py --config $c --no-cpu-baseline > "$OUT/bench_$c.json" 2> /dev/null; done
PSK_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-detail > "$OUT/bench_cfg2_forced_dist_1rank.json" 2> /dev/null
ls -la "$OUT"
