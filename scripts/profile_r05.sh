#!/bin/bash
# Round-5 evidence (run on the GPU box via gpurun): kernel-trace stats of the default bench, PMC traffic AND L2 hit rate per
# operation (FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum+TCC_MISS_sum in separate passes), incl. cfg 4's stream and the m = 2^31
# kernels of cfg 5, and the bench lines of every configuration -> gpurun_out/r05/
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
# (the headline's own kernels: the extra configurations of the default line would mix their launches into the averages)
scripts/trace_bench.sh r05_cfg2 --steps 20 --warmup 5 --spinup 0.2 --no-extra-configs > /dev/null 2>&1
cp gpurun_out/trace_bench_r05_cfg2.txt "$OUT/rocprofv3_kernel_stats_cfg2.txt"
for c in cfg3 cfg4 cfg5; do
  scripts/trace_bench.sh r05_$c --config $c --steps 3 --warmup 1 --spinup 0.2 > /dev/null 2>&1
  cp gpurun_out/trace_bench_r05_$c.txt "$OUT/rocprofv3_kernel_stats_$c.txt"
done
for op in cbf_check cbf_check_kept cbf_add cbf_remove cms_check bloom_check_fresh; do
  scripts/trace_op.sh $op 10000000 $([ $op = cbf_check_kept ] && echo 40 || echo 5) > /dev/null 2>&1
  cp gpurun_out/trace_$op.txt "$OUT/rocprofv3_kernel_stats_$op.txt"
done
for op in bloom_add bloom_check bloom_check_fresh cms_add cms_check cbf_add cbf_check cbf_remove; do
  scripts/pmc_op.sh $op 10000000 5 > /dev/null 2>&1
  cp gpurun_out/pmc_$op.json "$OUT/"
done
# (40 launches: the two set-up lookups that build the kept images read the whole table and are averaged in -- ~ +8 %)
scripts/pmc_op.sh cbf_check_kept 10000000 40 > /dev/null 2>&1
cp gpurun_out/pmc_cbf_check_kept.json "$OUT/"
for op in bloom31_add bloom31_check; do
  scripts/pmc_op.sh $op 33554432 3 > /dev/null 2>&1
  cp gpurun_out/pmc_$op.json "$OUT/"
done
scripts/pmc_op.sh cfg4_stream 1000000 3 > /dev/null 2>&1
cp gpurun_out/pmc_cfg4_stream.json "$OUT/"
python scripts/make_pmc_json.py "$OUT" r05 > "$OUT/make_pmc_json.log" 2>&1
python bench.py > "$OUT/bench_cfg2_default.json" 2> /dev/null
python bench.py --steps 20 --warmup 5 > "$OUT/bench_cfg2_steps20.json" 2> /dev/null
for c in cfg3 cfg4 cfg5; do python bench.py --config $c --no-cpu-baseline > "$OUT/bench_$c.json" 2> /dev/null; done
python bench.py --config cfg4 --no-combine --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/bench_cfg4_nocombine.json" 2> /dev/null
PSK_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-detail > "$OUT/bench_cfg2_forced_dist_1rank.json" 2> /dev/null
ls -la "$OUT"
