#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes (each in its own run) for bench.py.
# usage: scripts/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{kernel_stats.txt,pmc_summary.txt,...}
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python "$REPO/bench.py" $ARGS > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$N" -o pmc -- python "$REPO/bench.py" $ARGS > "$OUT/bench_under_pmc_$N.json" 2> "$OUT/pmc_$N.err" || echo "pmc $C failed" >> "$OUT/errors.txt"
done
cd "$REPO"
python scripts/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
find "$OUT" -name "*.err" -delete
cat "$OUT/summary.txt" | cut -c1-250 | head -150
