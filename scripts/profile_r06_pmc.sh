#!/bin/bash
# Round-6 counter evidence (run on the GPU box via gpurun): PMC traffic + L2 hit rate of the operations the timed steps RUN NOW --
# Bloom lookups pinned to the tile-flag scheme (all present) / lazy gathers (all absent), the warmed automatic choice on a half-absent batch,
# cfg 4's stream through the DEFAULT API (update windows: k_win_fold) and its borrow_keys variant -- next to the unchanged operations.
# -> gpurun_out/r06/pmc_<op>.json ; scripts/make_pmc_json.py gpurun_out/r06 r06 turns them into profiles/r06_pmc_traffic.json + r06_l2_hit.json
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
OPS=${*:-bloom_add bloom_check bloom_check_fresh bloom_check_half cms_add cms_check cbf_add cbf_check cbf_remove}
for op in $OPS; do
  scripts/pmc_op.sh $op 10000000 5 > /dev/null 2>&1
  cp gpurun_out/pmc_$op.json "$OUT/"
done
if [ $# -eq 0 ]; then
  scripts/pmc_op.sh cbf_check_kept 10000000 40 > /dev/null 2>&1
  cp gpurun_out/pmc_cbf_check_kept.json "$OUT/"
  for op in bloom31_add bloom31_check; do
    scripts/pmc_op.sh $op 33554432 3 > /dev/null 2>&1
    cp gpurun_out/pmc_$op.json "$OUT/"
  done
  scripts/pmc_op.sh cfg4_stream 1000000 3 > /dev/null 2>&1
  cp gpurun_out/pmc_cfg4_stream.json "$OUT/"
  PSK_CFG4_MODE=borrow_window scripts/pmc_op.sh cfg4_stream 1000000 3 > /dev/null 2>&1
  cp gpurun_out/pmc_cfg4_stream.json "$OUT/pmc_cfg4_stream_borrow.json"
fi
python scripts/make_pmc_json.py "$OUT" r06 > "$OUT/make_pmc_json.log" 2>&1
cat "$OUT/make_pmc_json.log"
