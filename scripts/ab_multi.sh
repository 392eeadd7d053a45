#!/bin/bash
# same-box A/B of several engine builds (scripts/build_variant.sh): scripts/ab_multi.sh ROUNDS lib1.so lib2.so ...
R=$1; shift
for r in $(seq $R); do
  for L in "$@"; do
    PSK_LIB_PATH=$L python bench.py --steps 100 --warmup 5 --spinup 0.3 --no-cpu-baseline --no-detail 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-18s value=%.0f insert=%.0f check=%.0f Mkeys/s step=%.4f ms ok=%s' % ('$L'.split('/')[-1], d['value'], d['detail']['insert_Mkeys_s'], d['detail']['check_Mkeys_s'], d['ms_per_step'], d['detail']['all_inserted_found']))"
  done
done
