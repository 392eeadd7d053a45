#!/bin/bash
# same-box comparison of (library, part_debug) pairs: scripts/ab_dbg.sh "libA.so:0 libC.so:16" [rounds]
R=${2:-3}
for r in $(seq $R); do
  for P in $1; do
    L=${P%%:*}; D=${P##*:}
    PSK_LIB_PATH=$L PSK_PART_DEBUG=$D python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-detail 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], 'dbg=$D', 'value=%.0f insert=%.0f check=%.0f Mkeys/s step=%.3f ms' % (d['value'], d['detail']['insert_Mkeys_s'], d['detail']['check_Mkeys_s'], d['ms_per_step']))"
  done
done
