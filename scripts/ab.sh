#!/bin/bash
# same-box A/B of two engine builds: scripts/ab.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do
  for L in $A $B; do
    PSK_LIB_PATH=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-detail 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], 'value=%.0f insert=%.0f check=%.0f Mkeys/s step=%.3f ms' % (d['value'], d['detail']['insert_Mkeys_s'], d['detail']['check_Mkeys_s'], d['ms_per_step']))"
  done
done
