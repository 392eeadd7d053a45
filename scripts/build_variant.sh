#!/bin/bash
# A/B helper: scripts/build_variant.sh NAME "-DFOO=1 ..." -> ab/libpsk_NAME.so
# Recompiles only the power-of-two Bloom launcher units (insert + lookup: the bench path of cfg 2 / cfg 5; UNITS="psk_part_cms ..." names others) with the extra
# flags and links them with the shipped build's other objects (python -m pyprobables_amd.build first).  Load the result
# through PSK_LIB_PATH (scripts/ab.sh).
set -e
NAME=$1; shift
EXTRA="$*"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/pyprobables_amd/csrc
OUT=$ROOT/ab/$NAME
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fvisibility-inlines-hidden"
pids=()
for u in ${UNITS:-psk_part_bloom_add psk_part_bloom_check}; do
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -DPSK_TU_POW2=1 -c $CSRC/$u.hip -o $OUT/${u}_v1.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
objs=""
for o in $CSRC/build/*.o; do
  b=$(basename $o)
  if [ -f "$OUT/$b" ]; then objs="$objs $OUT/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -no-hip-rt -o $ROOT/ab/libpsk_$NAME.so $objs
echo $ROOT/ab/libpsk_$NAME.so
