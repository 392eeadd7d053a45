#!/bin/bash
# scripts/ab_pass1.sh "<tag>|<lib or ->|<ENV=..> <ENV=..>" ...   (run on the GPU box: every variant on the same box, twice, interleaved)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
for round in 1 2; do
  for spec in "$@"; do
    IFS='|' read -r tag lib envs <<< "$spec"
    if [ "$lib" = "-" ]; then lib=$REPO/pyprobables_amd/csrc/libpsk_hip.so; else lib=$REPO/$lib; fi
    env AB_TAG="$tag" PSK_LIB_PATH="$lib" $envs timeout 600 python scripts/ab_pass1.py 2>&1 | grep -v "^$" | tail -4
  done
done
