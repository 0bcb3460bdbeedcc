#!/usr/bin/env bash
# scripts/gpu_ab8.sh LIB... — run ON THE GPU BOX: like gpu_ab.sh, the 512-thread shape only (the one the engine takes for 1000 trials).
set -uo pipefail
for rep in 1 2; do for lib in "$@"; do for v in 0 1; do
  r=$(PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/$lib PRACH_ENG_OPTS=batch_waves=8 timeout -k 10 120 python3 scripts/gpu_batch.py 100 $v 0 2>&1 | grep -o "digest=[0-9a-f]*\|kernel=[0-9.]*ms" | tr '\n' ' ')
  echo "$lib v=$v w=8 $r"
done; done; done
