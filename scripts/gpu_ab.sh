#!/usr/bin/env bash
# scripts/gpu_ab.sh LIB... — run ON THE GPU BOX: the two batched regimes (1000 Beta.c trials, 1000 WithNOMA trials), both workgroup shapes, for every
# library variant given (paths relative to 5g-nr-randomaccess_amd/), twice each, on the SAME box — the only fair comparison (boxes differ by several percent).
set -uo pipefail
for rep in 1 2; do for lib in "$@"; do for v in 0 1; do for w in 8 16; do
  r=$(PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/$lib PRACH_ENG_OPTS=batch_waves=$w timeout -k 10 120 python3 scripts/gpu_batch.py 100 $v 0 2>&1 | grep -o "digest=[0-9a-f]*\|kernel=[0-9.]*ms" | tr '\n' ' ')
  echo "$lib v=$v w=$w $r"
done; done; done; done
