#!/usr/bin/env bash
# scripts/gpu_shape_probe.sh — run ON THE GPU BOX: batch_kernel's two workgroup shapes (512 threads x two trials per CU, 1024 threads x one) against the number of
# trials in the launch, Beta.c and WithNOMA sweeps — what the engine's per-launch choice (prach_engine.hip) is tuned on.
set -uo pipefail
for times in 3 6 13 26 51 100 200; do for v in 0 1; do
  line="trials=$((times * 10)) variant=$v"
  for w in 8 16; do
    r=$(PRACH_ENG_OPTS=batch_waves=$w timeout -k 10 300 python3 scripts/gpu_batch.py $times $v 1 2>&1 | grep -o "kernel=[0-9.]*ms" | head -1)
    line="$line w$w:$r"
  done
  echo "$line"
done; done
