#!/usr/bin/env bash
# scripts/gpu_r04_evidence.sh — run ON THE GPU BOX (gpurun): the round-4 evidence that needs no new kernel code.
#   1. batch_kernel phase stamps UNDER LOAD (diagnostic build; 1000 trials in flight), both workgroup shapes, Beta.c grid and WithNOMA config 3
#   2. rocprofv3 counter passes of the 1000-trial Beta.c grid (configs[4]'s regime), both workgroup shapes; config 3 with two trials per CU
#   3. lcluster_kernel<false> / <true>: per-workgroup phase stamps (diagnostic build) + SQ / LDS / VMEM counter passes of the bench command
set -euo pipefail
OUT=gpurun_out/r04
mkdir -p "$OUT"
DIAG="$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_diag.so"
for v in 0 1; do for w in 8 16; do
  PRACH_LIB=$DIAG python3 scripts/gpu_batch_stamps.py 100 $v $w > "$OUT/stamps_v${v}_w${w}.txt" 2>&1 || { echo "stamps v$v w$w failed"; tail -5 "$OUT/stamps_v${v}_w${w}.txt"; }
  head -1 "$OUT/stamps_v${v}_w${w}.txt"
done; done
PRACH_ENG_OPTS=batch_waves=8  scripts/gpu_pmc.sh r04 grid8  -- scripts/gpu_batch.py 100 0 0
PRACH_ENG_OPTS=batch_waves=16 scripts/gpu_pmc.sh r04 grid16 -- scripts/gpu_batch.py 100 0 0
PRACH_ENG_OPTS=batch_waves=8 PMC_GROUPS="fetch write sq tcc tcpw" scripts/gpu_pmc.sh r04 c3w8 -- scripts/gpu_batch.py 100 1 0
PRACH_LIB=$DIAG PRACH_PRINT_STAMPS=2 python3 tests/tools/gpu_single.py "cluster=32" 0 100000 > "$OUT/lcluster_philox_stamps.txt" 2>&1 || echo "lcluster philox stamps failed"
PRACH_LIB=$DIAG PRACH_PRINT_STAMPS=2 python3 tests/tools/gpu_single.py "cluster=32" 0 100000 - glibc > "$OUT/lcluster_glibc_stamps.txt" 2>&1 || echo "lcluster glibc stamps failed"
tail -1 "$OUT/lcluster_philox_stamps.txt"; tail -1 "$OUT/lcluster_glibc_stamps.txt"
PMC_GROUPS="sq lds vmem fetch write" scripts/gpu_pmc.sh r04 lc -- bench.py --steps 3 --warmup 1 --no-cpu --no-extras
echo done
