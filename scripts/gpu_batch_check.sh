#!/usr/bin/env bash
# scripts/gpu_batch_check.sh — run ON THE GPU BOX: the batch kernel's quick loop — timing of the two batched regimes in both workgroup shapes (with the
# result digest), then its fuzzers (Philox, both shapes, the small-list test build, the reference's stream).  Exits non-zero if a fuzzer reports a mismatch.
set -uo pipefail
OUT=gpurun_out/r04
mkdir -p "$OUT"
for v in 0 1; do for w in 8 16; do
  PRACH_ENG_OPTS=batch_waves=$w timeout -k 10 120 python3 scripts/gpu_batch.py 100 $v 0 2>&1 | grep -v "^\[prach\]" | head -1
done; done
fail=0
fz() { # name, env..., -- args
  local name="$1"; shift
  ( "$@" ) > "$OUT/fuzz_$name.log" 2>&1
  local last; last=$(grep -v "^\[prach\]" "$OUT/fuzz_$name.log" | tail -1)
  echo "$name: $last"
  [[ "$last" == *" 0 bad"* ]] || fail=1
}
fz batch        timeout -k 10 300 python3 tests/tools/gpu_fuzz_batch.py 64 ${FUZZ_N:-40}
fz batch_w8     env PRACH_ENG_OPTS=batch_waves=8 timeout -k 10 300 python3 tests/tools/gpu_fuzz_batch.py 66 ${FUZZ_N:-40}
fz batch_tinyq  env PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_tinyq.so timeout -k 10 300 python3 tests/tools/gpu_fuzz_batch.py 67 ${FUZZ_N:-40}
fz batch_big    timeout -k 10 300 python3 tests/tools/gpu_fuzz_batch.py 68 8 big
fz batch_glibc  timeout -k 10 300 python3 tests/tools/gpu_fuzz_batch.py 70 ${FUZZ_N:-40} small glibc
fz batch_glibc_tinyq env PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_tinyq.so timeout -k 10 300 python3 tests/tools/gpu_fuzz_batch.py 77 20 small glibc
exit $fail
