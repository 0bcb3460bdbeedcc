#!/usr/bin/env bash
# scripts/gpu_round_checks.sh — run ON THE GPU BOX (gpurun): the GPU parity suite plus every fuzzer on the build that is in the tree (general kernels in both
# RNG modes, the lean cluster kernel, big trials, the batch kernel in both workgroup shapes, in the reference's stream and with the small-list test build, NOMA.c
# in both streams).  Every step's full log goes to gpurun_out/checks/; the script exits non-zero if pytest fails or a fuzzer's last line does not say " 0 bad".
set -uo pipefail
OUT=gpurun_out/checks
mkdir -p "$OUT"
fail=0
if [ -z "${SKIP_PYTEST:-}" ]; then
  python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1 || fail=1
  tail -2 "$OUT/pytest.log"
fi
fz() { # name, command...
  local name="$1"; shift
  ( "$@" ) > "$OUT/fuzz_$name.log" 2>&1
  local rc=$?
  local last; last=$(grep -v "^\[prach\]" "$OUT/fuzz_$name.log" | tail -1)
  echo "$name: $last"
  [[ $rc -eq 0 && "$last" == *" 0 bad"* ]] || { echo "  ^ FAILED (exit $rc)"; fail=1; }
}
TQ=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_tinyq.so
N=${FUZZ_SCALE:-1}
S=${FUZZ_SEED_OFFSET:-0}   # (a campaign with fresh seeds: FUZZ_SEED_OFFSET=1000 FUZZ_SCALE=2 SKIP_PYTEST=1)
fz general      timeout -k 10 900 python3 tests/tools/gpu_fuzz.py $((61 + S)) $((400 * N))
fz lean         timeout -k 10 900 python3 tests/tools/gpu_fuzz_lean.py $((62 + S)) $((200 * N))
fz lean_big     timeout -k 10 900 python3 tests/tools/gpu_fuzz_lean.py $((65 + S)) $((30 * N)) big
fz big          timeout -k 10 900 python3 tests/tools/gpu_fuzz_big.py $((63 + S)) $((20 * N))
fz batch        timeout -k 10 900 python3 tests/tools/gpu_fuzz_batch.py $((64 + S)) $((60 * N))
fz batch_w8     env PRACH_ENG_OPTS=batch_waves=8 timeout -k 10 900 python3 tests/tools/gpu_fuzz_batch.py $((66 + S)) $((40 * N))
fz batch_tinyq  env PRACH_LIB=$TQ timeout -k 10 900 python3 tests/tools/gpu_fuzz_batch.py $((67 + S)) $((40 * N))
fz batch_big    timeout -k 10 900 python3 tests/tools/gpu_fuzz_batch.py $((68 + S)) $((10 * N)) big
fz batch_glibc  timeout -k 10 900 python3 tests/tools/gpu_fuzz_batch.py $((70 + S)) $((60 * N)) small glibc
fz batch_glibc_tinyq env PRACH_LIB=$TQ timeout -k 10 900 python3 tests/tools/gpu_fuzz_batch.py $((77 + S)) $((30 * N)) small glibc
fz noma         timeout -k 10 900 python3 tests/tools/gpu_fuzz_noma.py $((69 + S)) $((60 * N))
fz noma_glibc   timeout -k 10 900 python3 tests/tools/gpu_fuzz_noma.py $((78 + S)) $((60 * N)) glibc
exit $fail
