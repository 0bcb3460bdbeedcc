#!/usr/bin/env bash
# scripts/gpu_round_checks.sh — run ON THE GPU BOX (gpurun): the GPU parity suite plus every fuzzer on the build that is in the tree
# (general kernels in both RNG modes, the lean cluster kernel, big trials, the batch kernel in both workgroup shapes, in the reference's stream and with the
# 128-entry-queue test build).  Every fuzzer prints "... N bad"; the lean / batch ones also how many trials left their kernel.
set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 tests/tools/gpu_fuzz.py 61 400 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_lean.py 62 200 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_lean.py 65 30 big 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_big.py 63 20 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_batch.py 64 60 2>&1 | grep -v "^\[prach\]" | tail -1
PRACH_ENG_OPTS=batch_waves=8 python3 tests/tools/gpu_fuzz_batch.py 66 40 2>&1 | grep -v "^\[prach\]" | tail -1
PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_tinyq.so python3 tests/tools/gpu_fuzz_batch.py 67 40 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_batch.py 68 10 big 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_batch.py 70 60 small glibc 2>&1 | grep -v "^\[prach\]" | tail -1
PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_tinyq.so python3 tests/tools/gpu_fuzz_batch.py 77 30 small glibc 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_noma.py 69 60 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_noma.py 78 60 glibc 2>&1 | grep -v "^\[prach\]" | tail -1
