set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python3 tests/tools/gpu_fuzz.py 21 300 2>&1 | tail -3
python3 tests/tools/gpu_fuzz_lean.py 22 150 2>&1 | tail -3
python3 tests/tools/gpu_fuzz_batch.py 23 60 2>&1 | grep -v "^\[prach\]" | tail -2
PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_tinyq.so python3 tests/tools/gpu_fuzz_batch.py 24 40 2>&1 | grep -v "^\[prach\]" | tail -2
PRACH_ENG_OPTS=batch_waves=8 python3 tests/tools/gpu_fuzz_batch.py 25 40 2>&1 | grep -v "^\[prach\]" | tail -2
python3 tests/tools/gpu_fuzz_batch.py 26 10 big 2>&1 | grep -v "^\[prach\]" | tail -2
