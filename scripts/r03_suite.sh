set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 tests/tools/gpu_fuzz.py 61 400 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_lean.py 62 200 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_big.py 63 20 2>&1 | grep -v "^\[prach\]" | tail -1
python3 tests/tools/gpu_fuzz_batch.py 64 60 2>&1 | grep -v "^\[prach\]" | tail -1
