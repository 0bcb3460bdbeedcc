set -e
python -m pytest tests -m gpu -x -q -k "fallback or overflow or sector or lean_cluster or residency or two_engines" 2>&1 | tail -4
python3 tests/tools/gpu_fuzz_lean.py 31 300 2>&1 | grep -v "^\[prach\]" | tail -2
python3 tests/tools/gpu_fuzz_lean.py 32 30 big 2>&1 | grep -v "^\[prach\]" | tail -2
python3 tests/tools/gpu_fuzz_batch.py 33 80 2>&1 | grep -v "^\[prach\]" | tail -2
python3 tests/tools/gpu_fuzz.py 34 200 2>&1 | grep -v "^\[prach\]" | tail -2
