set -e
python3 tests/tools/gpu_single.py "cluster=32;cluster=16" 0 100000 check 2>&1 | tail -2
python3 tests/tools/gpu_single.py "cluster=32" 1 100000 check 2>&1 | tail -1
python3 tests/tools/gpu_single.py "cluster=32" 0 20000 check 2>&1 | tail -1
python3 tests/tools/gpu_fuzz_lean.py 41 200 2>&1 | grep -v "^\[prach\]" | tail -3
python3 tests/tools/gpu_fuzz_lean.py 42 20 big 2>&1 | grep -v "^\[prach\]" | tail -3
