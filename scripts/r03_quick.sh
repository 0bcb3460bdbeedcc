set -e
python3 tests/tools/gpu_glibc_time.py 2>&1 | grep "xcd_pack=1" | grep glibc | head -3
python -m pytest tests -m gpu -x -q -k "reproduces_reference_files or random_flags or xcd_packed" 2>&1 | tail -3
python3 tests/tools/gpu_fuzz.py 72 300 2>&1 | grep -v "^\[prach\]" | tail -1
