set -e
for X in "" x4; do
for W in 8 16; do
echo "variant=$X waves=$W"
if [ -n "$X" ]; then export PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_$X.so; fi
PRACH_ENG_OPTS=batch_waves=$W python3 scripts/gpu_batch.py 100 1 0 2>&1 | head -1
PRACH_ENG_OPTS=batch_waves=$W python3 scripts/gpu_batch.py 100 0 0 2>&1 | head -1
done
done
