set -e
for G in 0 2 4; do
echo "cluster=$G"
PRACH_ENG_OPTS=cluster=$G python3 scripts/gpu_noma_batch.py batch 10 2>&1 | tail -1 | cut -c1-260
done
