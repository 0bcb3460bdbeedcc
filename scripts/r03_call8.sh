set -e
mkdir -p gpurun_out/r03
python3 scripts/gpu_batch.py 100 1 0 2>&1 | tee gpurun_out/r03/c3_batch_v1.log
python3 scripts/gpu_batch.py 100 0 0 2>&1 | tee gpurun_out/r03/grid100_batch_v1.log
python3 bench.py --gpus 1 --workload grid --times 1000 --steps 1 --warmup 0 > gpurun_out/r03/grid1000_batch_v1.json 2>gpurun_out/r03/grid1000_batch_v1.err || tail -5 gpurun_out/r03/grid1000_batch_v1.err
python3 -c "
import json;d=json.load(open('gpurun_out/r03/grid1000_batch_v1.json'));print('grid1000 value',d['value'],'ms',d['ms_per_step'])"
python3 tests/tools/gpu_fuzz_batch.py 1 60 2>&1 | tail -5
PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_tinyq.so python3 tests/tools/gpu_fuzz_batch.py 2 40 2>&1 | tail -5
python3 tests/tools/gpu_fuzz_batch.py 3 12 big 2>&1 | tail -5
