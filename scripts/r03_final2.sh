set -e
mkdir -p gpurun_out/r03
scripts/gpu_profile.sh r03 > gpurun_out/r03/profile.log 2>&1 || { tail -20 gpurun_out/r03/profile.log; exit 1; }
tail -3 gpurun_out/r03/profile.log
scripts/gpu_pmc.sh r03c c3 -- scripts/gpu_batch.py 100 1 0 > gpurun_out/r03/pmc_c3.log 2>&1
tail -2 gpurun_out/r03/pmc_c3.log
python3 scripts/gpu_batch.py 100 1 0 > gpurun_out/r03/c3_final.log 2>&1; cat gpurun_out/r03/c3_final.log
python3 bench.py --gpus 1 --workload grid --times 1000 --steps 1 --warmup 0 > gpurun_out/r03/grid1000_final.json 2>gpurun_out/r03/grid1000_final.err
python3 -c "
import json;d=json.load(open('gpurun_out/r03/grid1000_final.json'));print('grid1000 value',d['value'],'ms',d['ms_per_step'], d['roofline']['own_bytes_per_update'])"
