set -e
scripts/gpu_pmc.sh r03base c3 -- scripts/gpu_batch.py 100 1 0
PMC_GROUPS="fetch write sq vmem" scripts/gpu_pmc.sh r03base noma1 -- scripts/gpu_noma_batch.py single
PMC_GROUPS="fetch write sq vmem" scripts/gpu_pmc.sh r03base nomaB -- scripts/gpu_noma_batch.py batch 10
python3 scripts/gpu_batch.py 100 1 0 > gpurun_out/prof_r03base/c3_plain.log 2>&1
python3 scripts/gpu_noma_batch.py single > gpurun_out/prof_r03base/noma1_plain.log 2>&1
python3 scripts/gpu_noma_batch.py batch 10 > gpurun_out/prof_r03base/nomaB_plain.log 2>&1
cat gpurun_out/prof_r03base/*_plain.log
