set -e
mkdir -p gpurun_out/r03
python3 bench.py --steps 5 --warmup 1 --cpu-full > gpurun_out/r03/bench_full.json 2> gpurun_out/r03/bench_full.err || { tail -20 gpurun_out/r03/bench_full.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/r03/bench_full.json'))
print('value',d['value'],'ms',d['ms_per_step'],'us/subframe',d['roofline']['us_per_subframe'])
print('cpu100k',d.get('cpu_baseline_at_metric_size'))
e=d['extras']
for k in e: print(k, json.dumps(e[k])[:300])
"
