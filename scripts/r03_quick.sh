set -e
mkdir -p gpurun_out/r03
python3 -c "import __graft_entry__ as g; g.smoke()"
python3 bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err || { tail -20 gpurun_out/r03/bench_default.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/r03/bench_default.json'))
print('value',d['value'],'frac',d['roofline']['frac'],'us',d['roofline']['us_per_subframe'])
print('glibc',d['extras']['glibc_mode_reference_point'])
print('cpu',d['cpu_baseline']['value'], d['cpu_baseline']['at_nUE_100000']['value'], d['cpu_baseline']['at_nUE_100000'].get('measured_in_this_run'))
print('noma', d['extras']['noma_c_experiment_batched']['roofline']['frac'], d['extras']['noma_c_single_trial']['roofline']['frac'])
"
