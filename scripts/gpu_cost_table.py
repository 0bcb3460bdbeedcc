"""Measures what a trial of each sweep point COSTS inside a full batched launch (the unit the multi-GPU dealing balances): for every program (Beta.c as
committed, RandomAccessWithNOMA defaults) and nUE point, 1024 trials of that one size in one call (two per CU, two rounds) -> kernel ms per trial.
Writes profiles/r04_cost_table.json; the numbers are compiled into prach_trial_cost (csrc/prach_host.c).   usage (GPU box): python3 scripts/gpu_cost_table.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
K = 1024
out = {"trials_per_call": K, "unit": "kernel microseconds per trial inside a launch of 1024 trials of the same size (batch_kernel, 512-thread shape)", "programs": {}}
for name, variant in (("beta_c", m.VARIANT_BETA_C), ("withnoma_c", m.VARIANT_WITHNOMA_C)):
    row = {}
    for n in range(10000, 100001, 10000):
        cfgs = [m.make_cfg(n, variant=variant, rng_mode=m.RNG_PHILOX, seed=s) for s in range(K)]
        res, _ = eng.run_trials(cfgs)
        tm = eng.timing()
        assert all(r.status == 0 for r in res) and tm.fallback_trials == 0, (name, n, tm.fallback_trials)
        row[str(n)] = round(1e3 * tm.kernel_ms / K, 2)
        print(name, n, f"{tm.kernel_ms:.1f} ms for {K} trials -> {row[str(n)]} us per trial", flush=True)
    out["programs"][name] = row
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_cost_table.json"), "w"), indent=1)
print(json.dumps(out))
