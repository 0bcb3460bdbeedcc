"""Development probe: a growing sequence of calls on one engine with PRACH_VERBOSE=1 — which HIP call of the reserved-range arena refuses what."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PRACH_VERBOSE"] = "1"
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for ntr in (3, 50, 300, 600):
    cfgs = [m.make_cfg(20000, variant=s % 2, rng_mode=m.RNG_PHILOX, seed=s, max_steps=200) for s in range(ntr)]
    res, _ = eng.run_trials(cfgs)
    print(ntr, "trials ok", sum(r.status for r in res), flush=True)
