import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for variant, times in ((1, 100), (0, 50)):
    cfgs = [m.make_cfg(n, variant=variant, rng_mode=m.RNG_PHILOX, seed=s) for s in range(times) for n in range(10000, 100001, 10000)]
    for two in (0, 1, 0, 1):
        eng.set("two_per_cu", two)
        res, _ = eng.run_trials(cfgs)
        tm = eng.timing()
        upd = sum(c.nUE * r.steps for c, r in zip(cfgs, res))
        print(f"variant={variant} trials={len(cfgs)} two_per_cu={two} kernel={tm.kernel_ms:.1f}ms upd/s={upd/(tm.kernel_ms*1e-3):.3e} fallback={tm.fallback_trials} bad={sum(r.status!=0 for r in res)}", flush=True)
