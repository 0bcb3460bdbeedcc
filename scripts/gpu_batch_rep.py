"""Development probe: best-of-N kernel time of the batched sweep (BASELINE config 3 shape) and of 500 Beta.c trials. Not a test."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for times, variant in ((100, 1), (50, 0)):
    cfgs = [m.make_cfg(n, variant=variant, rng_mode=m.RNG_PHILOX, seed=s) for s in range(times) for n in range(10000, 100001, 10000)]
    best = 1e30
    for _ in range(reps):
        res, _ = eng.run_trials(cfgs)
        best = min(best, eng.timing().kernel_ms)
    upd = sum(c.nUE * r.steps for c, r in zip(cfgs, res))
    print(f"trials={len(cfgs)} variant={variant} best kernel={best:.1f}ms upd/s={upd/(best*1e-3):.4e}", flush=True)
