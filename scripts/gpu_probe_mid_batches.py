"""Development probe: a few dozen to a few hundred concurrent trials (the sweep x `--times` 1..25) — automatic cluster size vs one workgroup per trial. Not a test."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for times in (1, 3, 6, 10, 15, 25):
    for variant in (0, 1):
        cfgs = [m.make_cfg(n, variant=variant, rng_mode=m.RNG_PHILOX, seed=s) for s in range(times) for n in range(10000, 100001, 10000)]
        for G in (0, 1, 2, 4):
            eng.set("cluster", G)
            best = 1e9
            for rep in range(2):
                res, _ = eng.run_trials(cfgs); tm = eng.timing(); best = min(best, tm.kernel_ms)
            upd = sum(c.nUE * r.steps for c, r in zip(cfgs, res))
            print(f"trials={len(cfgs)} variant={variant} cluster={G} -> G={tm.cluster_size} rec={tm.rec_mode} kernel={best:.1f}ms upd/s={upd/(best*1e-3):.3e} fallback={tm.fallback_trials}", flush=True)
