"""Development probe: automatic vs forced cluster sizes for a few concurrent Philox trials. Not a test."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for ntr in (2, 4, 8, 10, 16):
    cfgs = [m.make_cfg(100000, variant=0, rng_mode=m.RNG_PHILOX, seed=s) for s in range(ntr)]
    for G in (0, 8, 16, 32):
        eng.set("cluster", G)
        best = 1e9
        for rep in range(2):
            res, _ = eng.run_trials(cfgs); tm = eng.timing(); best = min(best, tm.kernel_ms)
        upd = sum(c.nUE * r.steps for c, r in zip(cfgs, res))
        print(f"trials={ntr} cluster={G} -> G={tm.cluster_size} rec={tm.rec_mode} packed={tm.xcd_packed} kernel={best:.1f}ms upd/s={upd/(best*1e-3):.3e} fallback={tm.fallback_trials}", flush=True)
