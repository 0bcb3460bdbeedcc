"""Development probe: the reference-stream (glibc) mode with `--times` seeds in flight — one launch per sweep point (the points of a seed are chained
through its rand() stream), `times` trials of the same nUE each.  Automatic cluster size vs forced ones. Not a test."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
times = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for variant in (0, 1):
    for n in (20000, 60000, 100000):
        cfgs = [m.make_cfg(n, variant=variant, rng_mode=m.RNG_GLIBC, seed=s) for s in range(times)]
        for G in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2".split(","))]:
            eng.set("cluster", G)
            res, _ = eng.run_trials(cfgs); tm = eng.timing()
            upd = sum(c.nUE * r.steps for c, r in zip(cfgs, res))
            print(f"trials={len(cfgs)} nUE={n} variant={variant} cluster={G} -> G={tm.cluster_size} rec={tm.rec_mode} kernel={tm.kernel_ms:.1f}ms wall={tm.total_ms:.1f}ms upd/s={upd/(tm.kernel_ms*1e-3):.3e} "
                  f"fallback={tm.fallback_trials} tk={tm.trial_kernel_reruns} launches={tm.launches}", flush=True)
