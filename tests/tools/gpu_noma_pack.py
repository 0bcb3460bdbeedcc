"""Development probe: NOMA kernel, XCD-packed launch on / off. Not a test."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for G in (0, 8, 32):
    for pack in (1, 0):
        eng.set("cluster", G); eng.set("xcd_pack", pack)
        cfg = m.make_cfg(100000, variant=m.VARIANT_NOMA_C, rng_mode=m.RNG_PHILOX, seed=0)
        best = 1e9
        for rep in range(3):
            (r,), _ = eng.run_trials([cfg]); tm = eng.timing(); best = min(best, tm.kernel_ms)
        print(f"G={tm.cluster_size} xcd_pack={pack} packed={tm.xcd_packed} status={r.status} succ={r.nSuccessUE} steps={r.steps} kernel={best:.2f}ms upd/s={1e5*r.steps/(best*1e-3):.3e}")
