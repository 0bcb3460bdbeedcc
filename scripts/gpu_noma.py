"""Development probe: NOMA kernel timing for cluster sizes. Not a test."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for G in [int(x) for x in sys.argv[1].split(',')]:
    eng.set("cluster", G)
    cfg = m.make_cfg(100000, variant=m.VARIANT_NOMA_C, rng_mode=m.RNG_PHILOX, seed=0)
    (r,), _ = eng.run_trials([cfg])
    tm = eng.timing()
    print(f"G={G} status={r.status} succ={r.nSuccessUE} steps={r.steps} kernel={tm.kernel_ms:.1f}ms upd/s={1e5*r.steps/(tm.kernel_ms*1e-3):.3e} upload={tm.upload_ms:.1f}ms")
