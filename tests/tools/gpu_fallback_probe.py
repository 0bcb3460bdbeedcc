"""Development probe: which plausible parameter settings at nUE=100k leave the cluster kernel for the fallback kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for v in (0, 1):
    for nP in (4, 8, 16, 30, 54, 64):
        for gr in (12, 54):
            cfg = m.make_cfg(100000, variant=v, rng_mode=1, seed=0, nPreamble=nP, nGrantUL=gr)
            (r,), _ = eng.run_trials([cfg]); tm = eng.timing()
            print(f"var={v} nP={nP} grants={gr} succ={r.nSuccessUE} launches={tm.launches} kernel={tm.kernel_ms:.0f}ms", flush=True)
