import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for G in (0, 1, 2, 4, 8):
    eng.set("cluster", G)
    for v in (0, 1):
        cfg = m.make_cfg(100000, variant=v, uniform=1, rng_mode=1, seed=0)
        (r,), _ = eng.run_trials([cfg]); tm = eng.timing()
        print(f"G={G} var={v} succ={r.nSuccessUE} steps={r.steps} kernel={tm.kernel_ms:.1f}ms us/step={1e3*tm.kernel_ms/max(1,r.steps):.2f} wgs={tm.workgroups}", flush=True)
