"""Development probe: NOMA.c in its own rand() stream — ten seeds of nUE = 100 000 in ONE launch (noma_glibc_trial_kernel). Not a test."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for kv in filter(None, os.environ.get("PRACH_ENG_OPTS", "").split(",")):
    eng.set(kv.split("=")[0], int(kv.split("=")[1]))
cfgs = [m.make_cfg(100000, variant=m.VARIANT_NOMA_C, rng_mode=m.RNG_GLIBC, seed=s) for s in range(10)]
t0 = time.time(); res, _ = eng.run_trials(cfgs); wall = time.time() - t0
tm = eng.timing()
upd = sum(c.nUE * r.steps for c, r in zip(cfgs, res))
print(f"trials={len(cfgs)} kernel={tm.kernel_ms:.1f}ms wall={wall*1e3:.1f}ms launches={tm.launches} rerun_with_host_activation={tm.fallback_trials} updates={upd:.3e} kernel_upd/s={upd/(tm.kernel_ms*1e-3):.3e} bad={sum(r.status != 0 for r in res)}")
