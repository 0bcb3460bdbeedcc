"""Development probe: the bench's single trial (BASELINE config 2) under engine options. Not a test.
usage: gpu_single.py "cluster=32,lds_records=1;cluster=16;lds_records=0" [variant] [nUE] [check|-] [glibc]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nUE = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
check = len(sys.argv) > 4 and sys.argv[4] == "check"
rng = m.RNG_GLIBC if len(sys.argv) > 5 and sys.argv[5] == "glibc" else m.RNG_PHILOX
ref = None
if check:
    from oracle import binding as ob
    ref, _ = ob.run_trial(ob.make_cfg(nUE, variant=variant), ob.Rng(ob.RNG_GLIBC if rng == m.RNG_GLIBC else ob.RNG_PHILOX, 0), want_ues=False)
DEFAULTS = dict(cluster=0, lds_records=1, pipeline=1, dense=0, fast=1)
for spec in sys.argv[1].split(";"):
    opts = dict(DEFAULTS)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        opts[k] = int(v)
    for k, v in opts.items():
        eng.set(k, v)
    cfg = m.make_cfg(nUE, variant=variant, rng_mode=rng, seed=0)
    best = 1e9
    for rep in range(3):
        (r,), _ = eng.run_trials([cfg])
        tm = eng.timing()
        best = min(best, tm.kernel_ms)
    ok = ""
    if ref is not None:
        ok = " parity=" + str((r.nSuccessUE, r.time_exit, r.collisionPreambles, r.totalPreambleTxop, r.sumTimer, r.draws) ==
                              (ref.nSuccessUE, ref.time_exit, ref.collisionPreambles, ref.totalPreambleTxop, ref.sumTimer, ref.draws))
    print(f"{spec or 'default':40s} G={tm.cluster_size} rec={tm.rec_mode} status={r.status} succ={r.nSuccessUE} kernel={best:.2f}ms "
          f"us/subframe={1e3*best/r.steps:.3f} upd/s={nUE*r.steps/(best*1e-3):.3e} fallback={tm.fallback_trials}{ok}", flush=True)
