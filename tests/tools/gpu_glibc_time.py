import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for kv in filter(None, os.environ.get("PRACH_ENG_OPTS", "").split(",")):
    eng.set(kv.split("=")[0], int(kv.split("=")[1]))
for variant, nUE in ((0, 100000), (1, 100000), (0, 40000)):
    for pack in (1, 0):
        eng.set("xcd_pack", pack)
        cfg = m.make_cfg(nUE, variant=variant, rng_mode=m.RNG_GLIBC, seed=0)
        best = 1e9
        for rep in range(3):
            (r,), _ = eng.run_trials([cfg])
            tm = eng.timing()
            best = min(best, tm.kernel_ms)
        print(f"glibc variant={variant} nUE={nUE} xcd_pack={pack} packed={tm.xcd_packed} G={tm.cluster_size} rec={tm.rec_mode} status={r.status} succ={r.nSuccessUE} kernel={best:.2f}ms upd/s={nUE*r.steps/(best*1e-3):.3e}")
# Philox cluster on the general kernel (nPreamble > 64)
for pack in (1, 0):
    eng.set("xcd_pack", pack)
    cfg = m.make_cfg(100000, variant=0, rng_mode=m.RNG_PHILOX, seed=0, nPreamble=128)
    best = 1e9
    for rep in range(3):
        (r,), _ = eng.run_trials([cfg]); tm = eng.timing(); best = min(best, tm.kernel_ms)
    print(f"philox nP=128 xcd_pack={pack} packed={tm.xcd_packed} G={tm.cluster_size} rec={tm.rec_mode} status={r.status} succ={r.nSuccessUE} kernel={best:.2f}ms")
