"""Development probe: one-workgroup-per-trial launches (prach::batch_kernel) against the oracle, field by field. Not a test.
usage: gpu_batch_debug.py nUE variant [key=value ...]   (cfg fields, e.g. uniform=1 nGrantUL=12 max_steps=200)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
from oracle import binding as ob
m = g.load_package()
nUE, variant = int(sys.argv[1]), int(sys.argv[2])
kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[3:])}
seed = kw.pop("seed", 0)
eng = m.Engine(0)
for kv in filter(None, os.environ.get("PRACH_ENG_OPTS", "").split(",")):  # e.g. PRACH_ENG_OPTS=batch_waves=16,batch=0
    eng.set(kv.split("=")[0], int(kv.split("=")[1]))
eng.set("cluster", 1)
cfg = m.make_cfg(nUE, variant=variant, rng_mode=m.RNG_PHILOX, seed=seed, **kw)
(res,), (logs,) = eng.run_trials([cfg], want_logs=True)
tm = eng.timing()
ores, oues = ob.run_trial(ob.make_cfg(nUE, variant=variant, **kw), ob.Rng(ob.RNG_PHILOX, seed))
print("rec_mode", tm.rec_mode, "G", tm.cluster_size, "status", res.status, "fallback", tm.fallback_trials, "kernel_ms", tm.kernel_ms)
for f in ("time_exit", "nSuccessUE", "collisionPreambles", "totalPreambleTxop", "preambleTxCount", "sumTimer", "draws", "steps", "activeCheck", "continueFaliedUEs", "failCounts"):
    a, b = getattr(res, f), getattr(ores, f)
    print(f"{f:20s} gpu={a} oracle={b} {'' if a == b else '  <-- DIFF'}")
a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
b = np.frombuffer(oues, dtype=np.int32).reshape(-1, 16)
diff = np.where((a != b).any(axis=1))[0]
print("UEs that differ:", diff.size, "of", nUE)
for i in diff[:6]:
    print(" UE", i)
    for k, f in enumerate(m.UE_FIELDS):
        if a[i, k] != b[i, k]:
            print(f"    {f:18s} gpu={a[i, k]} oracle={b[i, k]}")
