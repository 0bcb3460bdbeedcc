"""Fuzz the NOMA.c kernels against the oracle on random configurations (GPU box).  usage: gpu_fuzz_noma.py <seed> <cases> [glibc]
(glibc: the reference's own rand() stream — noma_glibc_trial_kernel, one launch per trial, activeUE on the device; the nonsector flag on a third of the cases)"""
import sys
import time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
from oracle import binding as ob
pkg = g.load_package()
eng = pkg.Engine(0)
seed, ncase = int(sys.argv[1]), int(sys.argv[2])
GLIBC = "glibc" in sys.argv[3:]
nfall = 0
rs = np.random.RandomState(seed)
bad = 0
t0 = time.time()
for k in range(ncase):
    nUE = int(rs.choice([1, 7, 64, 65, 500, 2000, 6000, 15000, 40000]))
    kw = dict(nPreamble=int(rs.choice([1, 2, 3, 8, 54, 64, 33])), backoff=int(rs.randint(1, 60)), nGrantUL=int(rs.choice([1, 2, 3, 5, 12, 40])),
              maxRarWindow=int(rs.randint(1, 10)), maxMsg2TxCount=int(rs.choice([0, 1, 3, 10, 25])), accessTime=int(rs.choice([1, 2, 5, 5, 5, 8, 10, 60])))
    if rs.rand() < 0.3:
        kw["max_steps"] = int(rs.randint(1, 6000))
    if rs.rand() < 0.3:
        kw["cellRadius"] = float(rs.choice([50.0, 250.0, 2000.0]))
    s = int(rs.randint(0, 1 << 31))
    G = int(rs.choice([0, 0, 1, 2, 3, 7, 32]))
    eng.set("cluster", G)
    desc = (nUE, kw, s, "G", G)
    okw = dict(kw); okw["maxMsg1ReTx"] = okw.pop("maxMsg2TxCount")
    nonsector = int(GLIBC and rs.rand() < 0.33)
    try:
        cfg = pkg.make_cfg(nUE, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_GLIBC if GLIBC else pkg.RNG_PHILOX, seed=s, flags=pkg.FLAG_NOMA_NONSECTOR if nonsector else 0, **kw)
        (res,), (logs,) = eng.run_trials([cfg], want_logs=True)
        nfall += eng.timing().fallback_trials
    except Exception as e:
        bad += 1
        print("case", k, desc, "EXC", e, flush=True)
        continue
    ocfg = ob.make_noma_cfg(nUE, nonsector=nonsector, **okw)
    ores, oues = ob.noma_run_trial(ocfg, ob.Rng(ob.RNG_GLIBC if GLIBC else ob.RNG_PHILOX, s))
    a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
    b = np.frombuffer(oues, dtype=np.dtype([("i", np.int32, 16), ("g", np.float64)]))["i"]
    ra = (res.status, res.nSuccessUE, res.sumTimer, res.preambleTxCount, res.failCounts, res.activeCheck, res.draws, res.time_exit)
    rb = (0, ores.nSuccessUE, ores.delay, ores.nTxP, ores.raFailedUEs, ores.activeCheck, ores.draws, ores.time_exit)
    if ra != rb or not (a == b).all():
        bad += 1
        d = np.where((a != b).any(axis=1))[0]
        print("case", k, desc, "MISMATCH", ra, rb, d[:5], flush=True)
    if k % 50 == 49:
        print(f"... {k + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("done", ncase, "cases", bad, "bad" + (f"; {nfall} trials rerun with host-side activation (a value inside the device libm's error band)" if GLIBC else ""))
