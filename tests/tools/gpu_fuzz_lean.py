"""Fuzz prach::lcluster_kernel (Philox clusters with LDS-resident UE state, nPreamble <= 64) against the oracle on random configurations
(GPU box).  usage: gpu_fuzz_lean.py <seed> <cases> [big] — prints every configuration whose result differs or that errors, and how many
cases really ran on the lean kernel (prach_timing.rec_mode == 3) / fell back."""
import sys
import time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
from oracle import binding as ob
import importlib.util
pkg = g.load_package()
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
eng = pkg.Engine(0)
seed, ncase = int(sys.argv[1]), int(sys.argv[2])
big = len(sys.argv) > 3 and sys.argv[3] == "big"
rs = np.random.RandomState(seed)
bad = lean = fb = tkr = 0
t0 = time.time()
for k in range(ncase):
    if big:
        nUE = int(rs.choice([20000, 40000, 70000, 100000]))
        kw = dict(nPreamble=int(rs.choice([8, 30, 54, 64])), backoff=int(rs.choice([5, 10, 20, 40])), nGrantUL=int(rs.choice([2, 6, 12, 30, 54])),
                  maxRarWindow=int(rs.choice([2, 4, 6, 9])), maxMsg2TxCount=int(rs.choice([1, 4, 9, 20])), accessTime=int(rs.choice([5, 5, 6, 10])))
        G = int(rs.choice([0, 16, 32, 48, 64]))
    else:
        nUE = int(rs.choice([65, 130, 300, 1000, 2000, 3500, 5000, 8000, 12000, 20000]))
        kw = dict(nPreamble=int(rs.choice([1, 2, 3, 4, 5, 7, 16, 33, 54, 64])), backoff=int(rs.randint(1, 80)),
                  nGrantUL=int(rs.choice([1, 2, 3, 4, 6, 12, 20, 54, 100])), maxRarWindow=int(rs.randint(1, 12)),
                  maxMsg2TxCount=int(rs.choice([0, 1, 2, 3, 9, 20])), accessTime=int(rs.randint(1, 20)), uniform=int(rs.rand() < 0.2))
        if kw["uniform"]:
            nUE = min(nUE, 3500)
        if rs.rand() < 0.3:
            kw["max_steps"] = int(rs.randint(1, 4000))
        G = int(rs.choice([2, 3, 5, 8, 16, 32, 64]))
    v, s = int(rs.randint(0, 2)), int(rs.randint(0, 1 << 31))
    eng.set("cluster", G)
    desc = (v, nUE, kw, s, "G", G)
    try:
        (res,), (logs,) = eng.run_trials([pkg.make_cfg(nUE, variant=v, rng_mode=1, seed=s, **kw)], want_logs=True)
    except Exception as e:
        bad += 1
        print("case", k, desc, "EXC", e, flush=True)
        continue
    tm = eng.timing()
    lean += tm.rec_mode == 3 and tm.fallback_trials == 0
    fb += tm.fallback_trials
    tkr += tm.trial_kernel_reruns
    ores, oues = ob.run_trial(ob.make_cfg(nUE, variant=v, **kw), ob.Rng(1, s))
    try:
        tgp.assert_same(pkg, res, logs, ores, oues, k)
    except AssertionError as e:
        bad += 1
        print("case", k, desc, "MISMATCH", str(e)[:400], flush=True)
    if k % 50 == 49:
        print(f"... {k + 1} cases, {bad} bad, {lean} on the lean kernel, {fb} fallbacks, {time.time() - t0:.0f} s", flush=True)
print("done", ncase, "cases", bad, "bad;", lean, "ran on the lean kernel,", fb, "fell back (to the batch kernel),", tkr, f"of them on to trial_kernel = {100.0 * tkr / ncase:.1f} % of the cases")
