"""Fuzz the product path against the oracle on random configurations (GPU box).
usage: gpu_fuzz.py <seed> <cases> — prints every configuration whose result differs or that errors."""
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
rs = np.random.RandomState(seed)
bad = 0
t0 = time.time()
for k in range(ncase):
    nUE = int(rs.choice([1, 5, 64, 100, 300, 1000, 2000, 3500, 5000, 8000, 12000]))
    kw = dict(nPreamble=int(rs.randint(1, 255)), backoff=int(rs.randint(1, 80)), nGrantUL=int(rs.choice([1, 2, 3, 4, 6, 12, 20, 54, 100])),
              maxRarWindow=int(rs.randint(1, 12)), maxMsg2TxCount=int(rs.choice([0, 1, 2, 3, 9, 20])), accessTime=int(rs.randint(1, 20)),
              uniform=int(rs.rand() < 0.2))
    if rs.rand() < 0.5:
        kw["nPreamble"] = int(rs.choice([1, 2, 3, 4, 5, 54, 64]))
    if kw["uniform"]:
        nUE = min(nUE, 3500)
    if rs.rand() < 0.3:
        kw["max_steps"] = int(rs.randint(1, 4000))
    v, r, s = int(rs.randint(0, 2)), int(rs.randint(0, 2)), int(rs.randint(0, 1 << 31))
    G = int(rs.choice([0, 0, 1, 2, 3, 5, 8, 64]))
    legacy = int(rs.rand() < 0.15)
    eng.set("cluster", G); eng.set("legacy", legacy)
    desc = (v, nUE, kw, r, s, "G", G, "legacy", legacy)
    try:
        (res,), (logs,) = eng.run_trials([pkg.make_cfg(nUE, variant=v, rng_mode=r, seed=s, **kw)], want_logs=True)
    except Exception as e:
        bad += 1
        print("case", k, desc, "EXC", e, flush=True)
        continue
    ores, oues = ob.run_trial(ob.make_cfg(nUE, variant=v, **kw), ob.Rng(r, s))
    try:
        tgp.assert_same(pkg, res, logs, ores, oues, k)
    except AssertionError as e:
        bad += 1
        print("case", k, desc, "MISMATCH", str(e)[:400], flush=True)
    if k % 50 == 49:
        print(f"... {k + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("done", ncase, "cases", bad, "bad")
