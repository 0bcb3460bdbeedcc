"""Debug helper: run the random parity cases one by one and report status / first mismatch."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
from oracle import binding as ob
pkg = g.load_package()
import importlib.util
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
eng = pkg.Engine(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for legacy in (0, 1):
    eng.set("legacy", legacy)
    eng.set("cluster", G)
    for k, (v, n, kw, r, s) in enumerate(tgp._random_cases(96, 20240 + G)):
        cfg = pkg.make_cfg(n, variant=v, rng_mode=r, seed=s, **kw)
        try:
            (res,), (logs,) = eng.run_trials([cfg], want_logs=True)
        except Exception as e:
            print("legacy", legacy, "case", k, (v, n, kw, r, s), "EXC", e, flush=True)
            continue
        ores, oues = ob.run_trial(ob.make_cfg(n, variant=v, **kw), ob.Rng(r, s))
        try:
            tgp.assert_same(pkg, res, logs, ores, oues, k)
        except AssertionError as e:
            print("legacy", legacy, "case", k, (v, n, kw, r, s), "MISMATCH", str(e)[:300], flush=True)
print("done")
