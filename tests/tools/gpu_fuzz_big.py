import sys, time, os
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
rs = np.random.RandomState(int(sys.argv[1])); bad = 0; t0 = time.time()
for k in range(int(sys.argv[2])):
    nUE = int(rs.choice([20000, 40000, 70000]))
    kw = dict(nPreamble=int(rs.choice([8, 30, 54, 64, 120])), backoff=int(rs.choice([5, 10, 20, 40])), nGrantUL=int(rs.choice([2, 6, 12, 30, 54])),
              maxRarWindow=int(rs.choice([2, 4, 6, 9])), maxMsg2TxCount=int(rs.choice([1, 4, 9, 20])), accessTime=int(rs.choice([5, 5, 6, 10])))
    v, s = int(rs.randint(0, 2)), int(rs.randint(0, 1 << 31))
    G = int(rs.choice([0, 8, 32, 1])); eng.set("cluster", G)
    try:
        (res,), (logs,) = eng.run_trials([pkg.make_cfg(nUE, variant=v, rng_mode=1, seed=s, **kw)], want_logs=True)
        ores, oues = ob.run_trial(ob.make_cfg(nUE, variant=v, **kw), ob.Rng(1, s))
        tgp.assert_same(pkg, res, logs, ores, oues, k)
    except Exception as e:
        bad += 1; print("case", k, (v, nUE, kw, s, G), "BAD", str(e)[:300], flush=True)
    if k % 10 == 9: print(f"... {k+1} cases, {bad} bad, {time.time()-t0:.0f} s, launches={eng.timing().launches}", flush=True)
print("done", bad, "bad")
