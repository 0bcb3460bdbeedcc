"""Fuzz prach::batch_kernel (one workgroup per trial, Philox) against the oracle: random configurations inside the kernel's limits,
several trials per call.  usage: gpu_fuzz_batch.py <seed> <calls> [big] [glibc]   (PRACH_LIB=...tinyq.so: the queue's global part)"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
from oracle import binding as ob
import importlib.util
from concurrent.futures import ThreadPoolExecutor
pkg = g.load_package()
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
eng = pkg.Engine(0)
for kv in filter(None, os.environ.get("PRACH_ENG_OPTS", "").split(",")):  # e.g. PRACH_ENG_OPTS=batch_waves=16
    eng.set(kv.split("=")[0], int(kv.split("=")[1]))
eng.set("cluster", 1)
seed, ncalls = int(sys.argv[1]), int(sys.argv[2])
big = len(sys.argv) > 3 and sys.argv[3] == "big"
GLIBC = "glibc" in sys.argv[3:]  # the reference's own rand() stream: batch_kernel<16, true>
RNG = pkg.RNG_GLIBC if GLIBC else pkg.RNG_PHILOX
rs = np.random.RandomState(seed)
bad = ntr = notbatch = tkr = 0
t0 = time.time()
sizes = [1, 5, 63, 64, 65, 100, 300, 1000, 2000, 3500, 5000, 8000, 12000] + ([20000, 30000, 50000, 70000] if big else [])
for k in range(ncalls):
    cfgs, descs = [], []
    for _ in range(int(rs.randint(1, 5 if big else 9))):
        nUE = int(rs.choice(sizes))
        kw = dict(nPreamble=int(rs.randint(1, 65)), backoff=int(rs.randint(1, 80)), nGrantUL=int(rs.choice([1, 2, 3, 4, 6, 12, 20, 54, 100])),
                  maxRarWindow=int(rs.randint(1, 12)), maxMsg2TxCount=int(rs.choice([0, 1, 2, 3, 9, 20])), accessTime=int(rs.randint(1, 20)),
                  uniform=int(rs.rand() < 0.15))
        if rs.rand() < 0.5:
            kw["nPreamble"] = int(rs.choice([1, 2, 3, 4, 5, 54, 64]))
        if rs.rand() < 0.1:
            kw["maxRarWindow"] = int(rs.choice([20, 40, 64]))
        if kw["uniform"]:
            nUE = min(nUE, 3500)
        if rs.rand() < 0.3:
            kw["max_steps"] = int(rs.randint(1, 4000))
        v, s = int(rs.randint(0, 2)), int(rs.randint(0, 1 << 31))
        sect = int(v == 1 and rs.rand() < 0.25)  # the dormant per-sector grant path (WithNOMA only), on the batch kernel too
        cfgs.append(pkg.make_cfg(nUE, variant=v, rng_mode=RNG, seed=s, flags=pkg.FLAG_SECTOR_GRANTS if sect else 0, **kw))
        descs.append((v, nUE, kw, s, sect))
    try:
        res, logs = eng.run_trials(cfgs, want_logs=True)
    except Exception as e:
        bad += 1
        print("call", k, descs, "EXC", e, flush=True)
        continue
    tm = eng.timing()
    notbatch += tm.rec_mode != 4 or tm.fallback_trials != 0
    tkr += tm.trial_kernel_reruns

    def one(j):
        v, nUE, kw, s, sect = descs[j]
        return ob.run_trial(ob.make_cfg(nUE, variant=v, sector_grants=sect, **kw), ob.Rng(ob.RNG_GLIBC if GLIBC else ob.RNG_PHILOX, s))
    with ThreadPoolExecutor(max_workers=8) as ex:
        outs = list(ex.map(one, range(len(cfgs))))
    for j, (ores, oues) in enumerate(outs):
        ntr += 1
        try:
            tgp.assert_same(pkg, res[j], logs[j], ores, oues, k)
        except AssertionError as e:
            bad += 1
            print("call", k, "trial", j, descs[j], "MISMATCH", str(e)[:400], flush=True)
    if k % 20 == 19:
        print(f"... {k + 1} calls, {ntr} trials, {bad} bad, {notbatch} calls not (only) on the batch kernel, {time.time() - t0:.0f} s", flush=True)
print("done", ncalls, "calls", ntr, "trials", bad, "bad", notbatch, "calls not (only) on the batch kernel;", tkr, f"trials rerun on trial_kernel = {100.0 * tkr / max(ntr, 1):.1f} %")
