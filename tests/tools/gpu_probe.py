"""Development probe (GPU box): parity vs oracle at a few sizes + kernel timing. Not a test."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
from oracle import binding as ob

m = g.load_package()
eng = m.Engine(0)
sizes = [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else "3000,10000,30000,100000".split(','))]
check = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
ok_all = True
for variant in (0, 1):
    for rng in (0, 1):
        for n in sizes:
            cfg = m.make_cfg(n, variant=variant, rng_mode=rng, seed=0)
            t0 = time.time()
            (res,), (logs,) = eng.run_trials([cfg], want_logs=True)
            wall = time.time() - t0
            tm = eng.timing()
            line = f"var={variant} rng={rng} nUE={n} status={res.status} succ={res.nSuccessUE} exit={res.time_exit} draws={res.draws} kernel={tm.kernel_ms:.1f}ms upload={tm.upload_ms:.1f}ms wall={wall*1e3:.0f}ms upd/s={n*res.steps/ (tm.kernel_ms*1e-3):.3e}"
            if check:
                ocfg = ob.make_cfg(n, variant=variant)
                ores, oues = ob.run_trial(ocfg, ob.Rng(rng, 0))
                a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
                b = np.frombuffer(oues, dtype=np.int32).reshape(-1, 16)
                keys = ("time_exit", "nSuccessUE", "preambleTxCount", "failCounts", "collisionPreambles", "totalPreambleTxop",
                        "activeCheck", "continueFaliedUEs", "finalSuccessUEs", "sumTimer", "draws", "steps")
                bad = {k: (getattr(res, k), getattr(ores, k)) for k in keys if getattr(res, k) != getattr(ores, k)}
                nd = int((a != b).any(axis=1).sum())
                ok = not bad and nd == 0 and res.totalDelay == ores.totalDelay
                ok_all &= ok
                line += "  PARITY " + ("OK" if ok else f"FAIL {bad} ue_diff={nd}")
                if nd:
                    rows = np.where((a != b).any(axis=1))[0][:4]
                    for r in rows:
                        line += f"\n   row {r}\n    gpu {a[r]}\n    ora {b[r]}"
            print(line, flush=True)
print("ALL PARITY OK" if ok_all else "PARITY FAILURES")
