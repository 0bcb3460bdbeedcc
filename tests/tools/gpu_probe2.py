"""Development probe: cluster kernel parity + timing for several cluster sizes G. Not a test."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
from oracle import binding as ob

m = g.load_package()
eng = m.Engine(0)
sizes = [int(x) for x in sys.argv[1].split(',')]
Gs = [int(x) for x in sys.argv[2].split(',')]
variants = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(',')]
check = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
ok_all = True
ocache = {}
for variant in variants:
    for n in sizes:
        for G in Gs:
            eng.set("cluster", G)
            cfg = m.make_cfg(n, variant=variant, rng_mode=1, seed=0)
            (res,), (logs,) = eng.run_trials([cfg], want_logs=True)
            tm = eng.timing()
            line = f"var={variant} nUE={n} G={G} status={res.status} succ={res.nSuccessUE} exit={res.time_exit} launches={tm.launches} wgs={tm.workgroups} kernel={tm.kernel_ms:.1f}ms us/step={1e3*tm.kernel_ms/max(1,res.steps):.2f} upd/s={n*res.steps/(tm.kernel_ms*1e-3):.3e}"
            if check:
                if (variant, n) not in ocache:
                    ocache[(variant, n)] = ob.run_trial(ob.make_cfg(n, variant=variant), ob.Rng(1, 0))
                ores, oues = ocache[(variant, n)]
                a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
                b = np.frombuffer(oues, dtype=np.int32).reshape(-1, 16)
                keys = ("time_exit", "nSuccessUE", "preambleTxCount", "failCounts", "collisionPreambles", "totalPreambleTxop",
                        "activeCheck", "continueFaliedUEs", "finalSuccessUEs", "sumTimer", "draws", "steps")
                bad = {k: (getattr(res, k), getattr(ores, k)) for k in keys if getattr(res, k) != getattr(ores, k)}
                nd = int((a != b).any(axis=1).sum())
                ok = not bad and nd == 0
                ok_all &= ok
                line += "  PARITY " + ("OK" if ok else f"FAIL {bad} ue_diff={nd}")
                if nd:
                    rows = np.where((a != b).any(axis=1))[0][:3]
                    for r in rows:
                        line += f"\n   row {r}\n    gpu {a[r]}\n    ora {b[r]}"
            print(line, flush=True)
print("ALL PARITY OK" if ok_all else "PARITY FAILURES")
