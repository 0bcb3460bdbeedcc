"""Development probe: the NOMA.c program on the GPU. Not a test.
  gpu_noma_batch.py single        one nUE = 100 000 trial (BASELINE configs[3])
  gpu_noma_batch.py batch [S]     NOMA.c's own experiment in ONE call: S seeds (default 10) x the ten sweep points
Prints kernel time, updates/s and the kernel's own record traffic (a record is loaded and stored once per 5 ms access slot)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for kv in filter(None, os.environ.get("PRACH_ENG_OPTS", "").split(",")):
    eng.set(kv.split("=")[0], int(kv.split("=")[1]))
mode = sys.argv[1] if len(sys.argv) > 1 else "single"
if mode == "single":
    cfgs = [m.make_cfg(100000, variant=m.VARIANT_NOMA_C, rng_mode=m.RNG_PHILOX, seed=0)]
else:
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    cfgs = [m.make_cfg(n, variant=m.VARIANT_NOMA_C, rng_mode=m.RNG_PHILOX, seed=s) for s in range(S) for n in range(10000, 100001, 10000)]
eng.run_trials(cfgs)  # warm-up (arena, code object)
t0 = time.time()
res, _ = eng.run_trials(cfgs)
wall = time.time() - t0
tm = eng.timing()
upd = sum(c.nUE * r.steps for c, r in zip(cfgs, res))
slots = sum(c.nUE * ((r.steps + 4) // 5) for c, r in zip(cfgs, res))
print(f"mode={mode} trials={len(cfgs)} G={tm.cluster_size} launches={tm.launches} wgs={tm.workgroups} kernel={tm.kernel_ms:.2f}ms upload={tm.upload_ms:.2f}ms wall={wall*1e3:.1f}ms "
      f"updates={upd:.4e} kernel_upd/s={upd/(tm.kernel_ms*1e-3):.3e} wall_upd/s={upd/wall:.3e} ue_slots={slots:.4e} bad={sum(r.status != 0 for r in res)} fallback={tm.fallback_trials} host_recomputed_ues={tm.noma_host_ues}")
