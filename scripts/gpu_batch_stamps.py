"""Development probe (GPU box, DIAGNOSTIC build only): where a batch_kernel subframe goes UNDER LOAD — the per-phase cycle stamps of every
trial of a batched launch (thread 0 of each workgroup, `make DIAG=1 lib`), averaged per nUE point.
usage: PRACH_LIB=.../libprach_hip_diag.so python3 scripts/gpu_batch_stamps.py TIMES VARIANT WAVES
  TIMES x the 10-point sweep, VARIANT 0 Beta.c (the grid of configs[4]) / 1 WithNOMA (configs[2]), WAVES 8 | 16 | 0 (the engine's choice).
The engine prints one "[prach fine stamps/step]" line per trial on stderr; this script captures its own stderr and tabulates it."""
import os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PRACH_PRINT_STAMPS"] = "1"
import __graft_entry__ as g
m = g.load_package()
times, variant, waves = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
eng = m.Engine(0)
eng.set("batch_waves", waves)
cfgs = [m.make_cfg(n, variant=variant, rng_mode=m.RNG_PHILOX, seed=s) for s in range(times) for n in range(10000, 100001, 10000)]
tmp = tempfile.TemporaryFile(mode="w+")
sys.stderr.flush()
saved = os.dup(2)
os.dup2(tmp.fileno(), 2)
try:
    res, _ = eng.run_trials(cfgs)
finally:
    sys.stderr.flush()
    os.dup2(saved, 2)
tm = eng.timing()
tmp.seek(0)
rows = {}
names = None
for line in tmp:
    if not line.startswith("[prach fine stamps/step]"):
        continue
    kv = dict(re.findall(r"([\w:+|\-()]+)=([0-9.]+)", line))
    n = int(kv.pop("nUE")); steps = int(float(kv.pop("steps")))
    if names is None:
        names = list(kv)
    rows.setdefault(n, []).append((steps, [float(kv[k]) for k in names]))
upd = sum(c.nUE * r.steps for c, r in zip(cfgs, res))
print(f"trials={len(cfgs)} variant={variant} waves={waves} kernel={tm.kernel_ms:.1f} ms updates={upd:.3e} upd/s={upd / (tm.kernel_ms * 1e-3):.3e} "
      f"visits={tm.group_visits:.3e} events={tm.event_ues:.3e} bad={sum(r.status != 0 for r in res)} fallback={tm.fallback_trials}")
if not names:
    sys.exit("no stamps: PRACH_LIB must point at the diagnostic build (make -C 5g-nr-randomaccess_amd/csrc DIAG=1 lib)")
print("cycles per subframe (mean over the trials of a point; thread 0 of the workgroup)")
print("nUE     steps " + " ".join(f"{k[:12]:>12s}" for k in names) + "        total")
for n in sorted(rows):
    tr = rows[n]
    k = len(tr)
    mean = [sum(t[1][j] for t in tr) / k for j in range(len(names))]
    print(f"{n:6d} {sum(t[0] for t in tr) / k:6.0f} " + " ".join(f"{v:12.0f}" for v in mean) + f" {sum(mean):12.0f}")
