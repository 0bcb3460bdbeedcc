"""Development probe: batch_kernel<16, true> — 100 seeds of one sweep point in the reference's own rand() stream (what `prach_sim -t 100` issues per point).
usage: gpu_glibc_batch.py [nUE=100000] [seeds=100]   prints kernel ms for Beta.c and WithNOMA and a digest of the results."""
import sys, os, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for kv in filter(None, os.environ.get("PRACH_ENG_OPTS", "").split(",")):
    eng.set(kv.split("=")[0], int(kv.split("=")[1]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for variant in (0, 1):
    cfgs = [m.make_cfg(n, variant=variant, rng_mode=m.RNG_GLIBC, seed=s) for s in range(S)]
    eng.run_trials(cfgs)
    res, _ = eng.run_trials(cfgs)
    tm = eng.timing()
    d = 0
    for r in res:
        d = zlib.crc32(bytes(r), d)
    print(f"variant={variant} trials={S} nUE={n} rec_mode={tm.rec_mode} kernel={tm.kernel_ms:.1f}ms launches={tm.launches} fallback={tm.fallback_trials} digest={d:08x} bad={sum(r.status != 0 for r in res)}")
