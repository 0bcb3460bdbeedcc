"""Development probe: what `prach_sim -t TIMES` does in the reference's own rand() stream — TIMES seeds in flight, the ten sweep points of a seed chained
through its stream (one engine call per sweep point) — with the engine's own clocks per call. Not a test.   usage: gpu_probe_glibc_chain.py [times] [variant]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for kv in filter(None, os.environ.get("PRACH_ENG_OPTS", "").split(",")):
    eng.set(kv.split("=")[0], int(kv.split("=")[1]))
times = int(sys.argv[1]) if len(sys.argv) > 1 else 100
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
off = [0] * times
tk = tw = 0.0
t00 = time.time()
for n in range(10000, 100001, 10000):
    cfgs = [m.make_cfg(n, variant=variant, rng_mode=m.RNG_GLIBC, seed=s, stream_offset=off[s]) for s in range(times)]
    t0 = time.time()
    res, _ = eng.run_trials(cfgs)
    wall = time.time() - t0
    tm = eng.timing()
    for s, r in enumerate(res):
        off[s] += r.draws
    tk += tm.kernel_ms; tw += wall * 1e3
    print(f"nUE={n} G={tm.cluster_size} rec={tm.rec_mode} launches={tm.launches} kernel={tm.kernel_ms:.1f} upload={tm.upload_ms:.1f} total={tm.total_ms:.1f} wall={wall*1e3:.1f} ms fallback={tm.fallback_trials}", flush=True)
print(f"times={times} variant={variant}: kernel {tk:.0f} ms, wall {tw:.0f} ms, script {1e3*(time.time()-t00):.0f} ms")
