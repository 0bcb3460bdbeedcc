"""Development probe: batched sweep throughput (BASELINE config 3 shape). Not a test — but it prints a digest of every trial's
result block, so that an experimental build whose output differs (and is therefore not measuring the same work) is seen at once."""
import sys, time, os, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
for kv in filter(None, os.environ.get("PRACH_ENG_OPTS", "").split(",")):  # e.g. PRACH_ENG_OPTS=batch_waves=16,batch=0
    eng.set(kv.split("=")[0], int(kv.split("=")[1]))
times = int(sys.argv[1]) if len(sys.argv) > 1 else 26
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 1
G = int(sys.argv[3]) if len(sys.argv) > 3 else 0
eng.set("cluster", G)
cfgs = [m.make_cfg(n, variant=variant, rng_mode=m.RNG_PHILOX, seed=s) for s in range(times) for n in range(10000, 100001, 10000)]
t0 = time.time()
res, _ = eng.run_trials(cfgs)
wall = time.time() - t0
tm = eng.timing()
upd = sum(c.nUE * r.steps for c, r in zip(cfgs, res))
digest = 0
for r in res:
    digest = zlib.crc32(bytes(r), digest)
print(f"digest={digest:08x} trials={len(cfgs)} variant={variant} G={G} launches={tm.launches} wgs={tm.workgroups} kernel={tm.kernel_ms:.1f}ms wall={wall:.2f}s updates={upd:.3e} kernel_upd/s={upd/(tm.kernel_ms*1e-3):.3e} algoGB/s={32*upd/(tm.kernel_ms*1e-3)/1e9:.0f} bad={sum(r.status!=0 for r in res)}")
# the kernel's OWN bytes.  batch_kernel (rec_mode 4): an event UE's 32-byte record is read from the chunk it sits in and written into the chunk of its next event's
# subframe (whole 64-record chunks are counted), 64 B; a contention window costs a 4-byte join-list entry out and in, 8 B.  The general kernel's 8 + 4 byte form: 8 B per lane and visit, ~40 B per event.
own = tm.event_ues * 64 + tm.group_visits * 8 if tm.rec_mode == 4 else tm.group_visits * 512 + tm.event_ues * 40
print(f"rec_mode={tm.rec_mode} fallback={tm.fallback_trials} own traffic: {tm.event_ues:.3e} event records x 64 B, {tm.group_visits:.3e} {'joins x 8 B' if tm.rec_mode == 4 else 'group visits x 512 B'} -> {own/1e9:.1f} GB = {own/upd:.2f} B/update = {own/(tm.kernel_ms*1e-3)/1e12:.2f} TB/s")
