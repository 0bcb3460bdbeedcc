"""GPU parity tests proper (run with -m gpu on an MI355X).  Everything goes through the C ABI
(libprach_hip.so); the oracle and the golden fixtures are only the checkers.  Bit-exact bar: every
per-trial counter and every logged field of every UE; totalDelay (float) exact as well."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden, split_stdout_blocks

pytestmark = pytest.mark.gpu

KEYS = ("time_exit", "nSuccessUE", "failedUEs", "preambleTxCount", "failCounts", "collisionPreambles",
        "totalPreambleTxop", "activeCheck", "nAccessUE", "continueFaliedUEs", "finalSuccessUEs", "sumTimer", "draws", "steps")


def assert_same(pkg, res, logs, ores, oues, what=""):
    assert res.status == 0, what
    bad = {k: (getattr(res, k), getattr(ores, k)) for k in KEYS if getattr(res, k) != getattr(ores, k)}
    assert not bad, (what, bad)
    assert res.totalDelay == ores.totalDelay, what
    a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
    b = np.frombuffer(oues, dtype=np.int32).reshape(-1, 16)
    diff = np.where((a != b).any(axis=1))[0]
    assert diff.size == 0, (what, diff[:5], a[diff[:3]], b[diff[:3]])


ORACLE_CASES = [
    # variant, nUE, overrides
    (0, 3000, {}), (1, 3000, {}), (0, 20000, {}), (1, 20000, {}), (1, 30000, {}),
    (0, 5000, dict(uniform=1, nGrantUL=12)),                       # BASELINE config 1 (Uniform, nUE=5000)
    (1, 9000, dict(uniform=1, nPreamble=54, backoff=2, nGrantUL=12, maxRarWindow=2)),
    (0, 1500, dict(nPreamble=2, backoff=3, nGrantUL=3, maxRarWindow=2, maxMsg2TxCount=3, accessTime=6)),
    (1, 1500, dict(nPreamble=2, backoff=3, nGrantUL=1, maxRarWindow=3, maxMsg2TxCount=0, accessTime=6)),
    (1, 4000, dict(nPreamble=1, backoff=1, nGrantUL=12, maxRarWindow=2, maxMsg2TxCount=3)),
    (0, 4000, dict(nPreamble=8, backoff=5, nGrantUL=12, maxRarWindow=2, maxMsg2TxCount=1)),
    (1, 1500, dict(nPreamble=200, backoff=1, nGrantUL=2, maxRarWindow=2, maxMsg2TxCount=3)),
    (0, 3000, dict(nPreamble=3, backoff=40, nGrantUL=12, maxRarWindow=6, maxMsg2TxCount=3, accessTime=6)),
    (1, 9000, dict(nPreamble=64, backoff=10, nGrantUL=8, maxRarWindow=5, maxMsg2TxCount=4, accessTime=10)),
    (0, 9000, dict(nPreamble=254, backoff=7, nGrantUL=200, maxRarWindow=255, maxMsg2TxCount=255)),
    (0, 1, {}), (1, 63, {}), (0, 64, {}), (1, 65, {}), (0, 1025, {}),  # ragged sizes around wave / workgroup width
    # reset storm: every expiry is a reset cycle and nobody is ever granted -> more re-join candidates per subframe than the
    # cluster kernel stages (falls back to the single-workgroup kernel, which walks them in index order)
    (1, 2600, dict(uniform=1, nPreamble=54, backoff=5, nGrantUL=1, maxRarWindow=1, maxMsg2TxCount=0, accessTime=10)),
]


@pytest.mark.parametrize("rng_mode", [0, 1])
@pytest.mark.parametrize("variant,nUE,kw", ORACLE_CASES)
def test_gpu_equals_oracle(pkg, ob, engine, variant, nUE, kw, rng_mode):
    seed = 5
    cfg = pkg.make_cfg(nUE, variant=variant, rng_mode=rng_mode, seed=seed, **kw)
    (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
    ores, oues = ob.run_trial(ob.make_cfg(nUE, variant=variant, **kw), ob.Rng(rng_mode, seed))
    assert_same(pkg, res, logs, ores, oues, (variant, nUE, kw, rng_mode))


@pytest.mark.parametrize("case", ["beta", "noma_default", "noma_uniform", "noma_odd", "noma_g54", "noma_seed2"])
def test_gpu_reproduces_reference_files(pkg, engine, case):
    """glibc mode, nUE sweep chained through the draw-stream offset like the reference's single
    srand() per seed: Results.txt bytes, stdout block and Logs.txt SHA-256 of EVERY trial of the
    reference run (up to nUE=100 000) — no oracle involved, only the committed fixtures."""
    g = load_golden(case)
    variant = 0 if g["variant"] == "BETA_C" else 1
    blocks = split_stdout_blocks(g["stdout"])
    offsets = {}
    for k, tr in enumerate(g["trials"]):
        off = offsets.get(tr["seed"], 0)
        cfg = pkg.make_cfg(tr["nUE"], variant=variant, rng_mode=pkg.RNG_GLIBC, seed=tr["seed"], stream_offset=off,
                           **g["cfg_overrides"])
        (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
        assert res.status == 0
        offsets[tr["seed"]] = off + res.draws
        txt = pkg.format_results(cfg, res, 0.0).decode()
        if variant == 0:
            txt = txt[:-len("0.000000")]
        assert txt == tr["results_text"], (case, tr["nUE"])
        so = pkg.format_stdout(cfg, res, 0.0).decode()
        so = "".join(l + "\n" for l in so.split("\n") if l and not l.startswith("Latency:"))
        assert so == blocks[k], (case, tr["nUE"])
        text = pkg.format_logs(logs, tr["nUE"])
        assert len(text) == tr["logs_bytes"]
        assert hashlib.sha256(text).hexdigest() == tr["logs_sha256"], (case, tr["nUE"])


@pytest.mark.parametrize("case", ["beta", "noma_default"])
def test_gpu_batch_kernel_in_the_reference_stream(pkg, ob, engine, case):
    """prach::batch_kernel<16, true> — one workgroup per trial, the draws at their positions in the reference's own rand() stream (count pass, prefix
    over the groups in index order, select pass) — is what `prach_sim -t 100` runs per sweep point.  (1) The reference's own run, chained through
    the stream offsets, with one workgroup per trial: Results.txt, stdout block and Logs.txt SHA-256 of every trial up to nUE = 100 000;
    (2) several seeds of one point in ONE call (as the CLI issues them) against the oracle, every field of every UE."""
    g = load_golden(case)
    variant = 0 if g["variant"] == "BETA_C" else 1
    blocks = split_stdout_blocks(g["stdout"])
    engine.set("cluster", 1)
    try:
        offsets = {}
        for k, tr in enumerate(g["trials"]):
            off = offsets.get(tr["seed"], 0)
            cfg = pkg.make_cfg(tr["nUE"], variant=variant, rng_mode=pkg.RNG_GLIBC, seed=tr["seed"], stream_offset=off, **g["cfg_overrides"])
            (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
            tm = engine.timing()
            assert res.status == 0 and tm.rec_mode == 4 and tm.fallback_trials == 0, (case, tr["nUE"], tm.rec_mode)
            offsets[tr["seed"]] = off + res.draws
            txt = pkg.format_results(cfg, res, 0.0).decode()
            if variant == 0:
                txt = txt[:-len("0.000000")]
            assert txt == tr["results_text"], (case, tr["nUE"])
            so = pkg.format_stdout(cfg, res, 0.0).decode()
            so = "".join(l + "\n" for l in so.split("\n") if l and not l.startswith("Latency:"))
            assert so == blocks[k], (case, tr["nUE"])
            text = pkg.format_logs(logs, tr["nUE"])
            assert len(text) == tr["logs_bytes"] and hashlib.sha256(text).hexdigest() == tr["logs_sha256"], (case, tr["nUE"])
    finally:
        engine.set("cluster", 0)
    # as the CLI issues a sweep point: many seeds in one call, automatic cluster size (100+ trials: one workgroup each)
    kw = dict(nGrantUL=12) if variant else {}
    cfgs = [pkg.make_cfg(20000, variant=variant, rng_mode=pkg.RNG_GLIBC, seed=s, stream_offset=1000 * s, **kw) for s in range(130)]
    res, logs = engine.run_trials(cfgs, want_logs=True)
    tm = engine.timing()
    assert tm.rec_mode == 4 and tm.cluster_size == 1 and tm.fallback_trials == 0
    for s in (0, 57, 129):
        rng = ob.Rng(ob.RNG_GLIBC, s)
        for _ in range(1000 * s):  # (the trial starts 1000 s values into its seed's stream)
            rng.next_glibc()
        ores, oues = ob.run_trial(ob.make_cfg(20000, variant=variant, **kw), rng)
        assert_same(pkg, res[s], logs[s], ores, oues, ("glibc batch", case, s))
    assert len({(r.nSuccessUE, r.draws) for r in res}) > 100  # (different seeds: different trials)


def test_full_size_100k_vs_oracle_and_properties(pkg, ob, engine):
    """BASELINE config 2 (nUE=100 000, Beta, 54 preambles, retx 10) at full size."""
    n = 100000
    for variant in (0, 1):
        cfg = pkg.make_cfg(n, variant=variant, rng_mode=pkg.RNG_PHILOX, seed=1)
        (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
        ores, oues = ob.run_trial(ob.make_cfg(n, variant=variant), ob.Rng(ob.RNG_PHILOX, 1))
        assert_same(pkg, res, logs, ores, oues, ("100k", variant))
        a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
        # size-independent properties of the procedure
        assert res.nSuccessUE + res.failedUEs == n and res.finalSuccessUEs == res.nSuccessUE
        assert int((a[:, 14] == 1).sum()) == res.nSuccessUE                      # msg4Flag count
        assert int(a[a[:, 14] == 1, 1].sum()) == res.sumTimer                    # timers of the successful UEs
        assert (a[a[:, 14] == 1, 2] == 0).all() and (a[a[:, 14] == 1, 12] == 1).all()  # done => active 0, msg2Flag 1
        assert (a[:, 7] < 54).all() and (a[:, 9] < 6).all() and (a[:, 10] <= 9).all()   # preamble / rarWindow / maxRarCounter ranges
        assert res.activeCheck == n and res.steps == res.time_exit == 10000 or res.nSuccessUE == n
        (res2,), (logs2,) = engine.run_trials([cfg], want_logs=True)             # idempotence / determinism
        assert bytes(logs2) == bytes(logs) and res2.as_dict() == res.as_dict()


def test_batch_equals_singles_and_mixed_modes(pkg, ob, engine):
    """Trials of one call run concurrently (one workgroup each); results are independent of batching,
    also with mixed RNG modes, variants and sizes in one call."""
    cfgs = [pkg.make_cfg(n, variant=v, rng_mode=r, seed=s, nGrantUL=gr)
            for (n, v, r, s, gr) in [(7000, 0, 1, 0, 54), (12000, 1, 1, 1, 12), (3000, 1, 0, 2, 12), (9000, 0, 0, 3, 20),
                                     (64, 1, 1, 4, 12), (15000, 1, 1, 5, 12), (11000, 0, 1, 6, 54), (2000, 0, 0, 7, 54)]]
    res_b, logs_b = engine.run_trials(cfgs, want_logs=True)
    for c, rb, lb in zip(cfgs, res_b, logs_b):
        (rs,), (ls,) = engine.run_trials([c], want_logs=True)
        assert rb.as_dict() == rs.as_dict() and bytes(lb) == bytes(ls)
        ores, oues = ob.run_trial(ob.make_cfg(c.nUE, variant=c.variant, nGrantUL=c.nGrantUL), ob.Rng(c.rng_mode, int(c.seed)))
        assert_same(pkg, rb, lb, ores, oues, (c.nUE, c.variant, c.rng_mode))


def test_many_concurrent_trials_sweep_times(pkg, ob, engine):
    """BASELINE config 3 shape (nUE sweep x --times), scaled to what the oracle checks in seconds:
    40 concurrent philox trials, aggregate checked per trial against the oracle."""
    cfgs = [pkg.make_cfg(n, variant=1, rng_mode=pkg.RNG_PHILOX, seed=s) for s in range(10) for n in (2000, 4000, 6000, 8000)]
    res, _ = engine.run_trials(cfgs)
    for c, r in zip(cfgs, res):
        o, _ = ob.run_trial(ob.make_cfg(c.nUE, variant=1), ob.Rng(ob.RNG_PHILOX, int(c.seed)), want_ues=False)
        assert (r.nSuccessUE, r.time_exit, r.collisionPreambles, r.totalPreambleTxop, r.sumTimer, r.draws) == \
               (o.nSuccessUE, o.time_exit, o.collisionPreambles, o.totalPreambleTxop, o.sumTimer, o.draws)


def test_config3_full_size_streaming_regime(pkg, ob, engine):
    """BASELINE config 3 AS THE BENCH LAUNCHES IT: nUE sweep 10k..100k x --times 100 = 1000 RandomAccessWithNOMA Philox trials
    in ONE call — one workgroup per trial, 8 + 4 byte hot records (the streaming regime; RandomAccessWithNOMA.c:216-221 runs
    the same grid serially).  Every counter of the 20 trials seed in {0, 1} x all ten nUE points against the oracle, every
    logged field of every UE of the two 100 000-UE ones; the other 980 trials through size-independent properties of the
    procedure; the whole call twice (determinism)."""
    points = list(range(10000, 100001, 10000))
    cfgs = [pkg.make_cfg(n, variant=pkg.VARIANT_WITHNOMA_C, rng_mode=pkg.RNG_PHILOX, seed=s) for s in range(100) for n in points]
    want = [k for k, c in enumerate(cfgs) if c.nUE == 100000 and int(c.seed) in (0, 1)]
    res, logs = engine.run_trials(cfgs, want_logs=want)
    tm = engine.timing()
    assert tm.launches == 1 and tm.workgroups == 1000 and tm.fallback_trials == 0  # G = 1: one workgroup per trial, no rerun
    import concurrent.futures as cf

    def oracle(k):
        c = cfgs[k]
        return k, ob.run_trial(ob.make_cfg(c.nUE, variant=1), ob.Rng(ob.RNG_PHILOX, int(c.seed)), want_ues=k in want)

    checked = [k for k, c in enumerate(cfgs) if int(c.seed) in (0, 1)]
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:  # (the oracle is plain C behind ctypes: no GIL)
        for k, (ores, oues) in ex.map(oracle, checked):
            r = res[k]
            assert r.status == 0
            bad = {f: (getattr(r, f), getattr(ores, f)) for f in KEYS if getattr(r, f) != getattr(ores, f)}
            assert not bad and r.totalDelay == ores.totalDelay, (cfgs[k].nUE, int(cfgs[k].seed), bad)
            if k in want:
                assert_same(pkg, r, logs[k], ores, oues, ("config3", cfgs[k].nUE, int(cfgs[k].seed)))
    seen = {}
    for c, r in zip(cfgs, res):
        n = c.nUE
        assert r.status == 0
        assert r.nSuccessUE + r.failedUEs == n and r.finalSuccessUEs == r.nSuccessUE and r.activeCheck == n
        assert (r.steps == 10000 and r.time_exit == 10000) or (r.nSuccessUE == n and r.steps == r.time_exit + 1)
        assert 0 < r.nSuccessUE <= n and r.preambleTxCount >= r.nSuccessUE      # every success transmitted at least once
        assert 17 * r.nSuccessUE <= r.sumTimer <= 10000 * r.nSuccessUE           # 1 + 11 + 6 subframes at the very least (Beta.c:340,378)
        assert r.collisionPreambles <= r.totalPreambleTxop and r.failCounts <= r.continueFaliedUEs
        assert r.draws >= 3 * n                                                  # two activation draws + a preamble per UE (WithNOMA:393-394)
        if n >= 50000:
            assert abs(r.nSuccessUE - 18800) < 700                               # 12 grants per 5 ms: the cell saturates (results.csv:5-10)
        seen.setdefault(n, set()).add((r.nSuccessUE, r.sumTimer, r.totalPreambleTxop))
    assert all(len(v) > 90 for v in seen.values())                               # the seeds really are different trials
    res2, _ = engine.run_trials(cfgs)
    assert [r.as_dict() for r in res2] == [r.as_dict() for r in res]


def test_grid_full_size_streaming_regime(pkg, ob, engine):
    """BASELINE configs[4]'s regime AS THE BENCH LAUNCHES IT ON ONE GPU (`grid_one_gpu`): --times 100 x the ten-point sweep = 1000 Beta.c-as-committed Philox trials
    in ONE call (512-thread workgroups, two trials per CU, the branch-free event body).  Every counter of the ten trials of seed 0 and of the 100 000-UE trials of seeds
    1..4 against the oracle, every logged field of every UE of the seed-0 100 000-UE trial; all 1000 through size-independent properties; both workgroup shapes agree."""
    points = list(range(10000, 100001, 10000))
    cfgs = [pkg.make_cfg(n, variant=pkg.VARIANT_BETA_C, rng_mode=pkg.RNG_PHILOX, seed=s) for s in range(100) for n in points]
    want = [k for k, c in enumerate(cfgs) if c.nUE == 100000 and int(c.seed) == 0]
    res, logs = engine.run_trials(cfgs, want_logs=want)
    tm = engine.timing()
    assert tm.launches == 1 and tm.workgroups == 1000 and tm.fallback_trials == 0 and tm.rec_mode == 4
    import concurrent.futures as cf

    def oracle(k):
        c = cfgs[k]
        return k, ob.run_trial(ob.make_cfg(c.nUE, variant=0), ob.Rng(ob.RNG_PHILOX, int(c.seed)), want_ues=k in want)

    checked = [k for k, c in enumerate(cfgs) if int(c.seed) == 0 or (c.nUE == 100000 and int(c.seed) <= 4)]
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for k, (ores, oues) in ex.map(oracle, checked):
            r = res[k]
            bad = {f: (getattr(r, f), getattr(ores, f)) for f in KEYS if getattr(r, f) != getattr(ores, f)}
            assert r.status == 0 and not bad and r.totalDelay == ores.totalDelay, (cfgs[k].nUE, int(cfgs[k].seed), bad)
            if k in want:
                assert_same(pkg, r, logs[k], ores, oues, ("grid", cfgs[k].nUE, int(cfgs[k].seed)))
    for c, r in zip(cfgs, res):
        n = c.nUE
        assert r.status == 0 and r.nSuccessUE + r.failedUEs == n and r.activeCheck == n
        assert (r.steps == 10000 and r.time_exit == 10000) or (r.nSuccessUE == n and r.steps == r.time_exit + 1)
        assert 17 * r.nSuccessUE <= r.sumTimer <= 10000 * r.nSuccessUE and r.collisionPreambles <= r.totalPreambleTxop
        if n <= 70000:
            assert r.nSuccessUE == n  # 54 grants per 5 ms serve every UE of the small points (the reference's own sweep)
    engine.set("batch_waves", 16)
    try:
        res16, _ = engine.run_trials(cfgs)
    finally:
        engine.set("batch_waves", 0)
    assert [r.as_dict() for r in res16] == [r.as_dict() for r in res]


def test_max_steps_and_stream_offset(pkg, ob, engine):
    cfg = pkg.make_cfg(8000, variant=1, rng_mode=0, seed=9, max_steps=2500)
    (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
    ores, oues = ob.run_trial(ob.make_cfg(8000, variant=1, max_steps=2500), ob.Rng(0, 9))
    assert_same(pkg, res, logs, ores, oues, "max_steps")
    rng = ob.Rng(0, 9)
    o1, _ = ob.run_trial(ob.make_cfg(5000, variant=0), rng, want_ues=False)
    o2, u2 = ob.run_trial(ob.make_cfg(6000, variant=0), rng)
    cfg2 = pkg.make_cfg(6000, variant=0, rng_mode=0, seed=9, stream_offset=int(o1.draws))
    (r2,), (l2,) = engine.run_trials([cfg2], want_logs=True)
    assert_same(pkg, r2, l2, o2, u2, "stream_offset")


def test_stream_budget_retry(pkg, ob, engine):
    """glibc draw-stream window too small on the first attempt: the engine reruns with a larger one."""
    kw = dict(nPreamble=1, backoff=1, nGrantUL=12, maxRarWindow=2, maxMsg2TxCount=3)
    engine.set("stream_factor", 20)
    try:
        cfg = pkg.make_cfg(4000, variant=1, rng_mode=0, seed=3, **kw)
        (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
        assert engine.timing().launches >= 2
    finally:
        engine.set("stream_factor", 0)
    ores, oues = ob.run_trial(ob.make_cfg(4000, variant=1, **kw), ob.Rng(0, 3))
    assert_same(pkg, res, logs, ores, oues, "retry")


def test_error_codes(pkg, engine):
    bad = pkg.make_cfg(1000)
    bad.nUE = 0
    with pytest.raises(pkg.PrachError) as ei:
        engine.run_trials([bad])
    assert ei.value.status == -1
    big = pkg.make_cfg(1000, nPreamble=255)
    with pytest.raises(pkg.PrachError) as ei:
        engine.run_trials([big])
    assert ei.value.status == -2
    assert pkg.lib().prach_run_trials(None, None, 0, None, None) == -1


def test_cli_drop_in_files(pkg, engine, tmp_path):
    """prach_sim with the reference's flags writes the reference's files: first two points of the
    default RandomAccessWithNOMA run (golden) byte-for-byte, stdout included."""
    g = load_golden("noma_default")
    p = subprocess.run([pkg.CLI_PATH, "--sweep", "10000:20000:10000", "--out", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    blocks = split_stdout_blocks(g["stdout"])
    assert p.stdout.startswith("Traffic model: Beta\n\n")
    assert split_stdout_blocks(p.stdout) == blocks[:2]
    for tr in g["trials"][:2]:
        d = tmp_path / tr["dir"]
        assert (d / tr["results_file"]).read_text() == tr["results_text"]
        assert hashlib.sha256((d / tr["logs_file"]).read_bytes()).hexdigest() == tr["logs_sha256"]


@pytest.mark.parametrize("G", [1, 2, 4, 16, 64])
def test_cluster_sizes_agree(pkg, ob, engine, G):
    """The Philox production kernel with G workgroups per trial (interleaved ownership + granule exchange)
    gives the same trial for every cluster size, = the oracle."""
    engine.set("cluster", G)
    try:
        for variant, n in ((0, 20000), (1, 20000)):
            cfg = pkg.make_cfg(n, variant=variant, rng_mode=pkg.RNG_PHILOX, seed=17)
            (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
            assert engine.timing().workgroups == G
            ores, oues = ob.run_trial(ob.make_cfg(n, variant=variant), ob.Rng(ob.RNG_PHILOX, 17))
            assert_same(pkg, res, logs, ores, oues, (G, variant))
    finally:
        engine.set("cluster", 0)


def test_several_clusters_in_one_launch(pkg, ob, engine):
    """4 trials x 16 workgroups in one launch (cluster id = blockIdx %% trials), mixed variants and sizes."""
    engine.set("cluster", 16)
    try:
        cfgs = [pkg.make_cfg(n, variant=v, rng_mode=pkg.RNG_PHILOX, seed=s) for (n, v, s) in
                [(25000, 0, 1), (18000, 1, 2), (30000, 1, 3), (16000, 0, 4)]]
        res, logs = engine.run_trials(cfgs, want_logs=True)
        assert engine.timing().workgroups == 64 and engine.timing().launches == 1
        for c, r, l in zip(cfgs, res, logs):
            ores, oues = ob.run_trial(ob.make_cfg(c.nUE, variant=c.variant), ob.Rng(ob.RNG_PHILOX, int(c.seed)))
            assert_same(pkg, r, l, ores, oues, (c.nUE, c.variant))
    finally:
        engine.set("cluster", 0)


def test_xcd_packed_clusters(pkg, ob, engine):
    """The lean cluster kernel's XCD-packed launch (a cluster on the blocks of equal blockIdx % 8, its granules resident in that XCD's L2
    once the cluster has VERIFIED that it runs on one XCD) against the plain launch and the oracle: one cluster, eleven clusters (two
    chunks of eight, the second one partly empty: blocks of trials past the last one leave at once), and a cluster size that cannot be
    packed (64 workgroups > the CUs of an XCD: plain launch, same results).  prach_timing.xcd_packed reports what was launched."""
    one = [pkg.make_cfg(30000, variant=0, rng_mode=pkg.RNG_PHILOX, seed=5)]
    many = [pkg.make_cfg(n, variant=v, rng_mode=pkg.RNG_PHILOX, seed=s) for s, (n, v) in
            enumerate([(9000, 0), (12000, 1), (7000, 0), (15000, 1), (5000, 0), (11000, 0), (8000, 1), (14000, 0), (6000, 1), (10000, 0), (13000, 1)])]
    try:
        for cfgs, G in ((one, 16), (many, 8), (one, 64)):
            engine.set("cluster", G)
            out = {}
            for pack in (1, 0):
                engine.set("xcd_pack", pack)
                res, logs = engine.run_trials(cfgs, want_logs=True)
                tm = engine.timing()
                assert tm.rec_mode == 3 and tm.fallback_trials == 0 and tm.spin_timeouts == 0, (G, pack, tm.rec_mode)
                assert tm.xcd_packed == (1 if pack and G <= 32 else 0), (G, pack, tm.xcd_packed)
                out[pack] = ([r.as_dict() for r in res], [bytes(l) for l in logs])
            assert out[0] == out[1], G
            for c, r, l in zip(cfgs, *engine.run_trials(cfgs, want_logs=True)):
                ores, oues = ob.run_trial(ob.make_cfg(c.nUE, variant=c.variant), ob.Rng(ob.RNG_PHILOX, int(c.seed)))
                assert_same(pkg, r, l, ores, oues, ("packed", G, c.nUE))
        # clusters in the reference's own rand() stream (two exchanges per subframe) launch the same way: on the lean kernel (round 3: the count
        # pass / count exchange / select pass on LDS-resident records) and, with engine option "fast" 0, on the general kernel
        engine.set("cluster", 16)
        gcfg = pkg.make_cfg(24000, variant=1, rng_mode=pkg.RNG_GLIBC, seed=3)
        ores, oues = ob.run_trial(ob.make_cfg(24000, variant=1), ob.Rng(ob.RNG_GLIBC, 3))
        for fast in (1, 0):
            engine.set("fast", fast)
            out = {}
            for pack in (1, 0):
                engine.set("xcd_pack", pack)
                (res,), (logs,) = engine.run_trials([gcfg], want_logs=True)
                tm = engine.timing()
                assert (tm.rec_mode == 3) == bool(fast) and tm.cluster_size == 16 and tm.xcd_packed == pack and tm.fallback_trials == 0, (fast, pack, tm.rec_mode, tm.xcd_packed)
                out[pack] = (res.as_dict(), bytes(logs))
            assert out[0] == out[1]
            engine.set("xcd_pack", 1)
            (res,), (logs,) = engine.run_trials([gcfg], want_logs=True)
            assert_same(pkg, res, logs, ores, oues, ("packed glibc cluster", fast))
    finally:
        engine.set("fast", 1)
        engine.set("xcd_pack", 1)
        engine.set("cluster", 0)


def test_cluster_capacity_fallback_is_exact(pkg, ob, engine):
    """backoff 1 makes every retransmission land on the same subframe: thousands of special events per subframe
    exceed the cluster kernel's per-subframe LDS capacities; the engine reruns such trials on the
    one-workgroup kernel (global scratch, no caps) — still bit-exact."""
    kw = dict(nPreamble=3, backoff=1, nGrantUL=12, maxRarWindow=2, maxMsg2TxCount=3)
    cfg = pkg.make_cfg(60000, variant=1, rng_mode=pkg.RNG_PHILOX, seed=2, **kw)
    (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
    tm = engine.timing()
    # cluster attempt + exact rerun on the batch kernel — whose join lists (every UE of the trial opens its window in the same subframe here) fill too: once more with
    # full-size lists; every rerun is counted and reported
    assert tm.launches >= 2 and 1 <= tm.fallback_trials <= 3
    ores, oues = ob.run_trial(ob.make_cfg(60000, variant=1, **kw), ob.Rng(ob.RNG_PHILOX, 2))
    assert_same(pkg, res, logs, ores, oues, "fallback")


def test_cluster_residency_is_explicit(pkg, ob, engine):
    """The workgroups of a cluster wait for each other, so a cluster launch is capped by what the runtime's occupancy query
    admits at once for the kernel and its LDS size (prach_timing.resident_limit); the "resident" hook narrows it."""
    cfgs = [pkg.make_cfg(20000, variant=1, rng_mode=pkg.RNG_PHILOX, seed=s) for s in range(3)]
    engine.set("cluster", 16)
    try:
        res, _ = engine.run_trials(cfgs)
        tm = engine.timing()
        assert tm.resident_limit >= 256 and tm.cluster_size == 16 and tm.workgroups == 48 and tm.fallback_trials == 0
        engine.set("resident", 30)  # 3 trials x 16 workgroups do not fit: 8 per trial do
        res2, _ = engine.run_trials(cfgs)
        tm = engine.timing()
        assert tm.resident_limit == 30 and tm.cluster_size == 8 and tm.workgroups == 24 and tm.fallback_trials == 0
        assert [r.as_dict() for r in res2] == [r.as_dict() for r in res]
    finally:
        engine.set("resident", 0)
        engine.set("cluster", 0)


def test_two_engines_share_one_device(pkg, ob):
    """Two engines (two host threads, two HIP streams) on ONE device, each launching 4 trials x 64 workgroups at the same
    time: together more workgroups than the device holds.  Members of a cluster are consecutive blocks, so whole clusters
    are resident in dispatch order; a cluster whose peers are late is bounded by the peer-wait limit and rerun exactly on
    the one-workgroup kernel — never silently: prach_timing reports how many."""
    import threading
    cfgs = [pkg.make_cfg(30000 + 2000 * k, variant=k & 1, rng_mode=pkg.RNG_PHILOX, seed=40 + k) for k in range(4)]
    out = {}

    def run(tag):
        eng = pkg.Engine(0)
        try:
            eng.set("cluster", 64)
            for rep in range(2):
                res, _ = eng.run_trials(cfgs)
                tm = eng.timing()
                out[(tag, rep)] = ([r.as_dict() for r in res], tm.fallback_trials, tm.spin_timeouts, tm.cluster_size)
        finally:
            eng.close()

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert len(out) == 4
    exp = []
    for c in cfgs:
        o, _ = ob.run_trial(ob.make_cfg(c.nUE, variant=c.variant), ob.Rng(ob.RNG_PHILOX, int(c.seed)), want_ues=False)
        exp.append(o)
    for (tag, rep), (res, nfb, nto, G) in out.items():
        assert 0 <= nto <= nfb <= 4, (tag, rep, nfb, nto)
        for r, o in zip(res, exp):
            assert r["status"] == 0
            assert {k: r[k] for k in KEYS} == {k: getattr(o, k) for k in KEYS}, (tag, rep)


def test_legacy_kernel_option(pkg, ob, engine):
    engine.set("legacy", 1)
    try:
        cfg = pkg.make_cfg(9000, variant=1, rng_mode=pkg.RNG_PHILOX, seed=8)
        (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
    finally:
        engine.set("legacy", 0)
    ores, oues = ob.run_trial(ob.make_cfg(9000, variant=1), ob.Rng(ob.RNG_PHILOX, 8))
    assert_same(pkg, res, logs, ores, oues, "legacy")


def test_cli_noma_program(pkg, ob, engine, tmp_path):
    """prach_sim --program noma: NOMA.c's output surface (one line per (seed, nUE), appended per-nUE files, Done)."""
    p = subprocess.run([pkg.CLI_PATH, "--program", "noma", "--times", "2", "--sweep", "3000:6000:3000", "--out", str(tmp_path)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    exp = []
    for seed in range(2):
        for n in (3000, 6000):
            ocfg = ob.make_noma_cfg(n)
            ores, _ = ob.noma_run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, seed), want_ues=False)
            exp.append(ob.noma_format_line(ocfg, ores).decode())
        exp.append("Done\n")
    assert p.stdout == "".join(exp)
    assert (tmp_path / "TestResults" / "Sector_3000_Result.txt").read_text() == exp[0] + exp[3]


def test_cli_noma_program_reference_stream(pkg, tmp_path):
    """prach_sim --program noma --rng glibc: NOMA.c's own program, byte for byte — its first two seeds' stdout (ten chained sweep points
    each, "Done" after every seed) and the appended per-nUE files, against what the compiled reference printed (tests/golden/noma_c.json)."""
    from conftest import load_golden
    g = load_golden("noma_c")
    lines = g["stdout"].split("\n")
    want, done = [], 0
    for l in lines:
        want.append(l)
        done += l == "Done"
        if done == 2:
            break
    p = subprocess.run([pkg.CLI_PATH, "--program", "noma", "--rng", "glibc", "--times", "2", "--out", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    assert p.stdout == "\n".join(want) + "\n"
    first = [l for l in want if l != "Done"]
    assert (tmp_path / "TestResults" / "Sector_10000_Result.txt").read_text() == first[0] + "\n" + first[10] + "\n"


def test_cli_philox_grid_in_one_call(pkg, ob, engine, tmp_path):
    """--rng philox: the --times x sweep grid is one batched call; stdout order and files follow the reference."""
    p = subprocess.run([pkg.CLI_PATH, "--rng", "philox", "--times", "2", "--sweep", "4000:8000:4000", "--out", str(tmp_path), "--logs", "1"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    blocks = split_stdout_blocks(p.stdout)
    assert len(blocks) == 4
    k = 0
    for seed in range(2):
        for n in (4000, 8000):
            ocfg = ob.make_cfg(n, variant=1)
            ores, oues = ob.run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, seed))
            assert blocks[k] == ob.format_stdout(ocfg, ores).decode()
            d = tmp_path / "NomaBetaResults"
            assert (d / f"{seed}_54_{n}_Results.txt").read_bytes() == ob.format_results(ocfg, ores)
            assert (d / f"{seed}_54_UE{n:05d}_Logs.txt").read_bytes() == ob.format_logs(oues, n)
            k += 1


def test_cli_results_csv(pkg, tmp_path):
    """prach_sim --program beta --csv: the results.csv AveragePerformance.py would write from the Results.txt files of the
    same run (mean over the seeds in seed order, np.around(., 3), csv.writer float repr, CRLF), in both RNG modes."""
    import numpy as np
    for rng in ("philox", "glibc"):
        out = tmp_path / rng
        out.mkdir()
        (out / "BasicBetaSimulationResults").mkdir()
        csvf = out / "results.csv"
        p = subprocess.run([pkg.CLI_PATH, "--program", "beta", "--rng", rng, "--times", "3", "--sweep", "3000:9000:3000", "--out", str(out),
                            "--logs", "0", "--csv", str(csvf)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        rows = []
        for n in (3000, 6000, 9000):
            acc = [0.0] * 6
            for seed in range(3):  # AveragePerformance.py:10-19
                data = [float(l.strip()) for l in (out / "BasicBetaSimulationResults" / f"{seed}_54_{n}_Results.txt").read_text().split("\n") if l.strip()]
                for i in range(6):
                    acc[i] += data[i]
            rows.append(",".join(repr(float(x)) for x in np.around(np.array(acc) / 3, 3)) + "\r\n")
        assert csvf.read_bytes() == "".join(rows).encode()
    assert subprocess.run([pkg.CLI_PATH, "--csv", "x.csv"], capture_output=True).returncode == 255  # needs --program beta


def test_cli_sharded_over_workers(pkg, tmp_path):
    """`prach_sim --gpus N` / `--devices LIST`: host C forks one worker per device BEFORE any HIP call, deals the --times x sweep
    grid by descending cost (glibc mode: whole seeds, the sweep of a seed is chained), every worker writes its own trials' files
    and the parent merges the results through shared memory, printing in the reference's order.  Rehearsed on one GPU with two
    workers on device 0: stdout, every file and results.csv equal the single-worker run, in both RNG modes."""
    for rng in ("philox", "glibc"):
        outs = {}
        for tag, extra in (("one", []), ("two", ["--devices", "0,0"])):
            d = tmp_path / f"{rng}_{tag}"
            d.mkdir()
            (d / "BasicBetaSimulationResults").mkdir()
            p = subprocess.run([pkg.CLI_PATH, "--program", "beta", "--rng", rng, "--times", "4", "--sweep", "3000:9000:3000", "--out", str(d),
                                "--logs", "1", "--csv", str(d / "results.csv")] + extra, capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-2000:]
            files = {f.name: f.read_bytes() for f in sorted((d / "BasicBetaSimulationResults").iterdir())}
            # (the sixth line of a Results.txt and the Latency line are wall clock)
            files = {k: (b"\n".join(v.split(b"\n")[:5]) if k.endswith("_Results.txt") else v) for k, v in files.items()}
            so = "\n".join(l for l in p.stdout.split("\n") if not l.startswith("Latency:"))
            csv5 = [row.split(b",")[:5] for row in (d / "results.csv").read_bytes().split(b"\r\n")]
            outs[tag] = (so, files, csv5)
        assert outs["one"] == outs["two"], rng
        assert len(outs["one"][1]) == 24 and len(split_stdout_blocks(outs["one"][0])) == 12


def test_cli_results_csv_vs_reference_script(pkg, tmp_path):
    """The published experiment end to end on the GPU: `prach_sim --program beta -g 12 -t 100 --rng philox --csv` — 1000 trials
    in one call — against the results.csv the reference's own AveragePerformance.py wrote from the oracle's files for the
    same 1000 (seed, nUE) trials (tests/golden/results_csv.json).  The sixth column is the reference's wall clock: the
    first five must be byte-identical."""
    import base64
    import json
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "results_csv.json")))
    want = base64.b64decode(fx["results_csv_b64"]).split(b"\r\n")[:10]
    (tmp_path / "BasicBetaSimulationResults").mkdir()
    csvf = tmp_path / "results.csv"
    p = subprocess.run([pkg.CLI_PATH, "--program", "beta", "-g", "12", "-t", "100", "--rng", "philox", "--logs", "0", "--out", str(tmp_path),
                        "--csv", str(csvf)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    got = csvf.read_bytes().split(b"\r\n")
    assert len(got) == 11 and got[10] == b""
    for g, w in zip(got[:10], want):
        assert g.split(b",")[:5] == w.split(b",")[:5], (g, w)
    # and the per-seed files are the oracle's (first five lines)
    for key in ("0_10000", "57_60000", "99_100000"):
        s_, n_ = key.split("_")
        txt = (tmp_path / "BasicBetaSimulationResults" / f"{s_}_54_{n_}_Results.txt").read_text()
        assert txt.split("\n")[:5] == fx["results_txt"][key].split("\n")[:5]


def test_sharded_sweep_driver_two_ranks(pkg, ob, tmp_path):
    """sweep.py (BASELINE config 5 shape) with 2 ranks rehearsed on one GPU (gloo): sharding + ONE all-reduce +
    row gather + results.csv; the aggregate equals the oracle's trial by trial."""
    import json
    import sys
    from conftest import ROOT
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "5g-nr-randomaccess_amd", "sweep.py"), "--times", "3", "--sweep",
           "3000:9000:3000", "--out", str(tmp_path), "--backend", "gloo", "--same-device"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    summary = json.loads([l for l in p.stdout.split("\n") if l.startswith("{")][-1])
    texts = {n: [] for n in (3000, 6000, 9000)}
    succ = {n: 0 for n in texts}
    upd = 0
    for s in range(3):
        for n in texts:
            ocfg = ob.make_cfg(n, variant=0)
            ores, _ = ob.run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, s), want_ues=False)
            texts[n].append(ob.format_results(ocfg, ores).decode() + "0.000000")
            succ[n] += ores.nSuccessUE
            upd += n * ores.steps
    assert summary["updates"] == upd
    assert summary["success_ratio"] == {str(n): succ[n] / (3 * n) for n in texts}
    assert (tmp_path / "results.csv").read_bytes() == pkg.results_csv([texts[n] for n in (3000, 6000, 9000)])


def _run_bench_ranks(nranks, backend, extra, port):
    import json
    from conftest import ROOT
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "1", "--warmup", "0",
           "--times", "3", "--backend", backend] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.split("\n") if l.startswith("{")][-1])


def _run_bench_plain(nranks, backend, extra):
    """`python bench.py --gpus N` WITHOUT torch.distributed.run: bench.py must start its N ranks itself (never a silent one-GPU run)."""
    import json
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "1", "--warmup", "0", "--times", "3", "--backend", backend] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.split("\n") if l.startswith("{")][-1])


def _check_grid_line(pkg, ob, out, nranks):
    """bench.py's N>1 line is BASELINE configs[4] (the --times x sweep grid, strong scaling): the aggregates equal the
    oracle's, trial by trial, and so does results.csv."""
    import hashlib
    assert out["n_gpus"] == nranks and out["scaling"] == "strong" and out["config"]["trials_per_step"] == 30
    assert out["value_per_gpu"] == out["value"] / nranks and "scaling_reference" not in out  # (no pasted one-GPU constant)
    assert len(out["per_rank_sim_seconds"]) == nranks and out["imbalance"] >= 0
    texts = {n: [] for n in range(10000, 100001, 10000)}
    succ = {n: 0 for n in texts}
    upd = 0
    import concurrent.futures as cf

    def one(job):
        s, n = job
        ocfg = ob.make_cfg(n, variant=0)
        ores, _ = ob.run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, s), want_ues=False)
        return s, n, ob.format_results(ocfg, ores).decode() + "0.000000", ores.nSuccessUE, n * ores.steps

    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for s, n, txt, ns, u in ex.map(one, [(s, n) for s in range(3) for n in texts]):
            texts[n].append((s, txt)); succ[n] += ns; upd += u
    assert out["config"]["updates_per_step"] == upd
    assert out["success_ratio"] == {str(n): succ[n] / (3 * n) for n in texts}
    csv = pkg.results_csv([[t for _, t in sorted(texts[n])] for n in texts])
    assert out["results_csv_sha256"] == hashlib.sha256(csv).hexdigest()


def test_bench_grid_two_ranks_rehearsal(pkg, ob):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run), rehearsed on ONE GPU: both ranks on cuda:0, gloo
    for the collective (a one-GPU box has no second device for RCCL)."""
    _check_grid_line(pkg, ob, _run_bench_ranks(2, "gloo", ["--same-device"], 29541), 2)


def test_bench_grid_two_ranks_plain_launch(pkg, ob):
    """The same started as a plain `python bench.py --gpus 2 --same-device --backend gloo` (no torch.distributed.run, no rendezvous
    variables): bench.py launches its two ranks itself and the line says n_gpus = 2 with the gloo backend."""
    out = _run_bench_plain(2, "gloo", ["--same-device"])
    assert out["config"]["collective_backend"] == "gloo"
    _check_grid_line(pkg, ob, out, 2)


def test_engine_arena_growth_paths(pkg, ob, capfd):
    """The engine's device arena (prach_engine.hip ensure_arena): one reserved virtual range that physical memory is mapped into in pieces as calls grow —
    over a small -> large -> small -> larger sequence of calls; the same with `plain_arena` (one hipMalloc allocation, re-allocated when it grows); and with
    the test hook `vmm_fail_after` = 1, which lets the range map ONE piece and then refuses (as a device out of mappable memory would): the engine says so on
    stderr — once — and carries on with a plain allocation.  Results are the same byte for byte in all three, and the oracle's where checked."""
    def seq(eng):
        out = []
        for ntr, nue, steps in ((3, 3000, 0), (300, 20000, 1500), (2, 3000, 0), (400, 20000, 1500)):
            cfgs = [pkg.make_cfg(nue, variant=s % 2, rng_mode=pkg.RNG_PHILOX, seed=s, max_steps=steps) for s in range(ntr)]
            res, _ = eng.run_trials(cfgs)
            assert all(r.status == 0 for r in res)
            out.append([bytes(r) for r in res])
        return out
    want = None
    for opts in ({}, {"plain_arena": 1}, {"vmm_fail_after": 1}):
        eng = pkg.Engine(0)
        for k, v in opts.items():
            eng.set(k, v)
        capfd.readouterr()
        got = seq(eng)
        err = capfd.readouterr().err
        eng.close()
        assert err.count("falling back to hipMalloc") == (1 if "vmm_fail_after" in opts else 0), (opts, err[-500:])
        if want is None:
            want = got
        assert got == want, opts
    ores, _ = ob.run_trial(ob.make_cfg(20000, variant=1, max_steps=1500), ob.Rng(ob.RNG_PHILOX, 299), want_ues=False)
    r = pkg.PrachResult.from_buffer_copy(want[1][299])
    assert (r.nSuccessUE, r.collisionPreambles, r.totalPreambleTxop, r.sumTimer, r.draws) == (ores.nSuccessUE, ores.collisionPreambles, ores.totalPreambleTxop, ores.sumTimer, ores.draws)


def test_bench_grid_four_ranks_on_one_device(pkg, ob):
    """The widest rehearsal a one-GPU box admits (its process guard allows six processes with the card open: this test process, bench.py's launcher and four
    ranks — five ranks were killed by it): a plain `python bench.py --gpus 4 --same-device --backend gloo` — four engines alive on one device, each with its
    own whole-memory address reservation for its arena, thirty trials dealt by the measured cost table — gives n_gpus = 4 and the oracle's aggregates and
    results.csv.  (World 8 itself: the gloo test of tests/test_dist_cpu.py on the CPU, and the driver's scaling run on the 8-GPU node.)"""
    out = _run_bench_plain(4, "gloo", ["--same-device"])
    assert out["config"]["collective_backend"] == "gloo" and len(out["per_rank_sim_seconds"]) == 4
    _check_grid_line(pkg, ob, out, 4)


def test_cli_four_workers_on_one_device(pkg, tmp_path):
    """`prach_sim --devices 0,0,0,0`: four forked workers on one device against the one-worker run — stdout, files and results.csv identical."""
    outs = {}
    for tag, extra in (("one", []), ("four", ["--devices", "0,0,0,0"])):
        d = tmp_path / tag
        d.mkdir()
        (d / "BasicBetaSimulationResults").mkdir()
        p = subprocess.run([pkg.CLI_PATH, "--program", "beta", "--rng", "philox", "--times", "4", "--sweep", "3000:9000:3000", "--out", str(d),
                            "--logs", "1", "--csv", str(d / "results.csv")] + extra, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        files = {f.name: f.read_bytes() for f in sorted((d / "BasicBetaSimulationResults").iterdir())}
        files = {k: (b"\n".join(v.split(b"\n")[:5]) if k.endswith("_Results.txt") else v) for k, v in files.items()}
        so = "\n".join(l for l in p.stdout.split("\n") if not l.startswith("Latency:"))
        csv5 = [row.split(b",")[:5] for row in (d / "results.csv").read_bytes().split(b"\r\n")]
        outs[tag] = (so, files, csv5)
    assert outs["one"] == outs["four"]


def test_bench_grid_two_ranks_rccl(pkg, ob):
    """The same over RCCL (backend nccl: all_reduce on device tensors + gather_object), one rank per GPU — needs two devices."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: the RCCL path needs two devices (the driver's 8-GPU scaling run exercises it)")
    _check_grid_line(pkg, ob, _run_bench_ranks(2, "nccl", [], 29542), 2)


def test_beyond_reference_sizes(pkg, ob, engine):
    """nUE = 250 000 (2.5x the reference's largest point): still bit-exact vs the oracle, and one workgroup cluster
    per trial still covers it (interleaved ownership scales with nUE, not with a per-CU capacity)."""
    n = 250000
    cfg = pkg.make_cfg(n, variant=0, rng_mode=pkg.RNG_PHILOX, seed=3, max_steps=4000)
    (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
    ores, oues = ob.run_trial(ob.make_cfg(n, variant=0, max_steps=4000), ob.Rng(ob.RNG_PHILOX, 3))
    assert_same(pkg, res, logs, ores, oues, "250k")


def test_legacy_kernel_glibc_mode(pkg, ob, engine):
    """The one-workgroup kernel (exact fallback) in glibc mode, forced."""
    engine.set("legacy", 1)
    try:
        cfg = pkg.make_cfg(12000, variant=0, rng_mode=pkg.RNG_GLIBC, seed=6)
        (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
        assert engine.timing().workgroups == 1
    finally:
        engine.set("legacy", 0)
    ores, oues = ob.run_trial(ob.make_cfg(12000, variant=0), ob.Rng(ob.RNG_GLIBC, 6))
    assert_same(pkg, res, logs, ores, oues, "legacy glibc")


def test_device_glibc_stream_matches_libc(pkg, engine):
    """The rand() stream generated on the device (matrix-power jump-ahead + one wavefront per chunk) == libc."""
    import ctypes as C2
    libc = C2.CDLL("libc.so.6")
    for seed, first, n in ((0, 0, 200_000), (7, 123_457, 150_001), (2022, 63_488 * 3 - 5, 70_000)):
        libc.srand(seed)
        ref = np.fromiter((libc.rand() for _ in range(first + n)), dtype=np.int64, count=first + n)[first:]
        dev = engine.device_glibc_stream(seed, first, n).astype(np.int64)
        assert (dev == ref).all(), (seed, first)
    big = engine.device_glibc_stream(3, 40_000_000, 1_000_000)
    assert (big == pkg.glibc_stream(3, 40_000_000, 1_000_000)).all()


def test_cli_multi_seed_glibc_matches_reference(pkg, engine, tmp_path):
    """`prach_sim -d 0 -p 30 -b 40 -g 20 -rc 3 -mrc 20 -t 3`: the reference's 3-seed run (golden noma_seed2), first two
    sweep points of every seed — stdout in the reference's order, Results.txt and Logs.txt byte-identical.  Internally
    the seeds of one sweep point run concurrently, each on its own chained rand() stream."""
    g = load_golden("noma_seed2")
    p = subprocess.run([pkg.CLI_PATH] + g["argv"] + ["--sweep", "10000:20000:10000", "--out", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    blocks = split_stdout_blocks(g["stdout"])
    exp_blocks = []
    for k, tr in enumerate(g["trials"]):
        if tr["nUE"] <= 20000:
            exp_blocks.append(blocks[k])
            d = tmp_path / tr["dir"]
            assert (d / tr["results_file"]).read_text() == tr["results_text"], (tr["seed"], tr["nUE"])
            assert hashlib.sha256((d / tr["logs_file"]).read_bytes()).hexdigest() == tr["logs_sha256"], (tr["seed"], tr["nUE"])
    assert split_stdout_blocks(p.stdout) == exp_blocks and len(exp_blocks) == 6


def _random_cases(n, seed):
    """Seeded random corners of the parameter space the CLI accepts (Beta.c:463-516 / WithNOMA:433-545):
    tiny and odd preamble counts, accessTime != 5 (the hard-coded 5 of Beta.c:112,389 then matters),
    grant budgets of 0 (nGrantUL=1) to plenty, immediate retransmission limits, both arrival laws."""
    rs = np.random.RandomState(seed)
    out = []
    for k in range(n):
        nUE = int(rs.choice([1, 2, 63, 64, 65, 130, 700, 1500, 2600, 4100, 6000]))
        kw = dict(nPreamble=int(rs.choice([1, 2, 3, 7, 16, 54, 64, 97, 200])),
                  backoff=int(rs.choice([1, 2, 5, 20, 33, 60])),
                  nGrantUL=int(rs.choice([1, 2, 3, 6, 12, 54, 300])),
                  maxRarWindow=int(rs.choice([1, 2, 3, 6, 11])),
                  maxMsg2TxCount=int(rs.choice([0, 1, 2, 9, 30])),
                  accessTime=int(rs.choice([1, 2, 5, 6, 10, 16])),
                  uniform=int(rs.rand() < 0.3))
        if kw["uniform"]:
            nUE = min(nUE, 2600)  # 60 000 subframes: keep the oracle in seconds
        out.append((int(rs.randint(0, 2)), nUE, kw, int(rs.randint(0, 2)), int(rs.randint(0, 1 << 30))))
    return out


@pytest.mark.parametrize("G", [0, 3])
def test_random_parameter_sweep(pkg, ob, engine, G):
    """96 random configurations in ONE call (concurrent trials, mixed RNG modes / variants / sizes), every
    counter and every logged field of every UE against the oracle; once with the engine's own
    cluster size and once forced to 3 workgroups per trial (uneven ownership)."""
    cases = _random_cases(96, 20240 + G)
    engine.set("cluster", G)
    try:
        cfgs = [pkg.make_cfg(n, variant=v, rng_mode=r, seed=s, **kw) for (v, n, kw, r, s) in cases]
        res, logs = engine.run_trials(cfgs, want_logs=True)
    finally:
        engine.set("cluster", 0)
    for (v, n, kw, r, s), rb, lb in zip(cases, res, logs):
        ores, oues = ob.run_trial(ob.make_cfg(n, variant=v, **kw), ob.Rng(r, s))
        assert_same(pkg, rb, lb, ores, oues, (v, n, kw, r, s, G))


def test_gpu_reproduces_reference_random_flags(pkg, engine):
    """The product path against the REAL reference under 63 random flag sets (tests/golden/ref_fuzz.json, made by
    tests/golden/fuzz_reference.py): every finished sweep point's Results.txt bytes, printed block and per-UE
    Logs.txt SHA-256, the rand() stream carried across the sweep through cfg.stream_offset — no oracle involved."""
    import json
    fz = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fuzz.json")))
    checked = 0
    for r in fz["runs"]:
        off = 0
        for k, tr in enumerate(r["trials"]):
            cfg = pkg.make_cfg(tr["nUE"], variant=1, rng_mode=pkg.RNG_GLIBC, seed=0, stream_offset=off, **r["cfg_overrides"])
            (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
            what = (r["argv"], tr["nUE"])
            assert res.status == 0, what
            off += res.draws
            assert pkg.format_results(cfg, res, 0.0).decode() == tr["results_text"], what
            so = pkg.format_stdout(cfg, res, 0.0).decode()
            so = "".join(l + "\n" for l in so.split("\n") if l and not l.startswith("Latency:"))
            assert so == r["stdout_blocks"][k], what
            text = pkg.format_logs(logs, tr["nUE"])
            assert len(text) == tr["logs_bytes"], what
            assert hashlib.sha256(text).hexdigest() == tr["logs_sha256"], what
            checked += 1
    assert checked >= 170


def test_gpu_sector_grants_variant(pkg, ob, engine):
    """PRACH_FLAG_SECTOR_GRANTS — the per-sector UL-grant path the reference's author left commented out
    (RandomAccessWithNOMA.c:260,271-273,312,626-637).  (1) glibc mode against the PATCHED reference's own files
    (tests/golden/ref_fuzz_sector.json: the reference compiled with exactly those lines swapped in, oracle/Makefile
    SED_SECTOR_GRANTS): Results.txt, printed block and Logs.txt SHA-256 of every finished sweep point, byte for byte;
    (2) both RNG modes against the oracle, every logged field of every UE."""
    import json
    fz = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fuzz_sector.json")))
    checked = 0
    for r in fz["runs"]:
        off = 0
        for k, tr in enumerate(r["trials"]):
            cfg = pkg.make_cfg(tr["nUE"], variant=1, rng_mode=pkg.RNG_GLIBC, seed=0, stream_offset=off, flags=pkg.FLAG_SECTOR_GRANTS, **r["cfg_overrides"])
            (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
            what = (r["argv"], tr["nUE"])
            assert res.status == 0, what
            off += res.draws
            assert pkg.format_results(cfg, res, 0.0).decode() == tr["results_text"], what
            so = pkg.format_stdout(cfg, res, 0.0).decode()
            so = "".join(l + "\n" for l in so.split("\n") if l and not l.startswith("Latency:"))
            assert so == r["stdout_blocks"][k], what
            text = pkg.format_logs(logs, tr["nUE"])
            assert len(text) == tr["logs_bytes"] and hashlib.sha256(text).hexdigest() == tr["logs_sha256"], what
            checked += 1
    assert checked >= 20
    for rng_mode, nUE, kw in ((1, 30000, {}), (0, 12000, dict(nGrantUL=3)), (1, 9000, dict(nGrantUL=2, nPreamble=16, backoff=5)),
                              (1, 5000, dict(uniform=1, nGrantUL=2)), (1, 65, {})):
        cfg = pkg.make_cfg(nUE, variant=1, rng_mode=rng_mode, seed=77, flags=pkg.FLAG_SECTOR_GRANTS, **kw)
        (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
        ores, oues = ob.run_trial(ob.make_cfg(nUE, variant=1, sector_grants=1, **kw), ob.Rng(rng_mode, 77))
        assert_same(pkg, res, logs, ores, oues, ("sector", rng_mode, nUE, kw))
    with pytest.raises(pkg.PrachError):  # sectors exist in RandomAccessWithNOMA.c only
        engine.run_trials([pkg.make_cfg(1000, variant=0, flags=pkg.FLAG_SECTOR_GRANTS)])


def test_gpu_sector_grants_on_the_batch_kernel(pkg, ob, engine):
    """the per-sector UL-grant budgets (WithNOMA:626-637) in Philox mode run on prach::batch_kernel (six budgets in its
    grant phase, a caller's sector recomputed from the UE's own first activation draw), not on the index-ordered trial_kernel; a call may mix
    such trials with ordinary ones; and the same trial is >= 5 x faster than on trial_kernel (engine option "legacy")."""
    S = pkg.FLAG_SECTOR_GRANTS
    cases = [(100000, S, dict(nGrantUL=12)), (40000, S, {}), (20000, 0, {}), (30000, S, dict(nGrantUL=3, backoff=7)), (8000, S, dict(nPreamble=64, nGrantUL=2)),
             (5000, 0, dict(nGrantUL=2)), (700, S, dict(nPreamble=5, nGrantUL=2, maxRarWindow=2)), (64, S, {})]
    cfgs = [pkg.make_cfg(n, variant=1, rng_mode=pkg.RNG_PHILOX, seed=300 + j, flags=f, **kw) for j, (n, f, kw) in enumerate(cases)]
    res, logs = engine.run_trials(cfgs, want_logs=True)
    tm = engine.timing()
    assert tm.trial_kernel_reruns == 0  # (an ordinary trial of the call may leave its cluster for the batch kernel: fallback_trials)
    for j, (n, f, kw) in enumerate(cases):
        ores, oues = ob.run_trial(ob.make_cfg(n, variant=1, sector_grants=1 if f else 0, **kw), ob.Rng(ob.RNG_PHILOX, 300 + j))
        assert_same(pkg, res[j], logs[j], ores, oues, ("sector on batch", n, kw))
    # the budgets bind: the 12-grant 100 000-UE trial ends differently with and without the flag
    (plain,), _ = engine.run_trials([pkg.make_cfg(100000, variant=1, rng_mode=pkg.RNG_PHILOX, seed=300, nGrantUL=12)])
    assert plain.nSuccessUE != res[0].nSuccessUE
    one = [cfgs[0]]
    engine.run_trials(one)
    fast = engine.timing()
    assert fast.rec_mode == 4
    engine.set("legacy", 1)
    try:
        (slow_r,), _ = engine.run_trials(one)
        slow = engine.timing()
    finally:
        engine.set("legacy", 0)
    assert (slow_r.nSuccessUE, slow_r.sumTimer, slow_r.draws) == (res[0].nSuccessUE, res[0].sumTimer, res[0].draws)
    print(f"sector grants, nUE = 100 000, 12 grants per sector: batch_kernel {fast.kernel_ms:.1f} ms, trial_kernel {slow.kernel_ms:.1f} ms")
    assert slow.kernel_ms > 5 * fast.kernel_ms


def test_dense_pass_option_agrees(pkg, ob, engine):
    """engine option "dense": the cluster kernel without the compacted two-phase pass (every group through the full
    per-UE body) gives the same trial, bit for bit."""
    cases = [(0, 9000, {}), (1, 9000, {}), (1, 5000, dict(nPreamble=3, backoff=2, nGrantUL=2, maxRarWindow=2, maxMsg2TxCount=1))]
    engine.set("dense", 1)
    try:
        for v, n, kw in cases:
            cfg = pkg.make_cfg(n, variant=v, rng_mode=pkg.RNG_PHILOX, seed=21, **kw)
            (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
            ores, oues = ob.run_trial(ob.make_cfg(n, variant=v, **kw), ob.Rng(ob.RNG_PHILOX, 21))
            assert_same(pkg, res, logs, ores, oues, ("dense", v, n, kw))
    finally:
        engine.set("dense", 0)


def test_pipeline_option_agrees(pkg, ob, engine):
    """A cluster runs phase A of subframe t+1 while the exchange of subframe t is in flight (UEs that then receive a grant are
    taken out of their bucket again and queued); engine option "pipeline" 0 keeps the two strictly one after the other."""
    cases = [(0, 12000, {}), (1, 12000, {}), (0, 4000, dict(uniform=1, nGrantUL=12)),
             (1, 6000, dict(nPreamble=3, backoff=2, nGrantUL=30, maxRarWindow=3, maxMsg2TxCount=1))]
    for pipe in (0, 1):
        engine.set("pipeline", pipe)
        engine.set("cluster", 4)
        try:
            for v, n, kw in cases:
                cfg = pkg.make_cfg(n, variant=v, rng_mode=pkg.RNG_PHILOX, seed=17, **kw)
                (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
                ores, oues = ob.run_trial(ob.make_cfg(n, variant=v, **kw), ob.Rng(ob.RNG_PHILOX, 17))
                assert_same(pkg, res, logs, ores, oues, ("pipeline", pipe, v, n, kw))
        finally:
            engine.set("pipeline", 1)
            engine.set("cluster", 0)


def test_record_layouts_agree(pkg, ob, engine):
    """One workgroup per trial keeps 8 + 4 byte hot records (16-bit subframe numbers); the 16-byte form is used when a
    subframe number may not fit (huge backoff indicator) or on request (engine option "wide_records"): same trials."""
    cases = [(1, 6000, {}), (0, 6000, dict(uniform=1, nGrantUL=12)), (1, 3000, dict(backoff=60000, nGrantUL=3)),
             (0, 2500, dict(nPreamble=2, backoff=3, nGrantUL=3, maxRarWindow=2, maxMsg2TxCount=3, accessTime=6))]
    for wide in (0, 1):
        engine.set("cluster", 1)
        engine.set("wide_records", wide)
        try:
            for v, n, kw in cases:
                cfg = pkg.make_cfg(n, variant=v, rng_mode=pkg.RNG_PHILOX, seed=33, **kw)
                (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
                ores, oues = ob.run_trial(ob.make_cfg(n, variant=v, **kw), ob.Rng(ob.RNG_PHILOX, 33))
                assert_same(pkg, res, logs, ores, oues, ("wide", wide, v, n, kw))
        finally:
            engine.set("cluster", 0)
            engine.set("wide_records", 0)


def test_event_queue_overflow_is_exact(pkg):
    """The compacted pass queues the UEs that have an event (8192 per workgroup and subframe); a wavefront that finds the
    queue full does its events in place.  libprach_hip_tinyq.so is the same library built with a 128-entry queue
    (make TINYQ=1 lib; __graft_entry__.build), so ordinary trials overflow it all the time: 250 random configurations
    (tests/tools/gpu_fuzz.py: forced cluster sizes, both RNG modes, both record layouts by size) against the oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "5g-nr-randomaccess_amd", "libprach_hip_tinyq.so")
    assert os.path.exists(lib), "build it: make -C 5g-nr-randomaccess_amd/csrc TINYQ=1 lib"
    env = dict(os.environ, PRACH_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "gpu_fuzz.py"), "31", "250"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "done 250 cases 0 bad" in out.stdout, out.stdout[-2000:]


def test_lean_cluster_local_overflow_leaves_together(ob):
    """A capacity that ONE workgroup of a lean cluster exceeds in its own pass (the early-leaver candidate list) is published to the
    whole cluster through the header granule's overflow bit: every workgroup leaves at the same subframe with PRACH_ERR_INTERNAL and
    the engine reruns the trial exactly, at once — no peer spins into a time-out, nothing is reported as "not co-resident".
    libprach_hip_tinyq.so is built with an 8-entry candidate list, so an ordinary overloaded trial gets there."""
    import ctypes as C2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, os, time
sys.path.insert(0, %r)
import numpy as np
import __graft_entry__ as g
from oracle import binding as ob
m = g.load_package()
eng = m.Engine(0)
eng.set("lds_records", 1); eng.set("cluster", 8)
cfg = m.make_cfg(20000, variant=1, rng_mode=m.RNG_PHILOX, seed=4)
t0 = time.time()
(res,), (logs,) = eng.run_trials([cfg], want_logs=True)
wall = time.time() - t0
tm = eng.timing()
ores, oues = ob.run_trial(ob.make_cfg(20000, variant=1), ob.Rng(ob.RNG_PHILOX, 4))
a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16); b = np.frombuffer(oues, dtype=np.int32).reshape(-1, 16)
print("RESULT", res.status, tm.fallback_trials, tm.spin_timeouts, int((a != b).any()), res.nSuccessUE == ores.nSuccessUE, round(wall, 2))
""" % root
    lib = os.path.join(root, "5g-nr-randomaccess_amd", "libprach_hip_tinyq.so")
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PRACH_LIB=lib), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.split("\n") if l.startswith("RESULT")][0].split()
    status, fallback, timeouts, differs, same_succ, wall = int(line[1]), int(line[2]), int(line[3]), int(line[4]), line[5], float(line[6])
    assert (status, fallback, timeouts, differs, same_succ) == (0, 1, 0, 0, "True"), out.stdout + out.stderr[-1500:]
    assert wall < 5.0, f"the overflowing cluster took {wall} s: its workgroups waited for each other"


@pytest.mark.gpu
def test_branch_free_event_body_equals_the_branched_form_on_the_device(tmp_path):
    """tests/test_host_logic.py's comparison with both forms compiled FOR THE DEVICE: every thread of a kernel runs prach_ue_body.h's branched and branch-free
    state machine on the same random case (states, parameters, caller tables, draws — random also where none is needed) and compares field by field."""
    import shutil
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "gpu_flat_equiv")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "tools", "gpu_flat_equiv.hip"), "-o", exe])
    p = subprocess.run([exe, "4000000", "21"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and " 0 differences" in p.stdout, p.stdout[-2000:]


def test_library_without_bitop3_agrees():
    """ROCm 7.2 folded a tree of the batched kernel's mask algebra into a WRONG v_bitop3_b32 in an experimental kernel shape (LABNOTES, round 4).  The shipped
    shape is fenced (prach_ue_body.h: lopaque) and checked against the oracle elsewhere in this suite; here the same library built WITHOUT the instruction
    (libprach_hip_nobitop3.so, `make NOBITOP3=1 lib`) must return the same results: 100-trial sweeps of both programs on the batched kernel in both workgroup
    shapes — a difference means some kernel's results depend on how the compiler selected its boolean instructions."""
    from conftest import ROOT
    nb = os.path.join(ROOT, "5g-nr-randomaccess_amd", "libprach_hip_nobitop3.so")
    if not os.path.exists(nb):
        pytest.skip("libprach_hip_nobitop3.so not built")

    def digests(lib):
        out = []
        for variant in ("0", "1"):
            for waves in ("8", "16"):
                env = dict(os.environ, PRACH_ENG_OPTS=f"batch_waves={waves}")
                if lib:
                    env["PRACH_LIB"] = lib
                else:
                    env.pop("PRACH_LIB", None)
                p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_batch.py"), "10", variant, "0"], env=env, capture_output=True, text=True, timeout=300)
                assert p.returncode == 0 and "bad=0" in p.stdout, p.stdout[-500:] + p.stderr[-500:]
                out.append(p.stdout.split("digest=")[1].split()[0])
        return out

    a, b = digests(None), digests(nb)
    assert a == b and a[0] == a[1] and a[2] == a[3], (a, b)
