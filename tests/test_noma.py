"""NOMA.c variant (sector power-level grouping, SURVEY §8 a13 / BASELINE config 4).
CPU: the oracle against the real NOMA program's output (glibc stream, chained sweep) and the host-side
activation table against the oracle.  GPU (-m gpu): the HIP kernel against the oracle in Philox mode."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden


def golden_lines():
    g = load_golden("noma_c")
    seeds, cur = [], []
    for l in g["stdout"].split("\n"):
        if l == "Done":
            seeds.append(cur)
            cur = []
        elif l:
            cur.append(l)
    return g, seeds


def _pool_map(fn, items):
    """The oracle is plain C behind ctypes (the GIL is released during a call, no global state): independent chains of
    trials run on all host cores, so that EVERY reference line is checked in the default CPU suite."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as ex:
        return list(ex.map(fn, items))


def test_noma_oracle_reproduces_reference_stdout(ob):
    """Pinned: ALL 100 'nUE nSucc succ% avgTx avgDelay' lines NOMA.c prints for its 10 seeds x 10 sweep points
    (NOMA.c:606-632), the glibc stream chained over the sweep of a seed (NOMA.c:644-647 seeds once per seed)."""
    g, seeds = golden_lines()
    assert len(seeds) == 10 and all(len(s) == 10 for s in seeds)
    assert seeds[0] == seeds[1]  # glibc: srand(0) == srand(1)

    def chain(seed):
        rng = ob.Rng(ob.RNG_GLIBC, seed)
        bad = []
        for k, n in enumerate(range(10000, 100001, 10000)):
            cfg = ob.make_noma_cfg(n)
            res, _ = ob.noma_run_trial(cfg, rng, want_ues=False)
            if ob.noma_format_line(cfg, res).decode().strip() != seeds[seed][k]:
                bad.append((seed, n))
        return bad

    assert sum(_pool_map(chain, range(10)), []) == []
    # the per-nUE files are the same lines appended seed after seed (mode "aw+", NOMA.c:605)
    assert g["files"]["Sector_10000_Result.txt"].split("\n")[0] == seeds[0][0]
    assert g["files"]["Sector_100000_Result.txt"].split("\n")[:10] == [s[9] for s in seeds]


def test_noma_activation_table_matches_oracle(pkg, ob):
    """prach_noma_activation_table (host C, same libm) == what the oracle's activeUE computes, bit for bit."""
    n, seed = 6000, 3
    cfg = pkg.make_cfg(n, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=seed)
    assert (cfg.nGrantUL, cfg.maxRarWindow, cfg.maxMsg2TxCount, cfg.accessTime, cfg.nPreamble, cfg.backoff) == (2, 5, 10, 5, 54, 20)  # NOMA.c:41-47
    pre0, sec, gain, lgain, nd = pkg.noma_activation_table(cfg)
    res, ues = ob.noma_run_trial(ob.make_noma_cfg(n), ob.Rng(ob.RNG_PHILOX, seed))
    a = np.frombuffer(ues, dtype=np.dtype([("i", np.int32, 16), ("g", np.float64)]))
    act = a["i"][:, 8] >= 0  # activated UEs have a sector
    assert act.sum() == res.activeCheck == n
    assert (a["i"][:, 8] == sec).all()
    assert (a["g"] == gain).all()
    assert (np.log(gain) == lgain).all() and (gain >= 1e-7).all() and (nd >= 4).all()
    assert sec.min() >= 0 and sec.max() <= 5 and pre0.min() >= 0 and pre0.max() < 54


NOMA_GPU_CASES = [(3000, 0, {}), (10000, 1, {}), (30000, 2, {}), (64, 3, {}), (1, 4, {}), (5000, 5, dict(nPreamble=8, backoff=3)),
                  (8000, 6, dict(nGrantUL=1, maxMsg2TxCount=3)), (8000, 7, dict(nPreamble=64, nGrantUL=5, backoff=40)),
                  # maxRarWindow > 5: NOMA.c:453-455 sets rarWindow to the literal 5 before the test, so nobody is rescheduled
                  (8000, 8, dict(maxRarWindow=6)), (6000, 9, dict(maxRarWindow=7, accessTime=3, nGrantUL=3))]


@pytest.mark.gpu
@pytest.mark.parametrize("nUE,seed,kw", NOMA_GPU_CASES)
def test_gpu_noma_equals_oracle(pkg, ob, engine, nUE, seed, kw):
    cfg = pkg.make_cfg(nUE, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=seed, **kw)
    (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
    okw = dict(kw)
    if "maxMsg2TxCount" in okw:
        okw["maxMsg1ReTx"] = okw.pop("maxMsg2TxCount")
    ocfg = ob.make_noma_cfg(nUE, **okw)
    ores, oues = ob.noma_run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, seed))
    assert res.status == 0
    assert (res.nSuccessUE, res.sumTimer, res.preambleTxCount, res.failCounts, res.activeCheck, res.draws) == \
           (ores.nSuccessUE, ores.delay, ores.nTxP, ores.raFailedUEs, ores.activeCheck, ores.draws)
    a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
    b = np.frombuffer(oues, dtype=np.dtype([("i", np.int32, 16), ("g", np.float64)]))["i"]
    diff = np.where((a != b).any(axis=1))[0]
    assert diff.size == 0, (diff[:5], a[diff[:3]], b[diff[:3]])
    if res.nSuccessUE:
        assert pkg.format_noma_line(cfg, res) == ob.noma_format_line(ocfg, ores)


@pytest.mark.gpu
def test_gpu_noma_full_size_and_statistics(pkg, ob, engine):
    """BASELINE config 4: nUE=100 000; aggregates AND every logged field of every UE bit-exact vs the oracle, and
    statistically consistent with the reference's glibc-stream run (26 884 successes at seed 0,
    tests/golden/noma_c.json) — different RNG, same process."""
    cfg = pkg.make_cfg(100000, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=0)
    (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
    ocfg = ob.make_noma_cfg(100000)
    ores, oues = ob.noma_run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, 0), want_ues=True)
    assert (res.status, res.nSuccessUE, res.sumTimer, res.preambleTxCount, res.failCounts, res.activeCheck, res.draws, res.time_exit) == \
           (0, ores.nSuccessUE, ores.delay, ores.nTxP, ores.raFailedUEs, ores.activeCheck, ores.draws, ores.time_exit)
    a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
    b = np.frombuffer(oues, dtype=np.dtype([("i", np.int32, 16), ("g", np.float64)]))["i"]
    diff = np.where((a != b).any(axis=1))[0]
    assert diff.size == 0, (diff[:5], a[diff[:3]], b[diff[:3]])
    assert pkg.format_noma_line(cfg, res) == ob.noma_format_line(ocfg, ores)
    assert abs(res.nSuccessUE - 26884) < 600


@pytest.mark.gpu
def test_gpu_noma_activation_table_device_vs_host(pkg, engine):
    """activeUE on the device (noma_activation_kernel, NOMA.c:131-192) against the host form built with the reference's libm:
    preamble, sector and draw count of EVERY UE identical; the gain of a UE the kernel did not flag within 32 ulp of the host's (the
    resolver's order band ACT_GAIN_ORDER_BAND = 4e-14 relative = 180 ulp covers two such gains off in opposite directions: prach_noma_act.h), of a flagged UE the host's bits (recomputed); few UEs flagged (the expected rate is
    3 x 128 / 2^29 = 7e-7: float rounding boundaries of x, y and the path loss)."""
    nflag = ntot = 0
    worst = 0
    for nUE, seed, kw in ((1000000, 0, {}), (1000000, 123456789, {}), (300000, 7, dict(cellRadius=60.0, nPreamble=7)), (1000, 1, dict(cellRadius=1500.0))):
        cfg = pkg.make_cfg(nUE, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=seed, **kw)
        hp, hs, hg, hl, hn = pkg.noma_activation_table(cfg)
        dp, ds, dg, dl, dn, fl = pkg.noma_activation_table_device(engine, cfg)
        assert (hp == dp).all() and (hs == ds).all() and (hn == dn).all(), (nUE, seed)
        f = fl != 0
        assert (hg[f] == dg[f]).all() and (hl[f] == dl[f]).all()
        ulp = np.abs(hg.view(np.int64) - dg.view(np.int64))  # (same sign, finite: the distance in representable doubles)
        assert np.isfinite(hg).all() and (hg >= 1e-7).all()
        worst = max(worst, int(ulp[~f].max()))
        assert np.abs(hl - dl)[~f].max() < 1e-13
        nflag += int(f.sum()); ntot += nUE
    assert worst <= 32, worst
    assert nflag <= 40, (nflag, ntot)
    print(f"device activation table: {ntot} UEs, {nflag} recomputed on the host, largest gain difference of the others {worst} ulp")


@pytest.mark.gpu
def test_gpu_noma_host_activation_option_and_ambiguity_rerun(pkg, ob, engine, capfd):
    """Same results whichever side builds the table; and the resolver's 'a gain comparison fell inside the error band' exit (forced by
    the test hook for every sort) reruns the trial with the host-built table, visibly, and still matches the oracle."""
    cfgs = [pkg.make_cfg(n, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=s) for n, s in ((20000, 3), (3000, 4))]
    res_d, logs_d = engine.run_trials(cfgs, want_logs=True)
    assert engine.timing().fallback_trials == 0
    engine.set("noma_host_activation", 1)
    try:
        res_h, logs_h = engine.run_trials(cfgs, want_logs=True)
        assert engine.timing().noma_host_ues == 0
    finally:
        engine.set("noma_host_activation", 0)
    engine.set("noma_ambiguity_test", 1)
    try:
        capfd.readouterr()
        res_a, logs_a = engine.run_trials(cfgs, want_logs=True)
        tm = engine.timing()
        assert tm.fallback_trials >= 1 and "host-built activation table" in capfd.readouterr().err
    finally:
        engine.set("noma_ambiguity_test", 0)
    for k, cfg in enumerate(cfgs):
        ores, oues = ob.noma_run_trial(ob.make_noma_cfg(cfg.nUE), ob.Rng(ob.RNG_PHILOX, cfg.seed))
        b = np.frombuffer(oues, dtype=np.dtype([("i", np.int32, 16), ("g", np.float64)]))["i"]
        for res, logs in ((res_d[k], logs_d[k]), (res_h[k], logs_h[k]), (res_a[k], logs_a[k])):
            assert (res.status, res.nSuccessUE, res.sumTimer, res.preambleTxCount, res.draws) == (0, ores.nSuccessUE, ores.delay, ores.nTxP, ores.draws)
            assert (np.frombuffer(logs, dtype=np.int32).reshape(-1, 16) == b).all()


@pytest.mark.gpu
@pytest.mark.parametrize("G", [1, 4, 32])
def test_gpu_noma_cluster_sizes(pkg, ob, engine, G):
    """The NOMA kernel with G workgroups per trial (one granule exchange per 5 ms slot) == the oracle for every G;
    also an early-finishing light-load trial (every UE succeeds: exit subframe of NOMA.c:707-710)."""
    engine.set("cluster", G)
    try:
        for nUE, kw in ((20000, {}), (64, dict(nGrantUL=30)), (8, dict(nGrantUL=30))):
            cfg = pkg.make_cfg(nUE, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=11, **kw)
            (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
            ocfg = ob.make_noma_cfg(nUE, **kw)
            ores, oues = ob.noma_run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, 11))
            assert (res.nSuccessUE, res.sumTimer, res.preambleTxCount, res.failCounts, res.draws, res.time_exit) == \
                   (ores.nSuccessUE, ores.delay, ores.nTxP, ores.raFailedUEs, ores.draws, ores.time_exit), (G, nUE)
            a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
            b = np.frombuffer(oues, dtype=np.dtype([("i", np.int32, 16), ("g", np.float64)]))["i"]
            assert (a == b).all(), (G, nUE)
    finally:
        engine.set("cluster", 0)


@pytest.mark.gpu
def test_gpu_noma_random_parameter_sweep(pkg, ob, engine):
    """48 seeded random configurations of NOMA.c's constants (NOMA.c:41-57) in ONE call (concurrent trials):
    aggregates and every logged field of every UE against the oracle."""
    rs = np.random.RandomState(77)
    cases = []
    for _ in range(48):
        kw = dict(nPreamble=int(rs.choice([1, 2, 3, 8, 33, 54, 64])), backoff=int(rs.randint(1, 60)), nGrantUL=int(rs.choice([1, 2, 3, 5, 12, 40])),
                  maxRarWindow=int(rs.randint(1, 10)), maxMsg2TxCount=int(rs.choice([0, 1, 3, 10, 25])), accessTime=int(rs.choice([1, 2, 3, 5, 5, 8, 10])))
        if rs.rand() < 0.3:
            kw["max_steps"] = int(rs.randint(1, 6000))
        if rs.rand() < 0.3:
            kw["cellRadius"] = float(rs.choice([50.0, 250.0, 2000.0]))
        cases.append((int(rs.choice([1, 7, 64, 65, 500, 2000, 6000, 12000])), int(rs.randint(0, 1 << 30)), kw))
    cfgs = [pkg.make_cfg(n, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=s, **kw) for n, s, kw in cases]
    res, logs = engine.run_trials(cfgs, want_logs=True)
    for (n, s, kw), r, lg in zip(cases, res, logs):
        okw = dict(kw)
        okw["maxMsg1ReTx"] = okw.pop("maxMsg2TxCount")
        ores, oues = ob.noma_run_trial(ob.make_noma_cfg(n, **okw), ob.Rng(ob.RNG_PHILOX, s))
        assert (r.status, r.nSuccessUE, r.sumTimer, r.preambleTxCount, r.failCounts, r.activeCheck, r.draws, r.time_exit) == \
               (0, ores.nSuccessUE, ores.delay, ores.nTxP, ores.raFailedUEs, ores.activeCheck, ores.draws, ores.time_exit), (n, s, kw)
        a = np.frombuffer(lg, dtype=np.int32).reshape(-1, 16)
        b = np.frombuffer(oues, dtype=np.dtype([("i", np.int32, 16), ("g", np.float64)]))["i"]
        assert (a == b).all(), (n, s, kw)


def test_noma_oracle_reproduces_reference_random_parameters(ob):
    """48 random settings of NOMA.c's file-scope parameters (NOMA.c:41-57; set by oracle/noma_params_main.c in front
    of the reference's own main, tests/golden/fuzz_reference_noma.py): ALL 192 lines the REAL program printed for the
    sweep points it finished, glibc stream chained over the sweep (the runs are independent: all host cores)."""
    import json
    import os
    fz = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fuzz_noma.json")))
    assert len(fz["runs"]) >= 45

    def run(r):
        rng = ob.Rng(ob.RNG_GLIBC, 0)
        bad, checked = [], 0
        for k, line in enumerate(r["lines"]):
            n = 10000 * (k + 1)
            assert int(line.split()[0]) == n
            cfg = ob.make_noma_cfg(n, **r["cfg_overrides"])
            res, _ = ob.noma_run_trial(cfg, rng, want_ues=False)
            if ob.noma_format_line(cfg, res).decode().strip() != line:
                bad.append((r["argv"], n))
            checked += 1
        return bad, checked

    out = _pool_map(run, fz["runs"])
    assert sum((b for b, _ in out), []) == []
    assert sum(c for _, c in out) == sum(len(r["lines"]) for r in fz["runs"]) >= 190


def test_noma_oracle_nonsector_reproduces_patched_reference(ob):
    """NOMA.c's cell-wide grouping preambleCollisionDetection (NOMA.c:325-447), whose call the author left commented
    out (NOMA.c:688).  Pin = the reference compiled with lines 688 / 689 swapped by the sed recipe of oracle/Makefile
    (SED_NONSECTOR_NOMA) and its file-scope parameters set by oracle/noma_params_main.c (tests/golden/fuzz_reference_noma.py,
    PRACH_FUZZ_VARIANT=nonsector): every line the patched program printed, glibc stream chained over the sweep."""
    import json
    import os
    fz = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fuzz_noma_nonsector.json")))
    assert fz["nonsector"] == 1 and len(fz["runs"]) >= 16

    def run(r):
        rng, rng6 = ob.Rng(ob.RNG_GLIBC, 0), ob.Rng(ob.RNG_GLIBC, 0)
        bad, checked, differs = [], 0, 0
        for k, line in enumerate(r["lines"]):
            n = 10000 * (k + 1)
            cfg = ob.make_noma_cfg(n, nonsector=1, **r["cfg_overrides"])
            res, _ = ob.noma_run_trial(cfg, rng, want_ues=False)
            if ob.noma_format_line(cfg, res).decode().strip() != line:
                bad.append((r["argv"], n))
            checked += 1
            if k == 0:  # the per-sector grouping gives another trial
                cfg6 = ob.make_noma_cfg(n, **r["cfg_overrides"])
                res6, _ = ob.noma_run_trial(cfg6, rng6, want_ues=False)
                differs += ob.noma_format_line(cfg6, res6).decode().strip() != line
        return bad, checked, differs

    out = _pool_map(run, fz["runs"])
    assert sum((b for b, _, _ in out), []) == []
    assert sum(c for _, c, _ in out) == sum(len(r["lines"]) for r in fz["runs"]) >= 40
    assert sum(d for _, _, d in out) >= len(fz["runs"]) // 2


@pytest.mark.gpu
def test_gpu_noma_nonsector_equals_oracle(pkg, ob, engine):
    """PRACH_FLAG_NOMA_NONSECTOR on the GPU (one resolver wavefront, one grant budget per access slot) == the oracle's
    cell-wide grouping, aggregates and every logged field of every UE, for several cluster sizes."""
    cases = [(20000, 1, {}), (100000, 2, {}), (6000, 3, dict(nGrantUL=5, nPreamble=64)), (9000, 4, dict(nGrantUL=1, backoff=7)),
             (64, 5, dict(nGrantUL=30)), (12000, 6, dict(nPreamble=8, nGrantUL=3, accessTime=3))]
    for G in (0, 1, 8):
        engine.set("cluster", G)
        try:
            for nUE, seed, kw in cases:
                cfg = pkg.make_cfg(nUE, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=seed, flags=pkg.FLAG_NOMA_NONSECTOR, **kw)
                (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
                ocfg = ob.make_noma_cfg(nUE, nonsector=1, **kw)
                ores, oues = ob.noma_run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, seed))
                assert (res.status, res.nSuccessUE, res.sumTimer, res.preambleTxCount, res.failCounts, res.activeCheck, res.draws, res.time_exit) == \
                       (0, ores.nSuccessUE, ores.delay, ores.nTxP, ores.raFailedUEs, ores.activeCheck, ores.draws, ores.time_exit), (G, nUE, kw)
                a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
                b = np.frombuffer(oues, dtype=np.dtype([("i", np.int32, 16), ("g", np.float64)]))["i"]
                assert (a == b).all(), (G, nUE, kw)
                o6, _ = ob.noma_run_trial(ob.make_noma_cfg(nUE, **kw), ob.Rng(ob.RNG_PHILOX, seed), want_ues=False)
                if nUE >= 6000:
                    assert (o6.nSuccessUE, o6.delay) != (ores.nSuccessUE, ores.delay)  # not the per-sector trial
        finally:
            engine.set("cluster", 0)
    with pytest.raises(pkg.PrachError):  # the flag belongs to NOMA_C
        engine.run_trials([pkg.make_cfg(1000, variant=pkg.VARIANT_BETA_C, flags=pkg.FLAG_NOMA_NONSECTOR)])


@pytest.mark.gpu
def test_gpu_noma_glibc_reproduces_reference_lines(pkg, ob, engine):
    """BASELINE config 4 against the REFERENCE'S OWN OUTPUT, no oracle in between: NOMA_C in glibc mode (the arrivals of every access slot
    activated on the host in stream order with the reference's libm, everything else on the device at the reference's stream positions:
    prach_noma_glibc.hip), the rand() stream chained over the ten-point sweep of a seed like NOMA.c:644-647 — the lines NOMA.c printed
    (tests/golden/noma_c.json), for seeds 0 and 7, up to nUE = 100 000."""
    g, seeds = golden_lines()
    for seed in (0, 7):
        off = 0
        for k, n in enumerate(range(10000, 100001, 10000)):
            cfg = pkg.make_cfg(n, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_GLIBC, seed=seed, stream_offset=off)
            (res,), _ = engine.run_trials([cfg])
            assert res.status == 0
            off += res.draws
            assert pkg.format_noma_line(cfg, res).decode().strip() == seeds[seed][k], (seed, n)


@pytest.mark.gpu
def test_gpu_noma_glibc_equals_oracle(pkg, ob, engine):
    """... and against the oracle in glibc mode, every logged field of every UE, on parameter sets the reference binary cannot take
    (NOMA.c has no command line), including the cell-wide grouping variant and an early-finishing trial."""
    cases = [(6000, 3, {}, 0), (20000, 1, dict(nGrantUL=5, nPreamble=64, backoff=7), 0), (9000, 4, dict(maxRarWindow=6), 0),
             (12000, 5, dict(nPreamble=8, nGrantUL=3, accessTime=3, maxMsg2TxCount=2), 0), (64, 6, dict(nGrantUL=30), 0),
             (15000, 8, {}, 1), (7000, 9, dict(nGrantUL=1, backoff=3, max_steps=3000), 1)]
    for nUE, seed, kw, nonsector in cases:
        cfg = pkg.make_cfg(nUE, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_GLIBC, seed=seed, flags=pkg.FLAG_NOMA_NONSECTOR if nonsector else 0, **kw)
        (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
        okw = dict(kw)
        if "maxMsg2TxCount" in okw:
            okw["maxMsg1ReTx"] = okw.pop("maxMsg2TxCount")
        ocfg = ob.make_noma_cfg(nUE, nonsector=nonsector, **okw)
        ores, oues = ob.noma_run_trial(ocfg, ob.Rng(ob.RNG_GLIBC, seed))
        assert (res.status, res.nSuccessUE, res.sumTimer, res.preambleTxCount, res.failCounts, res.activeCheck, res.draws, res.time_exit, res.steps) == \
               (0, ores.nSuccessUE, ores.delay, ores.nTxP, ores.raFailedUEs, ores.activeCheck, ores.draws, ores.time_exit, ores.steps), (nUE, seed, kw)
        a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
        b = np.frombuffer(oues, dtype=np.dtype([("i", np.int32, 16), ("g", np.float64)]))["i"]
        diff = np.where((a != b).any(axis=1))[0]
        assert diff.size == 0, (nUE, seed, kw, diff[:5], a[diff[:3]], b[diff[:3]])
        # the same trial slot by slot with the arrivals activated by the host (the form a trial falls back to when the single-launch kernel finds a
        # value inside the device libm's error band) — also reached through the test hook
        for key in ("noma_host_activation", "noma_ambiguity_test"):
            engine.set(key, 1)
            try:
                (res2,), (logs2,) = engine.run_trials([cfg], want_logs=True)
                assert engine.timing().fallback_trials == (1 if key == "noma_ambiguity_test" else 0)
            finally:
                engine.set(key, 0)
            assert (res2.nSuccessUE, res2.sumTimer, res2.draws, res2.time_exit, res2.steps) == (res.nSuccessUE, res.sumTimer, res.draws, res.time_exit, res.steps)
            assert bytes(logs2) == bytes(logs)


@pytest.mark.gpu
def test_gpu_noma_radius_just_above_the_exclusion_zone(pkg, ob, engine, capfd):
    """cellRadius = 35.01 is a valid configuration (NOMA.c:167-172 redraws the distance until it exceeds 35 m: ~1 750 draws per UE here): the device's redraw
    loop gives up after 4 096 iterations and flags such a UE for the host — every tenth one, more than the kernel's list holds.  The engine then builds the
    launch's activation table on the host (the path a NOMA_AMBIGUOUS rerun takes) instead of failing: results equal the oracle's, one line on stderr."""
    n = 100000
    cfg = pkg.make_cfg(n, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=5, cellRadius=35.01, max_steps=600)
    capfd.readouterr()
    (res,), (logs,) = engine.run_trials([cfg], want_logs=True)
    err = capfd.readouterr().err
    assert "the activation table of this launch is built on the host" in err
    ocfg = ob.make_noma_cfg(n, cellRadius=35.01, max_steps=600)
    ores, oues = ob.noma_run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, 5), want_ues=True)
    assert (res.status, res.nSuccessUE, res.sumTimer, res.preambleTxCount, res.failCounts, res.activeCheck, res.draws, res.time_exit) == \
           (0, ores.nSuccessUE, ores.delay, ores.nTxP, ores.raFailedUEs, ores.activeCheck, ores.draws, ores.time_exit)
    a = np.frombuffer(logs, dtype=np.int32).reshape(-1, 16)
    b = np.frombuffer(oues, dtype=np.dtype([("i", np.int32, 16), ("g", np.float64)]))["i"]
    assert (a == b).all()
