"""Host side of the C ABI (no GPU): defaults, validation, arrival schedule, text surfaces, file
names, CLI argument behaviour — against the oracle and the reference's golden outputs."""
import ctypes as C
import os
import subprocess

import pytest

from conftest import load_golden, split_stdout_blocks


def test_defaults_match_reference_programs(pkg):
    b = pkg.make_cfg(100000, variant=pkg.VARIANT_BETA_C)
    assert (b.nPreamble, b.backoff, b.nGrantUL, b.maxRarWindow, b.maxMsg2TxCount, b.accessTime) == (54, 20, 54, 6, 9, 5)  # Beta.c:47-54
    w = pkg.make_cfg(100000, variant=pkg.VARIANT_WITHNOMA_C)
    assert (w.nPreamble, w.backoff, w.nGrantUL, w.maxRarWindow, w.maxMsg2TxCount, w.accessTime) == (54, 20, 12, 6, 9, 5)  # WithNOMA:71-78
    assert abs(w.cellRadius - 400.0) < 1e-6 and abs(w.hBS - 10.0) < 1e-6 and abs(w.hUT - 1.8) < 1e-6


def test_validate(pkg):
    ok = pkg.make_cfg(1000)
    assert pkg.lib().prach_cfg_validate(C.byref(ok)) == 0
    for kw, code in ((dict(nUE=0), -1), (dict(nPreamble=0), -1), (dict(backoff=0), -1), (dict(accessTime=0), -1),
                     (dict(nGrantUL=0), -1), (dict(rng_mode=7), -1), (dict(variant=9), -1),
                     (dict(nPreamble=255), -2), (dict(maxRarWindow=256), -2), (dict(maxMsg2TxCount=256), -2)):
        c = pkg.make_cfg(1000)
        for k, v in kw.items():
            setattr(c, k, v)
        assert pkg.lib().prach_cfg_validate(C.byref(c)) == code, kw
    assert pkg.lib().prach_cfg_validate(None) == -1


@pytest.mark.parametrize("nUE,uniform,aT", [(5000, 1, 5), (10000, 0, 5), (100000, 0, 5), (100000, 1, 5), (30000, 0, 10),
                                            (777, 0, 7), (1, 0, 5), (64, 1, 6)])
def test_arrival_schedule_matches_oracle(pkg, ob, nUE, uniform, aT):
    pc = pkg.make_cfg(nUE, uniform=uniform, accessTime=aT)
    oc = ob.make_cfg(nUE, uniform=uniform, accessTime=aT)
    ps, pna = pkg.arrival_schedule(pc)
    os_, ona = ob.arrival_schedule(oc)
    assert ps == os_ and pna == ona
    assert ps[-1] == nUE or uniform  # Beta(3,4) with the 0.0165 normaliser over-delivers: saturates early
    if not uniform and nUE == 100000 and aT == 5:
        assert ps.index(100000) * 5 == 7970  # activeCheck saturates at t=7970


def _to_presult(pkg, ores):
    r = pkg.PrachResult()
    for n, _ in pkg.PrachResult._fields_:
        if hasattr(ores, n):
            setattr(r, n, getattr(ores, n))
    return r


@pytest.mark.parametrize("case", ["beta", "noma_uniform", "noma_odd"])
def test_text_surfaces_byte_identical_to_reference(pkg, ob, case):
    """prach_format_results / _stdout / _logs on the oracle's numbers reproduce the reference's files."""
    import hashlib
    g = load_golden(case)
    variant = 0 if g["variant"] == "BETA_C" else 1
    blocks = split_stdout_blocks(g["stdout"])
    rng = ob.Rng(ob.RNG_GLIBC, 0)
    for k, tr in enumerate(g["trials"][:2]):
        ocfg = ob.make_cfg(tr["nUE"], variant=variant, **g["cfg_overrides"])
        ores, oues = ob.run_trial(ocfg, rng)
        pcfg = pkg.make_cfg(tr["nUE"], variant=variant, **g["cfg_overrides"])
        pres = _to_presult(pkg, ores)
        txt = pkg.format_results(pcfg, pres, 1.5).decode()
        if variant == 0:
            assert txt == tr["results_text"] + "1.500000"  # Beta.c:481: latency, no newline
        else:
            assert txt == tr["results_text"]
        so = pkg.format_stdout(pcfg, pres, 1.5).decode()
        so = "".join(l + "\n" for l in so.split("\n") if l and not l.startswith("Latency:"))
        assert so == blocks[k]
        logs = C.cast(oues, C.POINTER(pkg.PrachUeLog))
        assert hashlib.sha256(pkg.format_logs(logs, tr["nUE"])).hexdigest() == tr["logs_sha256"]
        name = C.create_string_buffer(256)
        pkg.lib().prach_result_file_name(C.byref(pcfg), 0, name, 256)
        assert name.value.decode() == f"{tr['dir']}/{tr['results_file']}"
        pkg.lib().prach_result_file_name(C.byref(pcfg), 1, name, 256)
        assert name.value.decode() == f"{tr['dir']}/{tr['logs_file']}"


def test_write_trial_files(pkg, ob, tmp_path):
    ocfg = ob.make_cfg(1200, variant=1)
    ores, oues = ob.run_trial(ocfg, ob.Rng(ob.RNG_GLIBC, 0))
    pcfg = pkg.make_cfg(1200, variant=1)
    rc = pkg.lib().prach_write_trial_files(C.byref(pcfg), C.byref(_to_presult(pkg, ores)),
                                           C.cast(oues, C.POINTER(pkg.PrachUeLog)), 0.0, str(tmp_path).encode())
    assert rc == 0
    d = tmp_path / "NomaBetaResults"
    assert (d / "0_54_1200_Results.txt").read_bytes() == ob.format_results(ocfg, ores)
    assert (d / "0_54_UE01200_Logs.txt").read_bytes() == ob.format_logs(oues, 1200)


CLI_ERRORS = [  # argv, message (WithNOMA.c:94-157): printed to stdout, exit(-1) == 255
    (["-t", "0"], "Simulation count must be greater than zero."),
    (["-d", "3"], "Traffic model just choose 1 or 2"),
    (["-p", "0"], "Number of preamble must be greater than zero."),
    (["-b", "0"], "Backoff indicator must be greater than zero."),
    (["-g", "0"], "The number of Up Link Grant per RAR must be greater than zero."),
    (["-rc", "0"], "The maximum RAR window size must be greater than zero."),
    (["-mrc", "0"], "Maximum retransmissions must be greater than zero."),
    (["-s", "4"], "The size of the subframe must be at least 5."),
    (["-c", "399"], "The radius of the cell is entered in diameter units and must be greater than 400m."),
    (["-bs", "9"], "The height of the BS must be between 10m and 20m."),
    (["-ut", "23"], "The height of the UE must be between 1.5m and 22.5m."),
]


@pytest.mark.parametrize("argv,msg", CLI_ERRORS)
def test_cli_error_behaviour(pkg, argv, msg):
    p = subprocess.run([pkg.CLI_PATH] + argv, capture_output=True, text=True)
    assert p.returncode == 255 and p.stdout == msg


def test_cli_usage_on_unknown_flag(pkg):
    p = subprocess.run([pkg.CLI_PATH, "--bogus", "1"], capture_output=True, text=True)
    assert p.returncode == 255 and p.stdout.startswith("--times         -t : Simulation times (int)\n")
    assert "--hut           -u : Height of UE from ground (float)" in p.stdout
    p = subprocess.run([pkg.CLI_PATH, "-t"], capture_output=True, text=True)  # the reference segfaults here
    assert p.returncode == 255


def test_results_csv_matches_reference_script(pkg):
    """PINNED by the reference's own post-processor: tests/golden/results_csv.json holds 1000 Results.txt texts (oracle,
    Beta.c program, 12 grants, seeds 0..99 x nUE 10k..100k) and the bytes of the results.csv that
    /root/reference/AveragePerformance.py — executed unmodified, by path, in a scratch directory
    (tests/golden/make_results_csv.py) — wrote from them.  prach_results_csv_accumulate / prach_results_csv_row
    (AveragePerformance.py:10-24 in host C) must reproduce those bytes."""
    import base64
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "results_csv.json")))
    want = base64.b64decode(fx["results_csv_b64"])
    rows = [[fx["results_txt"][f"{s}_{n}"] for s in fx["seeds"]] for n in fx["points"]]
    assert len(rows) == 10 and all(len(r) == 100 for r in rows)
    got = pkg.results_csv(rows)
    assert got == want
    assert want.count(b"\r\n") == 10 and want.startswith(b"10000.0,100.0,10000.0,")
    # (the published file of the reference's own 100-seed run, results.csv:10, reads 100000.0,18.99,18989.29,5.76,96.001,...)
    last = want.split(b"\r\n")[9].split(b",")
    assert abs(float(last[1]) - 18.99) < 0.05 and abs(float(last[4]) - 96.0) < 0.1


def test_results_csv_matches_numpy_csv_writer(pkg, tmp_path):
    """results.csv rows == what AveragePerformance.py's arithmetic and csv.writer produce
    (sequential double sum in seed order, /nseeds, np.around(.,3), float repr, CRLF)."""
    import csv
    import io
    import random

    import numpy as np
    rnd = random.Random(7)
    rows = []
    for nue in range(10000, 100001, 10000):
        texts = []
        for s in range(100):
            succ = rnd.randint(0, nue)
            texts.append("%d\n%.2lf\n%d\n%.2lf\n%.2lf\n%lf" % (nue, 100.0 * succ / nue, succ, rnd.uniform(1, 8), rnd.uniform(15, 120),
                                                               rnd.uniform(0.1, 1600)))
        rows.append(texts)
    mine = pkg.results_csv(rows)
    avg = [[0, 0, 0, 0, 0, 0] for _ in rows]
    for n, texts in enumerate(rows):
        for t in texts:
            data = [float(l.strip()) for l in t.split("\n")]
            for i in range(6):
                avg[n][i] += data[i]
    buf = io.StringIO(newline="")
    w = csv.writer(buf)
    for lists in np.array(avg):
        w.writerow(np.around(lists / 100, 3))
    assert mine == buf.getvalue().encode()
    # the shape of the reference's own file (results.csv:1): trailing zeros dropped, ".0" kept
    ref_like = pkg.results_csv([["10000\n100.00\n10000\n2.59\n47.11\n2.242"] * 4])
    assert ref_like == b"10000.0,100.0,10000.0,2.59,47.11,2.242\r\n"


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` on a node with fewer than N GPUs exits non-zero with a message before anything runs
    (it must never degrade to a one-GPU line); a WORLD_SIZE that contradicts --gpus is refused as well."""
    import subprocess, sys, os
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "visible GPU" in p.stderr and not any(l.startswith("{") for l in p.stdout.splitlines())
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600,
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr


def test_noma_cell_radius_must_exceed_the_redraw_threshold(pkg):
    """NOMA.c:167-172 redraws a UE's distance until it exceeds 35 m: a smaller cell never terminates in the reference; the library refuses it."""
    ok = pkg.make_cfg(100, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, cellRadius=36.0)
    bad = pkg.make_cfg(100, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, cellRadius=35.0)
    assert pkg.lib().prach_cfg_validate(ok) == pkg.OK and pkg.lib().prach_cfg_validate(bad) == -1  # PRACH_ERR_ARG
    beta = pkg.make_cfg(100, cellRadius=1.0)  # (the other programs parse the flag and never read it: WithNOMA:80-82)
    assert pkg.lib().prach_cfg_validate(beta) == pkg.OK


def test_branch_free_event_body_equals_the_branched_form(tmp_path):
    """prach_batch.hip's event body runs prach_ue_body.h's state machine in a branch-free form (flat_catch_up / flat_plan / flat_select / flat_schedule);
    every other kernel, and the oracle-checked history of this one, runs the branched form (pw_catch_up / ue_plan / ue_select / pw_schedule).  Both are
    host-callable: tests/tools/flat_equiv.hip runs them side by side on the CPU on random UE states, parameters, caller tables and draws (random also where the
    UE needs none: an unneeded draw may reach no output) and stops at the first difference.  No GPU involved; hipcc only compiles the host program."""
    import shutil
    import subprocess
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "flat_equiv")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "tools", "flat_equiv.hip"), "-o", exe])
    for seed in (11, 12):
        p = subprocess.run([exe, "2000000", str(seed)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and " 0 differences" in p.stdout, p.stdout[-2000:]
