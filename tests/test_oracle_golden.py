"""PIN: the oracle (oracle/prach_oracle.c) against outputs of the REAL reference programs, compiled
from /root/reference and run in the build container (tests/golden/*.json).  Byte-identical
Results.txt, byte-identical stdout blocks, and SHA-256 of the per-UE Logs.txt for every trial of a
sweep, with the glibc rand() stream carried across the sweep exactly as the reference does
(one srand per seed, Beta.c:69-71).

The default run stops each chain early so the CPU suite stays at a few minutes
(PRACH_FULL_GOLDEN=1 runs every point; all 10 points of every case have been verified that way).
"""
import hashlib
import os

import pytest

from conftest import load_golden, split_stdout_blocks

FULL = os.environ.get("PRACH_FULL_GOLDEN") == "1"
LIMITS = {"beta": 100000, "noma_default": 30000, "noma_uniform": 30000, "noma_odd": 50000, "noma_g54": 40000,
          "noma_seed2": 20000}


@pytest.mark.parametrize("case", ["beta", "noma_default", "noma_uniform", "noma_odd", "noma_g54", "noma_seed2"])
def test_oracle_reproduces_reference(ob, case):
    g = load_golden(case)
    variant = ob.VARIANT_BETA_C if g["variant"] == "BETA_C" else ob.VARIANT_WITHNOMA_C
    blocks = split_stdout_blocks(g["stdout"])
    assert len(blocks) == len(g["trials"])
    rngs = {}
    checked = 0
    for k, tr in enumerate(g["trials"]):
        if not FULL and tr["nUE"] > LIMITS[case]:
            continue
        if not FULL and tr["seed"] > 0 and tr["nUE"] > 10000:
            continue
        rng = rngs.setdefault(tr["seed"], ob.Rng(ob.RNG_GLIBC, tr["seed"]))
        cfg = ob.make_cfg(tr["nUE"], variant=variant, **g["cfg_overrides"])
        res, ues = ob.run_trial(cfg, rng)
        assert ob.format_results(cfg, res).decode() == tr["results_text"], (case, tr["nUE"])
        assert ob.format_stdout(cfg, res).decode() == blocks[k], (case, tr["nUE"])
        logs = ob.format_logs(ues, cfg.nUE)
        assert len(logs) == tr["logs_bytes"]
        assert hashlib.sha256(logs).hexdigest() == tr["logs_sha256"], (case, tr["nUE"])
        checked += 1
    assert checked >= 2


def test_literal_scan_equals_matched_sets(ob):
    """The reference's own O(N) scan per preambleCollision call (Beta.c:321-330) and the O(1)
    matched-set bookkeeping give identical trials."""
    import numpy as np
    for variant, n, kw in ((0, 2500, {}), (1, 2500, {}), (1, 1500, dict(nPreamble=4, backoff=3, nGrantUL=3)),
                           (0, 1200, dict(maxMsg2TxCount=0, backoff=1, nPreamble=6))):
        out = []
        for mode in (ob.SCAN_SETS, ob.SCAN_LITERAL):
            cfg = ob.make_cfg(n, variant=variant, scan_mode=mode, **kw)
            res, ues = ob.run_trial(cfg, ob.Rng(ob.RNG_GLIBC, 3))
            d = res.as_dict()
            out.append((d, np.frombuffer(ues, dtype=np.int32).copy()))
        assert out[0][0] == out[1][0]
        assert (out[0][1] == out[1][1]).all()


def test_reference_published_statistics(ob):
    """results.csv / README.md:89-97 (100-seed means on the author's Mac, 12 grants): statistical
    agreement only (different libc), success ratio at nUE=30000 ~ 60.9 %."""
    cfg = ob.make_cfg(30000, variant=ob.VARIANT_WITHNOMA_C)
    res, _ = ob.run_trial(cfg, ob.Rng(ob.RNG_GLIBC, 0), want_ues=False)
    assert abs(100.0 * res.nSuccessUE / 30000 - 60.913) < 1.5


def load_ref_fuzz():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fuzz.json")))


def test_oracle_reproduces_reference_random_flags(ob):
    """63 random flag sets (-d -p -b -g -rc -mrc -s) of the REAL RandomAccessWithNOMA, the sweep points it
    finished in its time budget (tests/golden/fuzz_reference.py): Results.txt bytes, printed block and the
    SHA-256 of the per-UE Logs.txt of every point, with the rand() stream carried across the sweep.
    The default run checks the first sweep point of every Beta run and of every third Uniform run (60 000
    subframes each), so the CPU suite stays at a few minutes; PRACH_FULL_GOLDEN=1 checks all 175 trials (all
    verified that way), and the GPU suite checks all of them against the product path."""
    fz = load_ref_fuzz()
    assert len(fz["runs"]) >= 60
    checked = 0
    for r in fz["runs"]:
        rng = ob.Rng(ob.RNG_GLIBC, 0)
        for k, tr in enumerate(r["trials"]):
            if not FULL and (k > 0 or (r["cfg_overrides"]["uniform"] and r["run"] % 3 != 0)):
                break
            cfg = ob.make_cfg(tr["nUE"], variant=ob.VARIANT_WITHNOMA_C, **r["cfg_overrides"])
            res, ues = ob.run_trial(cfg, rng)
            what = (r["argv"], tr["nUE"])
            assert ob.format_results(cfg, res).decode() == tr["results_text"], what
            assert ob.format_stdout(cfg, res).decode() == r["stdout_blocks"][k], what
            logs = ob.format_logs(ues, cfg.nUE)
            assert len(logs) == tr["logs_bytes"], what
            assert hashlib.sha256(logs).hexdigest() == tr["logs_sha256"], what
            checked += 1
    assert checked >= 45


def test_oracle_sector_grants_reproduces_patched_reference(ob):
    """the per-sector UL-grant path the reference's author left commented out (RandomAccessWithNOMA.c:260,271-273:
    sectorGrants[6]; :312 the call that passes it; :626-637 the grantCheck[sector] test).  Pin = the reference compiled with
    exactly those lines swapped in by the sed recipe of oracle/Makefile (SED_SECTOR_GRANTS; the source streams from
    /root/reference into the compiler, no edited copy exists), run under random flag sets by tests/golden/fuzz_reference.py
    (PRACH_FUZZ_VARIANT=sector): Results.txt bytes, printed block and SHA-256 of the per-UE Logs.txt of every finished sweep
    point, the rand() stream carried across the sweep.  Default: the first point of every run; PRACH_FULL_GOLDEN=1: all."""
    import json
    fz = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fuzz_sector.json")))
    assert fz["sector_grants"] == 1 and len(fz["runs"]) >= 12
    checked = differs = 0
    for r in fz["runs"]:
        rng = ob.Rng(ob.RNG_GLIBC, 0)
        for k, tr in enumerate(r["trials"]):
            if not FULL and (k > 0 or (r["cfg_overrides"]["uniform"] and r["run"] % 2 != 0)):
                break
            cfg = ob.make_cfg(tr["nUE"], variant=ob.VARIANT_WITHNOMA_C, sector_grants=1, **r["cfg_overrides"])
            res, ues = ob.run_trial(cfg, rng)
            what = (r["argv"], tr["nUE"])
            assert ob.format_results(cfg, res).decode() == tr["results_text"], what
            assert ob.format_stdout(cfg, res).decode() == r["stdout_blocks"][k], what
            logs = ob.format_logs(ues, cfg.nUE)
            assert len(logs) == tr["logs_bytes"] and hashlib.sha256(logs).hexdigest() == tr["logs_sha256"], what
            checked += 1
            if k == 0:  # ... and the variant is not a no-op: with one cell-wide budget the same trial ends differently
                cfg0 = ob.make_cfg(tr["nUE"], variant=ob.VARIANT_WITHNOMA_C, **r["cfg_overrides"])
                res0, _ = ob.run_trial(cfg0, ob.Rng(ob.RNG_GLIBC, 0), want_ues=False)
                differs += ob.format_results(cfg0, res0).decode() != tr["results_text"]
    assert checked >= 8 and differs >= 3  # (at the first, lightly loaded sweep point the budgets rarely bind)
