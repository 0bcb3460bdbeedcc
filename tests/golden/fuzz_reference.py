#!/usr/bin/env python3
"""Random-flag runs of the REAL reference program (oracle/_ref/RandomAccessWithNOMA, compiled by
`make -C oracle ref` from /root/reference — nothing copied), digested into tests/golden/ref_fuzz.json.

The reference has no nUE flag: every run is its own 10k..100k sweep, O(nUE^2) per subframe.  Each run
gets a wall-clock budget; the sweep points it FINISHED in that budget (Results.txt present: it is
written after the per-UE log, WithNOMA:363-366) are kept, at most KEEP per run.  A fixture is data only:
argv, the printed block, the Results.txt text and the SHA-256 of the per-UE Logs.txt.

Build container only (needs the compiled reference).  usage: tests/golden/fuzz_reference.py [nruns] [budget_s] [jobs]
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# PRACH_FUZZ_VARIANT=sector: the same for oracle/_ref/RandomAccessWithNOMA_sector — the reference with the author's own
# commented-out per-sector grant lines enabled by the sed recipe in oracle/Makefile -> ref_fuzz_sector.json
VARIANT = os.environ.get("PRACH_FUZZ_VARIANT", "")
BIN = os.path.join(ROOT, "oracle", "_ref", "RandomAccessWithNOMA" + ("_sector" if VARIANT == "sector" else ""))
RUNS = os.path.join(ROOT, "oracle", "_ref", "runs", "fuzz" + ("_sector" if VARIANT == "sector" else ""))
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_fuzz_sector.json" if VARIANT == "sector" else "ref_fuzz.json")
KEEP = 3
SEED = 4242 if VARIANT != "sector" else 5151


def flag_sets(n):
    rs = np.random.RandomState(SEED)
    out = []
    for _ in range(n):
        f = {"-d": int(rs.rand() < 0.3),
             "-p": int(rs.choice([2, 3, 5, 8, 16, 30, 54, 64, 100])),
             "-b": int(rs.choice([1, 2, 5, 10, 20, 33, 60])),
             "-g": int(rs.choice([1, 2, 3, 5, 8, 12, 20, 54])),
             "-rc": int(rs.randint(1, 10)),
             "-mrc": int(rs.choice([1, 2, 3, 5, 10, 20, 50])),
             "-s": int(rs.choice([5, 6, 7, 8, 10, 13, 20]))}
        out.append(f)
    return out


def run_one(k, flags, budget):
    d = os.path.join(RUNS, f"{k:03d}")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    argv = [x for kv in flags.items() for x in (kv[0], str(kv[1]))] + ["-t", "1"]
    with open(os.path.join(d, "stdout.txt"), "w") as so:
        try:
            subprocess.run(["stdbuf", "-oL", BIN] + argv,  # line-buffered: the run is cut off by the budget
                           cwd=d, stdout=so, stderr=subprocess.DEVNULL, timeout=budget)
        except subprocess.TimeoutExpired:
            pass
    return digest(k, flags, argv, d)


def digest(k, flags, argv, d):
    over = {"uniform": 1 if flags["-d"] == 1 else 0, "nPreamble": flags["-p"], "backoff": flags["-b"], "nGrantUL": flags["-g"],
            "maxRarWindow": flags["-rc"] + 1, "maxMsg2TxCount": flags["-mrc"] - 1, "accessTime": flags["-s"]}  # WithNOMA:123-140
    trials = []
    for sub in sorted(os.listdir(d)):
        sd = os.path.join(d, sub)
        if not os.path.isdir(sd):
            continue
        for fn in sorted(os.listdir(sd)):
            m = re.match(r"(\d+)_(\d+)_(\d+)_Results\.txt$", fn)
            if not m:
                continue
            seed, npre, nue = map(int, m.groups())
            lp = os.path.join(sd, f"{seed}_{npre}_UE{nue:05d}_Logs.txt")
            text = open(os.path.join(sd, fn)).read()
            if not os.path.exists(lp) or text.count("\n") < 5:
                continue
            h = hashlib.sha256(open(lp, "rb").read()).hexdigest()
            trials.append({"seed": seed, "nUE": nue, "results_text": text, "logs_sha256": h, "logs_bytes": os.path.getsize(lp)})
    trials.sort(key=lambda r: r["nUE"])
    trials = trials[:KEEP]
    # the printed block of each kept point (wall-clock lines are not reproducible)
    lines = [l for l in open(os.path.join(d, "stdout.txt")).read().split("\n") if not l.startswith("Latency:")]
    return {"run": k, "argv": argv, "cfg_overrides": over, "trials": trials, "stdout": "\n".join(lines)}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    sets = flag_sets(n)
    with ThreadPoolExecutor(jobs) as ex:
        runs = list(ex.map(lambda kv: run_one(kv[0], kv[1], budget), enumerate(sets)))
    kept = [r for r in runs if r["trials"]]
    for r in kept:  # the printed block of every kept point
        blocks, cur = [], None
        for line in r.pop("stdout").split("\n"):
            if line.startswith("-------- "):
                if cur is not None:
                    blocks.append("\n".join(cur) + "\n")
                cur = [line]
            elif cur is not None and line != "":
                cur.append(line)
        if cur is not None:
            blocks.append("\n".join(cur) + "\n")
        r["stdout_blocks"] = blocks[:len(r["trials"])]
    json.dump({"generated_by": "tests/golden/fuzz_reference.py (reference compiled from /root/reference)", "seed": SEED,
               "program": "RandomAccessWithNOMA" + (" + per-sector grant lines un-commented (oracle/Makefile SED_SECTOR_GRANTS)" if VARIANT == "sector" else ""),
               "variant": "WITHNOMA_C", "sector_grants": int(VARIANT == "sector"), "runs": kept}, open(OUT, "w"), indent=1)
    print(f"{len(kept)} of {n} runs finished at least one sweep point in {budget:.0f} s; wrote {OUT}")


if __name__ == "__main__":
    main()
