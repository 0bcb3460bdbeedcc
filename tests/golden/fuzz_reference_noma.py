#!/usr/bin/env python3
"""Random-parameter runs of the REAL NOMA.c (oracle/_ref/NOMA_params = the reference's NOMA.c compiled as it
lies + oracle/noma_params_main.c, which only sets its file-scope parameters NOMA.c:41-57), digested into
tests/golden/ref_fuzz_noma.json: the 'nUE nSucc succ% avgTx avgDelay' lines (NOMA.c:606-632) of the sweep
points of seed 0 each run finished within its wall-clock budget (at most KEEP).

Build container only.  usage: tests/golden/fuzz_reference_noma.py [nruns] [budget_s] [jobs]
"""
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# PRACH_FUZZ_VARIANT=nonsector: the same for oracle/_ref/NOMA_nonsector_params — NOMA.c with the author's commented-out
# cell-wide grouping call enabled by the sed recipe in oracle/Makefile -> ref_fuzz_noma_nonsector.json
VARIANT = os.environ.get("PRACH_FUZZ_VARIANT", "")
BIN = os.path.join(ROOT, "oracle", "_ref", "NOMA_nonsector_params" if VARIANT == "nonsector" else "NOMA_params")
RUNS = os.path.join(ROOT, "oracle", "_ref", "runs", "fuzz_noma" + ("_nonsector" if VARIANT == "nonsector" else ""))
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_fuzz_noma_nonsector.json" if VARIANT == "nonsector" else "ref_fuzz_noma.json")
KEEP = 4
SEED = 777 if VARIANT != "nonsector" else 888


def param_sets(n):
    rs = np.random.RandomState(SEED)
    out = []
    for _ in range(n):
        out.append({"-p": int(rs.choice([1, 2, 3, 8, 33, 54, 64, 100])), "-b": int(rs.choice([1, 2, 5, 20, 33, 60])),
                    "-g": int(rs.choice([1, 2, 3, 5, 12, 40])), "-rw": int(rs.randint(1, 9)),
                    "-m": int(rs.choice([1, 2, 3, 10, 25])), "-s": int(rs.choice([1, 2, 3, 5, 5, 8, 10])),
                    "-c": float(rs.choice([50.0, 250.0, 500.0, 500.0, 2000.0]))})
    return out


def run_one(k, ps, budget):
    d = os.path.join(RUNS, f"{k:03d}")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, "TestResults"))
    argv = [x for kv in ps.items() for x in (kv[0], str(kv[1]))]
    with open(os.path.join(d, "stdout.txt"), "w") as so:
        try:
            subprocess.run([BIN] + argv, cwd=d, stdout=so, stderr=subprocess.DEVNULL, timeout=budget)
        except subprocess.TimeoutExpired:
            pass
    lines = []
    for l in open(os.path.join(d, "stdout.txt")).read().split("\n"):
        if l == "Done":
            break
        if l and len(l.split()) == 5:
            lines.append(l)
    over = {"nPreamble": ps["-p"], "backoff": ps["-b"], "nGrantUL": ps["-g"], "maxRarWindow": ps["-rw"], "maxMsg1ReTx": ps["-m"],
            "accessTime": ps["-s"], "cellRadius": ps["-c"]}
    return {"run": k, "argv": argv, "cfg_overrides": over, "lines": lines[:KEEP]}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    with ThreadPoolExecutor(jobs) as ex:
        runs = list(ex.map(lambda kv: run_one(kv[0], kv[1], budget), enumerate(param_sets(n))))
    kept = [r for r in runs if r["lines"]]
    json.dump({"generated_by": "tests/golden/fuzz_reference_noma.py (reference NOMA.c compiled from /root/reference + oracle/noma_params_main.c)",
               "seed": SEED, "program": "NOMA" + (" + cell-wide grouping call un-commented (oracle/Makefile SED_NONSECTOR_NOMA)" if VARIANT == "nonsector" else ""),
               "variant": "NOMA_C", "nonsector": int(VARIANT == "nonsector"), "runs": kept}, open(OUT, "w"), indent=1)
    print(f"{len(kept)} of {n} runs finished at least one sweep point in {budget:.0f} s; wrote {OUT}")


if __name__ == "__main__":
    main()
