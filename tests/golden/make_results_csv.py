#!/usr/bin/env python3
"""tests/golden/make_results_csv.py — BUILD CONTAINER ONLY (needs /root/reference): pins `results.csv`
with the reference's OWN post-processing script.

What it does
  1. runs the oracle (Beta.c program, 12 UL grants = the published experiment, Philox seeds 0..99, the ten
     nUE points 10 000..100 000) and formats every trial's six-line Results.txt exactly as Beta.c:460-482
     does (the sixth line, the reference's cumulative clock() seconds, is wall-clock in the reference and
     synthesised here from a fixed LCG so that the fixture is reproducible);
  2. lays the 1000 texts out as ./Beta_SimulationResults/{seed}_54_{nUE}_Results.txt in a scratch
     directory and EXECUTES /root/reference/AveragePerformance.py THERE, UNMODIFIED, BY PATH (nothing of it
     is copied into this repository);
  3. stores inputs (the 1000 texts) and output (the bytes of the results.csv the script wrote) in
     tests/golden/results_csv.json.

tests/test_host_logic.py then compares prach_results_csv_accumulate / prach_results_csv_row (host C,
csrc/prach_host.c) with those bytes, and tests/test_gpu_parity.py runs the same 1000-trial experiment through
the CLI on the GPU and compares the first five columns of its --csv with them (the sixth is wall clock).  There
is ONE fixture block: the unmodified script hard-codes 100 seeds x 10 points, so no smaller grid can be pinned.

Usage:  python tests/golden/make_results_csv.py [--threads 8]
"""
from __future__ import annotations

import argparse
import base64
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF_SCRIPT = "/root/reference/AveragePerformance.py"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    if not os.path.exists(REF_SCRIPT):
        raise SystemExit(f"{REF_SCRIPT} not found: this script only runs in the build container")
    from oracle import binding as ob

    seeds = list(range(100))
    points = list(range(10000, 110000, 10000))

    def latency(seed, n):  # fixed LCG -> microsecond-resolution "seconds" (the reference prints clock() with %lf)
        x = (seed * 1103515245 + n * 12345 + 1013904223) & 0x7FFFFFFF
        return (x % 2_000_000_000) / 1e6

    def one(job):
        seed, n = job
        cfg = ob.make_cfg(n, variant=ob.VARIANT_BETA_C, nGrantUL=12)
        res, _ = ob.run_trial(cfg, ob.Rng(ob.RNG_PHILOX, seed), want_ues=False)
        return (seed, n), ob.format_results(cfg, res).decode() + "%f" % latency(seed, n)

    jobs = [(s, n) for n in reversed(points) for s in seeds]  # longest first
    with ThreadPoolExecutor(max_workers=args.threads) as ex:  # the oracle is plain C behind ctypes: no GIL
        texts = dict(ex.map(one, jobs))

    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "Beta_SimulationResults"))
        for (seed, n), t in texts.items():
            with open(os.path.join(d, "Beta_SimulationResults", f"{seed}_54_{n}_Results.txt"), "w") as f:
                f.write(t)
        subprocess.check_call([sys.executable, REF_SCRIPT], cwd=d)  # the reference's script, unmodified, by path
        csv_bytes = open(os.path.join(d, "results.csv"), "rb").read()

    out = {
        "made_by": "tests/golden/make_results_csv.py: oracle Results.txt texts (Beta.c program, nGrantUL=12, Philox seeds 0..99) "
                   "averaged by /root/reference/AveragePerformance.py executed unmodified in a scratch directory",
        "seeds": seeds, "points": points,
        "results_txt": {f"{s}_{n}": texts[(s, n)] for s in seeds for n in points},
        "results_csv_b64": base64.b64encode(csv_bytes).decode(),
    }
    path = os.path.join(ROOT, "tests", "golden", "results_csv.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"wrote {path}: {len(texts)} texts, results.csv {len(csv_bytes)} bytes")
    print(csv_bytes.decode())


if __name__ == "__main__":
    main()
