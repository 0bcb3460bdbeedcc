#!/usr/bin/env python3
"""Digest the raw outputs of the REAL reference programs (tests/golden/run_reference.sh, run in the
build container from the sources under /root/reference) into small JSON fixtures.

A fixture is data only: the reference's printed numbers, the contents of its Results.txt files
(wall-clock latency stripped), the SHA-256 of every per-UE Logs.txt (24 MB each at nUE=100k: the
strongest pin — every logged field of every UE) and a handful of sampled log lines.  No reference
source text is stored.

usage: tests/golden/digest_reference.py [case ...]     (default: every finished case)
"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
RUNS = os.path.join(ROOT, "oracle", "_ref", "runs")
OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # case: (program, argv, variant, dict of cfg overrides)
    "beta": ("RandomAccessSimulatorBeta", [], "BETA_C", {}),
    "noma_default": ("RandomAccessWithNOMA", [], "WITHNOMA_C", {}),
    "noma_uniform": ("RandomAccessWithNOMA", ["-d", "1", "-t", "1"], "WITHNOMA_C", {"uniform": 1}),
    "noma_odd": ("RandomAccessWithNOMA", ["-p", "64", "-b", "10", "-g", "8", "-rc", "4", "-mrc", "5", "-s", "10", "-t", "1"],
                 "WITHNOMA_C", {"nPreamble": 64, "backoff": 10, "nGrantUL": 8, "maxRarWindow": 5, "maxMsg2TxCount": 4, "accessTime": 10}),
    "noma_g54": ("RandomAccessWithNOMA", ["-g", "54", "-t", "1"], "WITHNOMA_C", {"nGrantUL": 54}),
    "noma_seed2": ("RandomAccessWithNOMA", ["-d", "0", "-p", "30", "-b", "40", "-g", "20", "-rc", "3", "-mrc", "20", "-t", "3"],
                   "WITHNOMA_C", {"nPreamble": 30, "backoff": 40, "nGrantUL": 20, "maxRarWindow": 4, "maxMsg2TxCount": 19}),
    "noma_c": ("NOMA", [], "NOMA_C", {}),
}
SAMPLE_LINES = (0, 1, 2, 777, 4999, 9999)


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def digest(case):
    d = os.path.join(RUNS, case)
    if not os.path.isdir(d):
        return None
    prog, argv, variant, over = CASES[case]
    out = {"case": case, "program": prog, "argv": argv, "variant": variant, "cfg_overrides": over,
           "generated_by": "tests/golden/run_reference.sh + digest_reference.py (reference compiled from /root/reference)",
           "trials": []}
    stdout = open(os.path.join(d, "stdout.txt")).read()
    # wall-clock lines are not reproducible
    out["stdout"] = "\n".join(l for l in stdout.split("\n") if not l.startswith("Latency:"))
    if case == "noma_c":
        files = {}
        tr = os.path.join(d, "TestResults")
        for fn in sorted(os.listdir(tr)):
            files[fn] = open(os.path.join(tr, fn)).read()
        out["files"] = files
        return out
    for sub in sorted(os.listdir(d)):
        sd = os.path.join(d, sub)
        if not os.path.isdir(sd):
            continue
        for fn in sorted(os.listdir(sd)):
            m = re.match(r"(\d+)_(\d+)_(\d+)_Results\.txt$", fn)
            if not m:
                continue
            seed, npre, nue = map(int, m.groups())
            text = open(os.path.join(sd, fn)).read()
            if variant == "BETA_C":  # 6th line = cumulative clock() seconds, no newline (Beta.c:481)
                text = "\n".join(text.split("\n")[:5]) + "\n"
            logname = f"{seed}_{npre}_UE{nue:05d}_Logs.txt"
            lp = os.path.join(sd, logname)
            rec = {"seed": seed, "nPreamble": npre, "nUE": nue, "dir": sub, "results_file": fn, "results_text": text}
            if os.path.exists(lp):
                rec["logs_file"] = logname
                rec["logs_sha256"] = sha256_file(lp)
                rec["logs_bytes"] = os.path.getsize(lp)
                with open(lp) as f:
                    lines = f.readlines()
                rec["logs_sample"] = {str(i): lines[i].rstrip("\n") for i in SAMPLE_LINES if i < len(lines)}
            out["trials"].append(rec)
    out["trials"].sort(key=lambda r: (r["seed"], r["nUE"]))
    return out


def main():
    cases = sys.argv[1:] or list(CASES)
    for c in cases:
        log = os.path.join(ROOT, "oracle", "_ref", f"run_{c}.log")
        if not os.path.exists(log) or "done" not in open(log).read():
            print(f"skip {c}: run not finished")
            continue
        dg = digest(c)
        if dg is None:
            continue
        with open(os.path.join(OUT, f"{c}.json"), "w") as f:
            json.dump(dg, f, indent=1)
        print(f"wrote tests/golden/{c}.json ({len(dg.get('trials', []))} trials)")


if __name__ == "__main__":
    main()
