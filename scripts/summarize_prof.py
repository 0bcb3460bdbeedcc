#!/usr/bin/env python3
"""Condense a gpurun_out/prof_* directory (rocprofv3 --kernel-trace --stats run + separate --pmc
FETCH_SIZE / WRITE_SIZE passes of the same bench command) into tracked files under profiles/.
usage: scripts/summarize_prof.py gpurun_out/prof_r01 r01 [bench-json-log]"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)  # gpurun merges runs into the same directory: take the latest
stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
main = max(rows, key=lambda r: float(r["TotalDurationNs"]))
pmc = {}
for kind, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    fs = glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(max(fs, key=os.path.getmtime)))
            if r["Counter_Name"] == ctr and r["Kernel_Name"] == main["Name"]]
    pmc[ctr] = vals
    with open(os.path.join(out, f"{tag}_{kind}.csv"), "w") as f:
        f.write("kernel,counter,value_KB\n")
        for v in vals:
            f.write(f"\"{main['Name']}\",{ctr},{v}\n")
fetch = sum(pmc.get("FETCH_SIZE", [0])) / max(1, len(pmc.get("FETCH_SIZE", [])))
write = sum(pmc.get("WRITE_SIZE", [0])) / max(1, len(pmc.get("WRITE_SIZE", [])))
# MI355X_MICROARCH.md §HBM: counters are in KB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of
# a wide coalesced streaming read (16 B/lane) -> double it; WRITE_SIZE is exact for 16 B/lane stores.
hbm = (2.0 * fetch + write) * 1024.0
traffic = {"tag": tag, "kernel": main["Name"], "avg_kernel_ns": float(main["AverageNs"]), "calls": int(main["Calls"]),
           "FETCH_SIZE_KB_avg": fetch, "WRITE_SIZE_KB_avg": write, "fetch_correction": 2.0,
           "hbm_bytes_per_launch": hbm,
           "method": "separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras`"}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
with open(os.path.join(out, f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary {tag}\n\ncommand: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras`\n\n")
    f.write("| kernel | calls | avg ms | % |\n|---|---|---|---|\n")
    for r in rows:
        f.write(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {r['Percentage']} |\n")
    f.write(f"\nPMC (own passes): FETCH_SIZE avg {fetch:.1f} KB (x2 gfx950 correction), WRITE_SIZE avg {write:.1f} KB "
            f"=> HBM traffic {hbm/1e6:.1f} MB per launch.\n")
    if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
        line = [l for l in open(sys.argv[3]) if l.startswith("{")]
        if line:
            f.write("\nbench line of the same build:\n\n```json\n" + line[-1].strip() + "\n```\n")
print(json.dumps(traffic))
