#!/usr/bin/env python3
"""Condense one scripts/gpu_pmc.sh run (gpurun_out/prof_TAG/NAME_*) into two tracked files under profiles/:
   profiles/OUT_kernel_stats.csv  (rocprofv3 --kernel-trace --stats of the command)
   profiles/OUT_pmc.csv           (every counter of every --pmc pass: kernel, counter, dispatches, average value per dispatch)
usage: scripts/summarize_prof3.py gpurun_out/prof_r03b c3 r03_config3"""
import collections, csv, glob, os, shutil, sys
src, name, outname = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
f = glob.glob(os.path.join(src, name + "_trace", "*", "*kernel_stats.csv"))
if f:
    shutil.copy(f[0], os.path.join(out, outname + "_kernel_stats.csv"))
rows = []
for d in sorted(glob.glob(os.path.join(src, name + "_*"))):
    if not os.path.isdir(d) or d.endswith("_trace"):
        continue
    for cf in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(cf)):
            if "rocclr" in r["Kernel_Name"]:
                continue
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            rows.append((os.path.basename(d)[len(name) + 1:], k, c, len(v), sum(v) / len(v)))
with open(os.path.join(out, outname + "_pmc.csv"), "w") as fo:
    fo.write("pass,kernel,counter,dispatches,average_value_per_dispatch\n")
    for r in rows:
        fo.write(f"{r[0]},\"{r[1]}\",{r[2]},{r[3]},{r[4]:.6g}\n")
print(f"wrote profiles/{outname}_kernel_stats.csv and profiles/{outname}_pmc.csv ({len(rows)} counters)")
