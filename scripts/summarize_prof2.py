#!/usr/bin/env python3
"""Condense gpurun_out/prof_TAG (scripts/gpu_profile.sh) into tracked files under profiles/.
usage: scripts/summarize_prof2.py gpurun_out/prof_r02a r02 [bench.json]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")


def stats(d):
    f = glob.glob(os.path.join(src, d, "*", "*kernel_stats.csv"))
    return list(csv.DictReader(open(f[0]))) if f else []


def pmc(d):
    f = glob.glob(os.path.join(src, d, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


for d, name in (("trace", "kernel_stats"), ("c3_trace", "config3_kernel_stats"), ("probe_trace", "copy_probe_kernel_stats")):
    f = glob.glob(os.path.join(src, d, "*", "*kernel_stats.csv"))
    if f:
        shutil.copy(f[0], os.path.join(out, f"{tag}_{name}.csv"))
for d, name in (("pmc_fetch", "pmc_fetch"), ("pmc_write", "pmc_write"), ("c3_fetch", "config3_pmc_fetch"), ("c3_write", "config3_pmc_write"),
                ("c3_sq", "config3_pmc_sq"), ("probe_fetch", "copy_probe_pmc_fetch"), ("probe_write", "copy_probe_pmc_write")):
    avg, cnt = pmc(d)
    with open(os.path.join(out, f"{tag}_{name}.csv"), "w") as f:
        f.write("kernel,counter,dispatches,average_value\n")
        for (k, c), v in sorted(avg.items()):
            f.write(f"\"{k}\",{c},{cnt[(k, c)]},{v}\n")

main = max(stats("trace"), key=lambda r: float(r["TotalDurationNs"]))
c3 = max(stats("c3_trace"), key=lambda r: float(r["TotalDurationNs"]))
fa, _ = pmc("pmc_fetch"); wa, _ = pmc("pmc_write"); f3, _ = pmc("c3_fetch"); w3, _ = pmc("c3_write"); s3, _ = pmc("c3_sq")
pf, _ = pmc("probe_fetch"); pw, _ = pmc("probe_write")
fetch = fa[(main["Name"], "FETCH_SIZE")]; write = wa[(main["Name"], "WRITE_SIZE")]
fetch3 = f3[(c3["Name"], "FETCH_SIZE")]; write3 = w3[(c3["Name"], "WRITE_SIZE")]
probe = {k[0][:48]: v for k, v in pf.items() if "read_kernel" in k[0]}
probe_w = [v for k, v in pw.items() if "write_kernel" in k[0]][0]
corr = 1048576.0 / (sum(probe.values()) / len(probe))  # 1 GiB = 1 048 576 KB was read by every read_kernel
hbm = (corr * fetch + write) * 1024.0
hbm3 = (corr * fetch3 + write3) * 1024.0
traffic = {"tag": tag, "kernel": main["Name"], "avg_kernel_ns": float(main["AverageNs"]), "calls": int(main["Calls"]),
           "FETCH_SIZE_KB_avg": fetch, "WRITE_SIZE_KB_avg": write, "fetch_correction": corr,
           "hbm_bytes_per_launch": hbm, "hbm_GBps": hbm / float(main["AverageNs"]),
           "calibration": "profiles/tools/copy_probe.hip under the same counters: 1 GiB read with 4 / 8 / 16 B per lane reports FETCH_SIZE = "
                          + ", ".join(f"{v:.0f} KB" for v in probe.values()) + f" (x{corr:.3f}); 1 GiB written with 8 B per lane reports WRITE_SIZE = {probe_w:.0f} KB",
           "method": "separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras` (scripts/gpu_profile.sh)"}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
with open(os.path.join(out, f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary {tag} (scripts/gpu_profile.sh, MI355X)\n\n## bench workload (BASELINE configs[1]: one Beta.c trial, nUE = 100 000)\n\n")
    f.write("command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras`\n\n| kernel | calls | avg ms | % |\n|---|---|---|---|\n")
    for r in stats("trace"):
        f.write(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {r['Percentage']} |\n")
    f.write(f"\nPMC (own passes): FETCH_SIZE avg {fetch:.0f} KB (x{corr:.2f}, calibrated below), WRITE_SIZE avg {write:.0f} KB => {hbm/1e9:.2f} GB per launch "
            f"= {hbm/float(main['AverageNs']):.1f} GB/s; algorithmic 32 B x 1e9 updates = 32 GB per launch: the trial's state is LDS / L2 resident, "
            "the traffic is the exchange granules, the cold per-UE fields and the final dump.\n")
    upd3 = 546976000000
    f.write(f"\n## BASELINE configs[2]: 1000 concurrent trials (`scripts/gpu_batch.py 100 1 0`)\n\n| kernel | calls | avg ms |\n|---|---|---|\n")
    for r in stats("c3_trace"):
        f.write(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} |\n")
    t3 = float(c3["AverageNs"]) * 1e-9
    f.write(f"\nFETCH_SIZE {fetch3/1e6:.1f} GB as counted (x{corr:.2f} = {corr*fetch3/1e6:.1f} GB), WRITE_SIZE {write3/1e6:.1f} GB, kernel {t3:.3f} s => "
            f"**{hbm3/t3/1e12:.2f} TB/s of real HBM traffic = {hbm3/t3/8e12*100:.0f} % of 8 TB/s** ({hbm3/t3/6.29e12*100:.0f} % of the 6.29 TB/s a copy reaches); "
            f"{upd3/t3:.3e} UE-subframe updates/s; {hbm3/upd3:.1f} B of HBM traffic per update.\n\n")
    wc = s3.get((c3["Name"], "SQ_WAVE_CYCLES"), 0)
    if wc:
        f.write("SQ counters (one pass): " + ", ".join(f"{c} {s3[(c3['Name'], c)]/wc*100:.1f} % of wave cycles" for c in ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"))
                + f"; {int(s3[(c3['Name'], 'SQ_WAVES')])} waves (16 per workgroup, one 1024-thread workgroup per CU: 4 waves per SIMD).\n")
    f.write("\n## FETCH_SIZE / WRITE_SIZE calibration (profiles/tools/copy_probe.hip, 1 GiB = 1 048 576 KB per kernel)\n\n| kernel | FETCH_SIZE KB | WRITE_SIZE KB |\n|---|---|---|\n")
    for k, v in pf.items():
        if "kernel" in k[0] and "rocclr" not in k[0]:
            f.write(f"| `{k[0][:60]}` | {v:.0f} | {pw.get((k[0], 'WRITE_SIZE'), 0):.0f} |\n")
    f.write(f"\nFETCH_SIZE reports 1/2 of the bytes for 4-, 8- and 16-byte-per-lane streaming loads alike (correction x{corr:.3f}); WRITE_SIZE is exact for 8-byte-per-lane stores.\n")
    if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
        line = [l for l in open(sys.argv[3]) if l.startswith("{")]
        if line:
            f.write("\n## bench line of the same build\n\n```json\n" + line[-1].strip() + "\n```\n")
print(json.dumps(traffic, indent=1))
