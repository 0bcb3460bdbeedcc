#!/usr/bin/env python3
"""Static instruction mix per SOURCE LINE of one kernel (hipcc -gline-tables-only -S --cuda-device-only): which lines of the source the scalar / branch
instructions of a kernel come from.  usage: isa_lines.py file.s kernel-name-substring [min-count]"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(src).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l)
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith("\t.section"))
def cls(op):
    if op.startswith("s_cbranch") or op == "s_branch": return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): return "wait"
    if op.startswith("s_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "lane"
    if op.startswith("v_cmp"): return "vcmp"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    return "vmem"
cur = ("?", 0)
tab = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
    tab[cur][cls(t.split()[0])] += 1
rows = sorted(tab.items(), key=lambda kv: -(kv[1]["salu"] + kv[1]["branch"]))
print("file:line  salu branch lane vcmp valu lds vmem wait")
for (f, ln), c in rows:
    if c["salu"] + c["branch"] < minc: break
    print(f"{f}:{ln}  {c['salu']} {c['branch']} {c['lane']} {c['vcmp']} {c['valu']} {c['lds']} {c['vmem']} {c['wait']}")
