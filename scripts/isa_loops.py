#!/usr/bin/env python3
"""Static look at a kernel's ISA (hipcc -S --cuda-device-only): every backward branch = a loop; per loop the instruction mix of its body (by class) —
where the scalar instructions of a kernel sit.  usage: isa_loops.py file.s mangled-kernel-name-substring"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(key_end := l.split(":")[0]) or (l.startswith("_ZN") and key in l))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith("\t.section"))
body = lines[start:end]
label_at = {}
ins = []
for k, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        label_at[m.group(1)] = len(ins)
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    ins.append((op, t, k + start + 1))
def cls(op):
    if op.startswith("s_cbranch") or op == "s_branch": return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "lane"
    if op.startswith("v_cmp"): return "vcmp"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"
loops = []
for k, (op, t, ln) in enumerate(ins):
    if cls(op) == "branch":
        tgt = t.split()[-1]
        if tgt in label_at and label_at[tgt] <= k:
            loops.append((label_at[tgt], k))
loops.sort(key=lambda ab: (ab[0], -ab[1]))
tot = collections.Counter(cls(op) for op, _, _ in ins)
print("kernel lines", start + 1, end, "instructions", len(ins), dict(tot))
for a, b in loops:
    c = collections.Counter(cls(op) for op, _, _ in ins[a:b + 1])
    depth = sum(1 for a2, b2 in loops if a2 <= a and b2 >= b) - 1
    if b - a < int(sys.argv[3]) if len(sys.argv) > 3 else 0: continue
    print("  " * depth + f"loop lines {ins[a][2]}-{ins[b][2]} n={b - a + 1}", dict(c))
