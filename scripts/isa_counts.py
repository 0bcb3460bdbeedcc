#!/usr/bin/env python3
"""scripts/isa_counts.py [--vs FILE] — static gfx950 instruction counts per kernel (hipcc -S of every .hip under csrc/), by class.
Used to check that a refactoring of shared device code leaves the kernels' instruction streams alone and, with
-Rpass-analysis=kernel-resource-usage, to regenerate the register / spill table of profiles/rNN_resource_usage.md at HEAD."""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "5g-nr-randomaccess_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FILES = ["prach_lcluster.hip", "prach_cluster.hip", "prach_batch.hip", "prach_kernels.hip", "prach_noma.hip", "prach_noma_glibc.hip", "prach_stream.hip"]


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def classify(op):
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")): return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm", "s_setprio", "s_sendmsg", "s_getreg", "s_setreg", "s_memtime", "s_memrealtime")): return "other"
    if op.startswith(("s_load", "s_buffer", "s_store", "s_dcache", "s_atomic")): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("v_"): return "valu"
    return "other"


def main():
    res, usage = {}, {}
    for f in FILES:
        with tempfile.TemporaryDirectory() as d:
            asm = os.path.join(d, "k.s")
            p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-ffp-contract=off", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                                os.path.join(SRC, f), "-o", asm], capture_output=True, text=True)
            if p.returncode:
                sys.exit(p.stderr[-2000:])
            cur = None
            for line in p.stderr.split("\n"):
                m = re.search(r"remark: Function Name: (\S+)", line)
                if m: cur = m.group(1); usage[cur] = {}
                m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\d+)", line)
                if m and cur: usage[cur][m.group(1).replace(" Spill", "Spill").split(" ")[0]] = int(m.group(2))
            cur = None
            for line in open(asm):
                m = re.match(r"^(_Z\w+):", line)
                if m and m.group(1) in usage:
                    cur = m.group(1); res[cur] = collections.Counter(); continue
                if cur is None: continue
                if line.startswith("\t.end_amdhsa_kernel") or re.match(r"^\.Lfunc_end", line): cur = None; continue
                m = re.match(r"^\t([a-z]\w+)", line)
                if m and not m.group(1).startswith("."):
                    res[cur][classify(m.group(1))] += 1; res[cur]["total"] += 1
    names = demangle(list(res))
    table = {names[k].replace("(anonymous namespace)::", "").replace("prach::", "").split("(")[0]: dict(res[k], **{("r_" + a): b for a, b in usage[k].items()}) for k in res}
    if "--json" in sys.argv:
        print(json.dumps(table, indent=1, sort_keys=True)); return
    old = json.load(open(sys.argv[sys.argv.index("--vs") + 1])) if "--vs" in sys.argv else None
    print("| kernel | instructions | valu | salu | branch | lds | vmem | VGPRs | SGPR spills | VGPR spills | scratch B | waves/SIMD |" + (" vs |" if old else ""))
    print("|---|---|---|---|---|---|---|---|---|---|---|---|" + ("---|" if old else ""))
    for k in sorted(table):
        t = table[k]
        vs = ""
        if old is not None:
            vs = f" {100.0 * (t['total'] - old[k]['total']) / old[k]['total']:+.1f} % |" if k in old else " new |"
        print(f"| `{k}` | {t['total']} | {t.get('valu', 0)} | {t.get('salu', 0)} | {t.get('branch', 0)} | {t.get('lds', 0)} | {t.get('vmem', 0)} | {t.get('r_VGPRs', '')} | "
              f"{t.get('r_SGPRsSpill', '?')} | {t.get('r_VGPRsSpill', '?')} | {t.get('r_ScratchSize', '')} | {t.get('r_Occupancy', '')} |" + vs)

if __name__ == "__main__":
    main()
