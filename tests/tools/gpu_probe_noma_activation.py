"""Probe: activeUE on the device (noma_activation_kernel) against the host form (the reference's libm), UE by UE.
usage: gpu_probe_noma_activation.py [UEs in millions, default 8]
Prints the distribution of the gain's distance in ulps, the UEs the kernel flagged (recomputed on the host), and the time of both forms."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
m = g.load_package()
eng = m.Engine(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hist = np.zeros(66, np.int64)
nflag = bad = 0
th = td = 0.0
worst_l = 0.0
for s in range(M):
    cfg = m.make_cfg(1000000, variant=m.VARIANT_NOMA_C, rng_mode=m.RNG_PHILOX, seed=1000 + s, cellRadius=[400.0, 60.0, 2000.0, 36.0][s % 4])
    t0 = time.time(); hp, hs, hg, hl, hn = m.noma_activation_table(cfg); th += time.time() - t0
    t0 = time.time(); dp, ds, dg, dl, dn, fl = m.noma_activation_table_device(eng, cfg); td += time.time() - t0
    f = fl != 0
    bad += int((hp != dp).sum() + (hs != ds).sum() + (hn != dn).sum() + (hg[f] != dg[f]).sum() + (hl[f] != dl[f]).sum())
    ulp = np.abs(hg.view(np.int64) - dg.view(np.int64))[~f]
    hist += np.bincount(np.minimum(ulp, 65), minlength=66)
    worst_l = max(worst_l, float(np.abs(hl - dl)[~f].max()))
    nflag += int(f.sum())
    print(f"seed {1000 + s} R={cfg.cellRadius:.0f}: flagged {int(f.sum())}, max gain distance {int(ulp.max())} ulp, draws/UE {hn.mean():.4f}", flush=True)
tot = hist.sum()
print(f"{M}M UEs: mismatching preamble/sector/draw-count/flagged-gain entries: {bad}; flagged {nflag} ({nflag / (M * 1e6):.2e})")
print("gain distance (ulps): " + ", ".join(f"{k}: {hist[k] / tot:.4f}" for k in range(9)) + f", 9-64: {hist[9:65].sum()}, >64: {hist[65]}")
print(f"largest |ln gain| difference {worst_l:.3e}; host table {th / M * 1e3:.1f} ms per 1M UEs (one core), device table incl. copies back {td / M * 1e3:.1f} ms")
