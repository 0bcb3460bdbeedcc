"""oracle/census_gate.py — TEST INFRASTRUCTURE (design study, CPU only): the go / no-go gate of VERDICT r3 item 5 — "ownership by preamble bucket" for the
single-trial kernel.  Runs the oracle on BASELINE configs[1] (nUE = 100 000, Beta, 54 preambles; Beta.c's 54 grants and the 12-grant default) with its census
switched on and prints, per subframe: preamble (re)draws = records that would have to MIGRATE between workgroups, singleton calls = callers whose index-ordered
grant ranking (Beta.c:332-347) would need a SECOND exchange, and what a 32-workgroup cluster's one granule round carries today.
usage: python3 oracle/census_gate.py [nUE]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob
L = ob.lib()
L.oracle_set_census.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
L.oracle_set_census.restype = None
nUE = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
G, LEPF = 32, 32  # workgroups of the lean cluster, event granules of every mailbox fetched in round 1 (prach_lcluster.hip)
for name, kw in (("Beta.c as committed (54 grants)", dict(variant=ob.VARIANT_BETA_C)), ("RandomAccessWithNOMA defaults (12 grants)", dict(variant=ob.VARIANT_WITHNOMA_C))):
    T = 10000
    red, sing, calls = (np.zeros(T, dtype=np.int32) for _ in range(3))
    L.oracle_set_census(red.ctypes.data, sing.ctypes.data, calls.ctypes.data, T)
    res, _ = ob.run_trial(ob.make_cfg(nUE, **kw), ob.Rng(ob.RNG_PHILOX, 0), want_ues=False)
    L.oracle_set_census(None, None, None, 0)
    n = int(res.time_exit) + 1 if res.time_exit < T else T
    r, s, c = red[:n], sing[:n], calls[:n]
    print(f"{name}: nUE {nUE}, {n} subframes, {res.nSuccessUE} successes")
    print(f"  preamble (re)draws per subframe: mean {r.mean():.1f}, 99th percentile {np.percentile(r, 99):.0f}, max {r.max()} "
          f"(a migrating 28-byte LDS record = 4 granules: mean {4 * r.mean() / G:.1f}, max {4 * r.max() / G:.1f} granules per mailbox against {LEPF} fetched in round 1)")
    print(f"  singleton calls per subframe:     mean {s.mean():.1f}, 99th percentile {np.percentile(s, 99):.0f}, max {s.max()}; subframes with at least one: {100.0 * (s > 0).mean():.1f} % "
          f"(each of them needs a second cluster-wide exchange before the grants are known: +1 all-gather of ~1 300 cycles on the subframe's chain)")
    print(f"  calls per subframe:               mean {c.mean():.1f}, max {c.max()}")
