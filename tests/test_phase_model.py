"""The parallel-exact decomposition the HIP kernels implement (oracle/phase_model.c) against the
sequential restatement, on ordinary and deliberately odd parameters (late joiners, reset cycles
that re-join, passive members, Msg3-timeout re-entries, accessTime != 5, no grants at all)."""
import numpy as np
import pytest

CASES = [
    # variant, nUE, rng, nranges, overrides
    (0, 6000, 0, 16, {}),
    (1, 6000, 1, 16, {}),
    (1, 12000, 0, 5, {}),
    (0, 1500, 0, 61, dict(nPreamble=2, backoff=3, nGrantUL=3, maxRarWindow=2, maxMsg2TxCount=3, accessTime=6)),
    (1, 1500, 1, 3, dict(nPreamble=2, backoff=3, nGrantUL=1, maxRarWindow=3, maxMsg2TxCount=0, accessTime=6)),
    (1, 4000, 0, 16, dict(nPreamble=1, backoff=1, nGrantUL=12, maxRarWindow=2, maxMsg2TxCount=3)),
    (0, 500, 1, 3, dict(uniform=1, nPreamble=54, backoff=1, nGrantUL=1, maxRarWindow=8, maxMsg2TxCount=1, accessTime=10)),
    (0, 4000, 0, 1, dict(nPreamble=8, backoff=5, nGrantUL=12, maxRarWindow=2, maxMsg2TxCount=1)),
    (1, 1500, 0, 16, dict(nPreamble=200, backoff=1, nGrantUL=2, maxRarWindow=2, maxMsg2TxCount=3)),
    (0, 3000, 1, 7, dict(nPreamble=3, backoff=40, nGrantUL=12, maxRarWindow=6, maxMsg2TxCount=3, accessTime=6)),
    (1, 2000, 0, 2, dict(uniform=1, nPreamble=64, backoff=5, nGrantUL=12, maxRarWindow=2, maxMsg2TxCount=3)),
]


@pytest.mark.parametrize("variant,nUE,rng_mode,nranges,kw", CASES)
def test_model_equals_sequential(ob, variant, nUE, rng_mode, nranges, kw):
    cfg = ob.make_cfg(nUE, variant=variant, **kw)
    seed = 11
    r1, u1 = ob.run_trial(cfg, ob.Rng(rng_mode, seed))
    stream = ob.glibc_stream(seed, int(r1.draws) + 16) if rng_mode == ob.RNG_GLIBC else None
    r2, u2 = ob.model_run_trial(cfg, rng_mode, seed, stream, 0, nranges)
    d1, d2 = r1.as_dict(), r2.as_dict()
    d1.pop("collisionCalls"), d2.pop("collisionCalls")
    assert d1 == d2
    a1 = np.frombuffer(u1, dtype=np.int32)
    a2 = np.frombuffer(u2, dtype=np.int32)
    assert (a1 == a2).all()


def test_model_stream_offset(ob):
    """glibc mode continues a seed's stream at an offset (the chained nUE sweep)."""
    cfg = ob.make_cfg(3000, variant=1)
    rng = ob.Rng(ob.RNG_GLIBC, 4)
    ob.run_trial(ob.make_cfg(2000, variant=1), rng, want_ues=False)
    off = rng.consumed()
    r1, u1 = ob.run_trial(cfg, rng)
    stream = ob.glibc_stream(4, off + int(r1.draws) + 8)
    r2, u2 = ob.model_run_trial(cfg, ob.RNG_GLIBC, 4, stream, off, 16)
    assert r1.nSuccessUE == r2.nSuccessUE and r1.draws == r2.draws
    assert bytes(u1) == bytes(u2)


def test_oracle_census_counts_every_call(ob):
    """The design-study census of the oracle (oracle/census_gate.py, profiles/r04_lcluster_phases.md): its per-subframe call counts add up to the oracle's
    own collisionCalls, singleton calls are calls, and every UE that transmitted drew a preamble at least once."""
    import ctypes as C
    L = ob.lib()
    L.oracle_set_census.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.oracle_set_census.restype = None
    T = 10000
    red, sing, calls = (np.zeros(T, dtype=np.int32) for _ in range(3))
    L.oracle_set_census(red.ctypes.data, sing.ctypes.data, calls.ctypes.data, T)
    try:
        res, ues = ob.run_trial(ob.make_cfg(6000, variant=1), ob.Rng(ob.RNG_PHILOX, 3))
    finally:
        L.oracle_set_census(None, None, None, 0)
    assert int(calls.sum()) == int(res.as_dict()["collisionCalls"]) and (sing <= calls).all() and sing.sum() > 0
    u = np.frombuffer(ues, dtype=np.int32).reshape(-1, 16)
    assert int(red.sum()) >= int((u[:, 7] >= 0).sum())  # (column 7: preamble; -1 = never drew one)
