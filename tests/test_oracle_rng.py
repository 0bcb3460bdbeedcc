"""The oracle's (and the product host's) RNG restatements against their sources of truth:
glibc's own rand() through ctypes, and the published Philox4x32-10 known-answer vectors."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.parametrize("seed", [0, 1, 2, 7, 2022, 123456789])
def test_glibc_clone_matches_libc(ob, seed):
    libc = C.CDLL("libc.so.6")
    libc.srand(seed)
    n = 200_000
    ref = np.fromiter((libc.rand() for _ in range(n)), dtype=np.int64, count=n)
    mine = ob.glibc_stream(seed, n).astype(np.int64)
    assert (ref == mine).all()
    rng = ob.Rng(ob.RNG_GLIBC, seed)
    assert [rng.next_glibc() for _ in range(100)] == ref[:100].tolist()
    assert rng.consumed() == 100


def test_glibc_seed0_equals_seed1(ob):
    assert (ob.glibc_stream(0, 1000) == ob.glibc_stream(1, 1000)).all()


def test_product_glibc_stream_matches_libc_and_offsets(pkg):
    libc = C.CDLL("libc.so.6")
    libc.srand(5)
    ref = np.fromiter((libc.rand() for _ in range(50_000)), dtype=np.int64, count=50_000)
    assert (pkg.glibc_stream(5, 0, 50_000).astype(np.int64) == ref).all()
    assert (pkg.glibc_stream(5, 12_345, 1000).astype(np.int64) == ref[12_345:13_345]).all()


PHILOX_KAT = [  # Random123 kat_vectors, philox4x32-10: counter, key, expected
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,exp", PHILOX_KAT)
def test_philox_known_answers(ob, ctr, key, exp):
    out = (C.c_uint32 * 4)()
    ob.lib().oracle_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    assert tuple(out) == exp


def test_philox_draw_convention(ob):
    # draw k of UE = philox(ctr={ue,k,nUE,variant}, key=seed)[0] >> 1
    out = (C.c_uint32 * 4)()
    seed = (0x299f31d0 << 32) | 0xa4093822
    ob.lib().oracle_philox4x32_10((C.c_uint32 * 4)(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
                                  (C.c_uint32 * 2)(0xa4093822, 0x299f31d0), out)
    assert ob.lib().oracle_philox_draw31(seed, 0x13198a2e, 0x03707344, 0x243f6a88, 0x85a308d3) == out[0] >> 1


def test_msg4_success_threshold(ob):
    """Beta.c:374-375: (float)r/(float)RAND_MAX > 0.1  <=>  r >= 214748361."""
    f = np.float32
    for r, want in ((214748360, False), (214748361, True), (0, False), (2147483647, True)):
        assert bool(np.float64(f(r) / f(2147483647)) > 0.1) is want
