// prach_device_fn.h — small __device__ helpers shared by the trial kernels (gfx950 only).
#pragma once
#include "prach_device.h"

namespace prach {

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int slot_align(int sub, int aT) { // Beta.c:268-277
    const int m = sub % aT;
    return m == 0 ? sub + 1 : (m == 1 ? sub : sub + (aT - m + 1));
}
// x % d for a wave-uniform divisor d and 0 <= x < 2^32: one mul-hi + one correction instead of the
// ~40-instruction software division (M = floor(2^32 / d); the quotient estimate is at most 1 short)
struct FastMod { unsigned d, M; };
__device__ __forceinline__ FastMod make_fastmod(int d) {
    FastMod f;
    f.d = (unsigned)d;
    f.M = d > 1 ? (unsigned)(0x100000000ull / (unsigned long long)d) : 0u;
    return f;
}
__device__ __forceinline__ int fastmod(int x, const FastMod f) {
    if (f.d == 1u) return 0;
    const unsigned q = __umulhi((unsigned)x, f.M);
    const unsigned r = (unsigned)x - q * f.d;
    return (int)(r >= f.d ? r - f.d : r);
}
__device__ __forceinline__ int slot_align_fm(int sub, const FastMod aT) { // Beta.c:268-277
    const int m = fastmod(sub, aT);
    return m == 0 ? sub + 1 : (m == 1 ? sub : sub + ((int)aT.d - m + 1));
}
__device__ __forceinline__ int now_backoff(int bo, int t) { return bo > 0 ? max(bo - t, 0) : bo; }
__device__ __forceinline__ int enc_backoff(int X, int t) { return X > 0 ? t + X : X; }

__device__ __forceinline__ int philox_draw31(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2,
                                             unsigned c3) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (int)(c0 >> 1);
}

__device__ __forceinline__ unsigned long long lanemask_le(int lane) {
    return lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
}
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }


// The hot record is read with a non-temporal 16-byte load (global_load_dwordx4 ... nt: served by L2, never
// by this CU's L1): the resolver sets the grant bit with an L2 atomic, which a stale L1 line would hide.
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 load_rec(const int4 *p) {
    const v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t *>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}

// decode / encode of the packed word of the hot record (prach_device.h)
struct UeState {
    int tx, tb, bo, act, conn, pre, rar, mrc, pend;
};
__device__ __forceinline__ UeState unpack(const int4 r) {
    UeState u;
    u.tx = r.x; u.tb = r.y; u.bo = r.z;
    u.act = (r.w >> PK_ACT_SHIFT) & 3; u.conn = (r.w >> PK_CONN_SHIFT) & 3; u.pre = (r.w >> PK_PRE_SHIFT) & 0xff;
    u.rar = (r.w >> PK_RAR_SHIFT) & 0xff; u.mrc = (r.w >> PK_MRC_SHIFT) & 0xff; u.pend = (r.w >> PK_PEND_SHIFT) & 7;
    return u;
}
__device__ __forceinline__ int4 pack(const UeState &u) {
    return make_int4(u.tx, u.tb, u.bo,
                     (u.act << PK_ACT_SHIFT) | (u.conn << PK_CONN_SHIFT) | (u.pre << PK_PRE_SHIFT) | (u.rar << PK_RAR_SHIFT) |
                         (u.mrc << PK_MRC_SHIFT) | (u.pend << PK_PEND_SHIFT));
}

} // namespace prach
