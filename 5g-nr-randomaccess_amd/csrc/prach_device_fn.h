// prach_device_fn.h — small __device__ helpers shared by the trial kernels (gfx950 only).
#pragma once
#include "prach_device.h"

// The per-UE state machine (prach_ue_body.h) and the arithmetic it rests on are host-callable as well: tests/tools/flat_equiv.hip runs the branched and the
// branch-free form side by side on the CPU (pytest -m "not gpu"), no GPU involved.
#define PRACH_HD __host__ __device__ __forceinline__

namespace prach {

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
PRACH_HD int slot_align(int sub, int aT) { // Beta.c:268-277
    const int m = sub % aT;
    return m == 0 ? sub + 1 : (m == 1 ? sub : sub + (aT - m + 1));
}
// x % d for a wave-uniform divisor d and 0 <= x < 2^32: one mul-hi + one correction instead of the
// ~40-instruction software division (M = floor(2^32 / d); the quotient estimate is at most 1 short)
struct FastMod { unsigned d, M; };
PRACH_HD unsigned mulhi32(const unsigned a, const unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); } // (v_mul_hi_u32; __umulhi is device-only)
PRACH_HD FastMod make_fastmod(int d) {
    FastMod f;
    f.d = (unsigned)d;
    f.M = d > 1 ? (unsigned)(0x100000000ull / (unsigned long long)d) : (d == 1 ? 0xFFFFFFFFu : 0u); // (d == 1: q = x - 1 for x > 0, so r = 1 and the correction gives 0 without a special case)
    return f;
}
PRACH_HD int fastmod(int x, const FastMod f) {
    if (f.d == 1u) return 0;
    const unsigned q = mulhi32((unsigned)x, f.M);
    const unsigned r = (unsigned)x - q * f.d;
    return (int)(r >= f.d ? r - f.d : r);
}
PRACH_HD int slot_align_fm(int sub, const FastMod aT) { // Beta.c:268-277
    const int m = fastmod(sub, aT);
    return m == 0 ? sub + 1 : (m == 1 ? sub : sub + ((int)aT.d - m + 1));
}
PRACH_HD int now_backoff(int bo, int t) { return bo > 0 ? max(bo - t, 0) : bo; }
PRACH_HD int enc_backoff(int X, int t) { return X > 0 ? t + X : X; }

// Philox4x32-10 (the draw = word 0 of the block >> 1).  Written for gfx950's instruction set: one v_mad_u64_u32 per 32 x 32 -> 64 product (the compiler
// emits a v_mul_hi_u32 + v_mul_lo_u32 pair for __umulhi and *), one v_bitop3_b32 per three-way xor — 6 vector instructions per round instead of 10.  The
// batched kernel is bound by instruction issue and the multiplies are its most expensive instructions (profiles/r04_grid.md, section 5).
#ifdef PRACH_NO_BITOP3 // (diagnostic builds that switch the instruction off: LABNOTES, round 4)
__device__ __forceinline__ unsigned xor3(const unsigned a, const unsigned b, const unsigned c) { return a ^ b ^ c; }
#else
__device__ __forceinline__ unsigned xor3(const unsigned a, const unsigned b, const unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
#endif
__device__ __forceinline__ int philox_draw31(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2,
                                             unsigned c3) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = xor3((unsigned)(p1 >> 32), c1, k0), n2 = xor3((unsigned)(p0 >> 32), c3, k1);
        c0 = n0; c1 = (unsigned)p1; c2 = n2; c3 = (unsigned)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (int)(c0 >> 1);
}
// two consecutive draws of one UE (counter words c1 and c1 + 1): the round keys are computed once
__device__ __forceinline__ void philox_draw31_x2(unsigned k0, unsigned k1, const unsigned c0_, const unsigned c1_, const unsigned c2_, const unsigned c3_, int &d1, int &d2) {
    unsigned a0 = c0_, a1 = c1_, a2 = c2_, a3 = c3_, b0 = c0_, b1 = c1_ + 1u, b2 = c2_, b3 = c3_;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long pa0 = (unsigned long long)0xD2511F53u * a0, pa1 = (unsigned long long)0xCD9E8D57u * a2;
        const unsigned long long pb0 = (unsigned long long)0xD2511F53u * b0, pb1 = (unsigned long long)0xCD9E8D57u * b2;
        const unsigned na0 = xor3((unsigned)(pa1 >> 32), a1, k0), na2 = xor3((unsigned)(pa0 >> 32), a3, k1);
        const unsigned nb0 = xor3((unsigned)(pb1 >> 32), b1, k0), nb2 = xor3((unsigned)(pb0 >> 32), b3, k1);
        a0 = na0; a1 = (unsigned)pa1; a2 = na2; a3 = (unsigned)pa0;
        b0 = nb0; b1 = (unsigned)pb1; b2 = nb2; b3 = (unsigned)pb0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    d1 = (int)(a0 >> 1); d2 = (int)(b0 >> 1);
}

// activateUEs (WithNOMA:393-410): theta from the first of the two activation draws fixes the UE's 60-degree sector (float arithmetic and
// comparisons with double constants only: the reference's value on any IEEE machine)
__device__ __forceinline__ int sector_of_draw(const int d) {
    const float pi = 3.14f;
    const float theta = (float)d / (float)2147483647 * 2 * pi;
    if (theta >= 0 && theta < ((1. / 3.) * pi)) return 0;
    if (theta >= ((1. / 3.) * pi) && theta < ((2. / 3.) * pi)) return 1;
    if (theta >= ((2. / 3.) * pi) && theta < 3.14) return 2;
    if (theta >= pi && theta < ((4. / 3.) * pi)) return 3;
    if (theta >= ((4. / 3.) * pi) && theta < ((5. / 3.) * pi)) return 4;
    return 5;
}

__device__ __forceinline__ unsigned long long lanemask_le(int lane) {
    return lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
}
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// Wavefront prefix sum / sum on the DPP network: six dependent VALU instructions (row_shr 1, 2, 4, 8 inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3) instead of six ds_bpermute round trips through the LDS pipeline
// (__shfl_up/__shfl_down: >= 64 cycles each plus an lgkmcnt wait) — these sit on the subframe's dependent chain.  Every lane of the
// wavefront must be active (a lane that is not contributes the `old` operand, 0, to its readers).
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dpp_add(const int x) {
    return x + __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ int wave_scan_incl(int x) {
    x = dpp_add<0x111, 0xf>(x); x = dpp_add<0x112, 0xf>(x); x = dpp_add<0x114, 0xf>(x); x = dpp_add<0x118, 0xf>(x);
    x = dpp_add<0x142, 0xa>(x); // row_bcast:15, rows 1 and 3
    x = dpp_add<0x143, 0xc>(x); // row_bcast:31, rows 2 and 3
    return x;
}
__device__ __forceinline__ int wave_sum(const int x) { return __builtin_amdgcn_readlane(wave_scan_incl(x), 63); } // wave-uniform
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dpp_max(const int x) {
    return max(x, __builtin_amdgcn_update_dpp(INT_MIN, x, CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ int wave_max(int x) { // wave-uniform maximum, same network
    x = dpp_max<0x111, 0xf>(x); x = dpp_max<0x112, 0xf>(x); x = dpp_max<0x114, 0xf>(x); x = dpp_max<0x118, 0xf>(x);
    x = dpp_max<0x142, 0xa>(x); x = dpp_max<0x143, 0xc>(x);
    return __builtin_amdgcn_readlane(x, 63);
}


// ---------------------------------------------------------------------------------------------
// Device memory is addressed in the GLOBAL address space, explicitly.  The per-trial pointers reach a kernel
// inside a parameter block loaded from memory, which the compiler can only treat as generic pointers: every
// access becomes a flat_* instruction (64-bit address arithmetic per lane, and it counts on lgkmcnt as well as
// vmcnt, so each LDS wait also drains the outstanding record loads — a software prefetch cannot overlap
// anything).  TrialG is the same block with address_space(1) pointers: global_load/store/atomic with a scalar
// base, and the record prefetch really stays in flight.
// ---------------------------------------------------------------------------------------------
#define PRACH_G __attribute__((address_space(1)))
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
static_assert(sizeof(prach_ue_log) == 64 && sizeof(Event) == 16, "records are stored as 16-byte vectors");

struct TrialG {
    int variant, uniform, nUE, nP, backoff, nGrantUL, maxRarWindow, maxMsg2, aT, rng_mode, maxTime, stop;
    unsigned seed_lo, seed_hi;
    unsigned long long stream_len;
    PRACH_G v4i_t *rec;
    PRACH_G int *ptc, *ftt, *stt, *fcnt;
    PRACH_G unsigned *nd;
    PRACH_G v4i_t *evbuf, *evbuf2; // Event records
    PRACH_G int *sidx;
    const PRACH_G int *sched;
    const PRACH_G int *stream;
    PRACH_G v4i_t *logs; // prach_ue_log = four 16-byte vectors; null: no per-UE log wanted
    PRACH_G int *timers;
    PRACH_G DevResult *out;
    int evw, mbstride, binshift;
    PRACH_G int *mbox;
    PRACH_G v2i_t *cand;
    int dense_pass, pipeline;
    const PRACH_G int *n_pre0, *n_sector;
    const PRACH_G double *n_gain, *n_lgain;
    const PRACH_G unsigned *n_nd0;
    int n_devact;
    int flags;
    PRACH_G int *sector;

    __device__ __forceinline__ explicit TrialG(const TrialDev &d)
        : variant(d.variant), uniform(d.uniform), nUE(d.nUE), nP(d.nP), backoff(d.backoff), nGrantUL(d.nGrantUL), maxRarWindow(d.maxRarWindow),
          maxMsg2(d.maxMsg2), aT(d.aT), rng_mode(d.rng_mode), maxTime(d.maxTime), stop(d.stop), seed_lo(d.seed_lo), seed_hi(d.seed_hi),
          stream_len(d.stream_len), rec((PRACH_G v4i_t *)d.rec), ptc((PRACH_G int *)d.ptc), ftt((PRACH_G int *)d.ftt), stt((PRACH_G int *)d.stt),
          fcnt((PRACH_G int *)d.fcnt), nd((PRACH_G unsigned *)d.nd), evbuf((PRACH_G v4i_t *)d.evbuf), evbuf2((PRACH_G v4i_t *)d.evbuf2),
          sidx((PRACH_G int *)d.sidx), sched((const PRACH_G int *)d.sched), stream((const PRACH_G int *)d.stream), logs((PRACH_G v4i_t *)d.logs),
          timers((PRACH_G int *)d.timers), out((PRACH_G DevResult *)d.out), evw(d.evw), mbstride(d.mbstride), binshift(d.binshift),
          mbox((PRACH_G int *)d.mbox), cand((PRACH_G v2i_t *)d.cand), dense_pass(d.dense_pass), pipeline(d.pipeline),
          n_pre0((const PRACH_G int *)d.n_pre0), n_sector((const PRACH_G int *)d.n_sector), n_gain((const PRACH_G double *)d.n_gain),
          n_lgain((const PRACH_G double *)d.n_lgain), n_nd0((const PRACH_G unsigned *)d.n_nd0), n_devact(d.n_devact), flags(d.flags), sector((PRACH_G int *)d.sector) {}
};

// The hot record is read with a non-temporal 16-byte load (global_load_dwordx4 ... nt: served by L2, never
// by this CU's L1): the resolver sets the grant bit with an L2 atomic, which a stale L1 line would hide.
__device__ __forceinline__ int4 load_rec(const PRACH_G v4i_t *p) {
    const v4i_t v = __builtin_nontemporal_load(p);
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int4 load_rec_plain(const PRACH_G v4i_t *p) {
    const v4i_t v = *p;
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_rec(PRACH_G v4i_t *p, const int4 r) {
    v4i_t v;
    v.x = r.x; v.y = r.y; v.z = r.z; v.w = r.w;
    *p = v;
}
__device__ __forceinline__ void grant_rec(PRACH_G v4i_t *p) { // one fire-and-forget L2 atomic on the packed word
    __hip_atomic_fetch_or(reinterpret_cast<PRACH_G unsigned *>(p) + 3, PK_GRANT_BIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_i2(PRACH_G v2i_t *p, const int a, const int b) {
    v2i_t v;
    v.x = a; v.y = b;
    *p = v;
}
__device__ __forceinline__ void store_log(PRACH_G v4i_t *logs, const int i, const prach_ue_log &o) {
    const int *w = reinterpret_cast<const int *>(&o);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v4i_t v;
        v.x = w[4 * k]; v.y = w[4 * k + 1]; v.z = w[4 * k + 2]; v.w = w[4 * k + 3];
        logs[(size_t)i * 4 + k] = v;
    }
}
// agent-scope relaxed read-modify-writes on global memory (results of a trial)
// ---- phase A's classification of one 64-UE group, as lane masks ----------------------------------------------------------
// Every predicate is ONE vector compare that writes a 64-bit lane mask; the boolean algebra on them runs on the scalar unit
// (written as `bool`s the compiler kept turning masks into 0/1 vector values and back across the merges: ~10 of the ~30 vector
// instructions of a group visit, in a pass that is bound by vector issue).  All 64 lanes must be active.
//   light: PEND_NONE / PEND_STAY / PEND_CALLER without a grant (pg 0..2), contending, RAR window stays open after this subframe
//          (a PEND_STAY record may be `age` subframes old — written at subframe rx, not touched since: rarWindow has grown by age);
//          SPEC (ahead of the previous subframe's resolver): a new caller may still get a grant and waits for phase B
//   quiet: nothing to do;  done: finished for good;  trig: txTime == t
struct PassMasks { unsigned long long light, quiet, done, trig; };
template <bool SPEC>
__device__ __forceinline__ PassMasks pass_masks(const unsigned pk, const int rx, const int rz, const int t, const unsigned rarlim, const bool front,
                                                const int i, const int acNow, const int acPrev, const int nUE) {
    typedef unsigned long long u64;
    const unsigned top = pk & 0xF0000000u; // deferred outcome | grant bit
    const u64 mM1 = __ballot((pk & 3u) == (unsigned)ACT_M1), mPre = __ballot((pk & (0xffu << PK_PRE_SHIFT)) != 0u);
    const u64 mBo = __ballot(rz <= t); // nowBackoff <= 0: stored as expiry subframe when positive
    const u64 mTrig = __ballot(rx == t);
    const u64 mPg0 = __ballot(pk < (1u << PK_PEND_SHIFT)), mAct2 = __ballot((pk & 2u) != 0u); // ACT_M1 or ACT_M3
    u64 mDone = __ballot((pk & 3u) == (unsigned)ACT_DONE);
    const u64 mCont = mM1 & mPre & mBo;
    const unsigned age = top == ((unsigned)PEND_STAY << PK_PEND_SHIFT) ? (unsigned)(t - 1 - rx) : 0u;
    const unsigned rarnow = (pk & (0xffu << PK_RAR_SHIFT)) + (age << PK_RAR_SHIFT);
    u64 mLight = __ballot(pk < (3u << PK_PEND_SHIFT)) & mCont & __ballot(rarnow < rarlim);
    if (SPEC) mLight &= ~__ballot(top == ((unsigned)PEND_CALLER << PK_PEND_SHIFT));
    u64 mQuiet = mPg0 & (~mAct2 | (~mCont & ~mTrig & ~(mM1 & ~mPre)));
    if (front) { // the (at most two) groups the arrival front is in: per-lane range checks
        const u64 mValid = __ballot(i < acNow), mOld = __ballot(i < acPrev);
        mLight &= mOld;
        mQuiet = ~mValid | (mOld & mQuiet);
        mDone = (mValid & mDone) | __ballot(i >= nUE);
    }
    return PassMasks{mLight, mQuiet, mDone, mTrig};
}

template <class T> __device__ __forceinline__ void gadd(PRACH_G T *p, const T v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void gmin(PRACH_G T *p, const T v) { __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void gmax(PRACH_G T *p, const T v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// decode / encode of the packed word of the hot record (prach_device.h)
struct UeState {
    int tx, tb, bo, act, conn, pre, rar, mrc, pend;
};
PRACH_HD UeState unpack(const int4 r) {
    UeState u;
    u.tx = r.x; u.tb = r.y; u.bo = r.z;
    u.act = (r.w >> PK_ACT_SHIFT) & 3; u.conn = (r.w >> PK_CONN_SHIFT) & 3; u.pre = (r.w >> PK_PRE_SHIFT) & 0xff;
    u.rar = (r.w >> PK_RAR_SHIFT) & 0xff; u.mrc = (r.w >> PK_MRC_SHIFT) & 0xff; u.pend = (r.w >> PK_PEND_SHIFT) & 7;
    return u;
}
PRACH_HD int4 pack(const UeState &u) {
    return make_int4(u.tx, u.tb, u.bo,
                     (u.act << PK_ACT_SHIFT) | (u.conn << PK_CONN_SHIFT) | (u.pre << PK_PRE_SHIFT) | (u.rar << PK_RAR_SHIFT) |
                         (u.mrc << PK_MRC_SHIFT) | (u.pend << PK_PEND_SHIFT));
}

} // namespace prach
