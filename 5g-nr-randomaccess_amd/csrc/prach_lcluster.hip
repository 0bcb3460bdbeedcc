// prach_lcluster.hip — the HEADLINE kernel: one Philox trial on a cluster of G workgroups whose UE state is resident in LDS.
//
// Same parallel-exact decomposition as prach_cluster.hip (static interleaved ownership of 64-UE groups, one
// granule exchange per subframe, set-wise ranks, compacted two-phase pass, phase A of subframe t+1 inside the exchange of
// subframe t) — and the same results, bit for bit (tests/test_gpu_parity.py runs both against the oracle).  What differs is
// how it is written.  A subframe of a single trial is a CHAIN of short dependent phases in which every wavefront executes a
// few hundred instructions; one wavefront issues a vector instruction every 4-8 cycles (MI355X_MICROARCH.md, "vector-
// instruction ISSUE cost"), so the subframe time is the number of instructions on that chain, not bytes and not lanes.  The
// general kernel carried ~250 spilled scalar registers through that chain (v_readlane / v_writelane on every phase boundary),
// 64-bit mailbox address arithmetic per granule, run-time LDS table offsets and loops around single-trip work.  Here:
//   * every LDS table sits at a COMPILE-TIME offset (bucket tables have the fixed stride NPCL = 64: nPreamble <= 64, the
//     reference's 54 / 64); an LDS address is an instruction immediate, not a live register;
//   * each thread's role in the exchange (which granules it loads, which bucket it publishes) is computed ONCE before the
//     step loop and kept in vector registers as 32-bit offsets from the mailbox base;
//   * the hot record (16 B) and the Philox draw index of every owned UE live in LDS for the whole trial (no L2 round trip
//     for a record, a draw index or an early-leaver candidate anywhere in the step loop); finished groups are a per-wavefront
//     register bit mask; subframe bookkeeping (t mod 5, t mod accessTime, the arrival table) is incremental scalar state;
//   * wavefront prefix sums / reductions run on the DPP network (wave_scan_incl, prach_device_fn.h), not through ds_bpermute round trips;
//   * launched XCD-packed (a cluster = the blocks of equal blockIdx % 8), a cluster that has VERIFIED one common XCD keeps its exchange
//     granules in that XCD's L2 (handshake in the prologue; any other placement keeps the write-through granules);
//   * the last wavefront is the exchange's (bucket publish, headers, event offsets) and walks no UE groups; the header leaves right
//     behind S1 when there is no early-leaver candidate (no barrier S2);
//   * only the Philox path, only clusters (G > 1), only the pipelined compacted pass: the engine falls back to
//     prach::cluster_kernel for everything else (glibc streams, one workgroup per trial, nPreamble > 64, more owned UEs than
//     LDS holds, diagnostic options).
// Reference semantics: RandomAccessSimulatorBeta.c:111-197 / RandomAccessWithNOMA.c:267-351.
#include "prach_device.h"
#include "prach_device_fn.h"
#include "prach_ue_body.h"
#include <limits.h>

namespace prach {

namespace {

#ifdef PRACH_STAMPS
#ifndef PRACH_STAMP_TID
#define PRACH_STAMP_TID 0 // the thread whose clock is read (-DPRACH_STAMP_TID=960: the wavefront that reads the headers)
#endif
// (accumulated in LDS behind the UE state, not in registers: 24 64-bit accumulators in registers made the diagnostic build spill the
//  exchange's address registers to scratch and serialise its loads — a profile of the profiler)
#define LSTAMP(k)                                                                                      \
    do {                                                                                               \
        if (threadIdx.x == PRACH_STAMP_TID) { const unsigned long long now_ = __builtin_readcyclecounter(); fstamps[k] += now_ - fprev; fprev = now_; } \
    } while (0)
static_assert(4 * RCCAP >= 8 * 28, "diagnostic accumulators");
#else
#define LSTAMP(k) do { } while (0)
#endif

constexpr int NPCL = 64;     // stride of the per-bucket tables (nPreamble <= 64)
// Wavefronts that walk the UE groups in phase A.  The last one does not: it publishes the bucket granules, reads the headers and builds
// the event offsets — the chain every other wavefront waits for at S3 (measured: with three group visits of its own in the exchange
// window it reached S3 ~1 000 cycles after everybody else).
constexpr int NWA = NW - 1;
constexpr int LEV = 4096;    // gathered events per subframe
constexpr int LSC = 2048;    // singleton callers per subframe
#ifdef PRACH_QCAP
constexpr int LCC = 8;       // (test build: the candidate list overflows in ordinary trials, so that the path below is exercised)
#else
constexpr int LCC = 1024;    // early-leaver candidates per workgroup and subframe
#endif
constexpr int LQ = CLUSTER_LQCAP; // event queue = at most every owned UE slot
constexpr int LGB = 1024;    // grant selection bins
constexpr int LEPF = 32;     // event granules of every mailbox fetched together with the bucket granules (more: a second round)
constexpr int LRQ = 1024;    // UEs per subframe whose next two Philox draws are recomputed ahead (more: drawn in place)
constexpr unsigned ND_READY = 0x80000000u; // lnd[slot] bit 31: ldraw[slot] holds draws nd, nd + 1 of this UE
constexpr unsigned LSPIN = 1u << 22;
constexpr int EVL_CALLER = UEV_CALLER, EVL_RESETCAND = UEV_RESETCAND, EVL_RJOIN = UEV_RJOIN, EVL_LEAVER = 4; // (= prach_cluster.hip's EVC_*)

// scalars in LDS
enum { S_NSUCC = 0, S_COLL, S_TXOP, S_CONTF, S_NS, S_NRC, S_NRJ, S_NEV = 8, S_NCAND /* = S_NEV + 1: read as a pair; both [2] by subframe parity (+ 2): slots 8..11 */, S_PTC = 13, S_FC, S_SUMT = 16,
       S_ND = 18, S_NCROSS = 20, S_NRQ = 23, S_SX = 24,
       S_QN = 26, // [2] event queue length, by subframe parity: phase A of subframe t+1 fills one while phase B of subframe t has just read the other
       // what every thread needs behind S3, in ONE 16-byte LDS read: status, gathered events | overflow flag << 30, events beyond round 1,
       // successes of the whole cluster
       S_STATUS = 28, S_NTOT = 29, S_NREM = 30, S_NSUCCTOT = 31 };

constexpr int SCHR = 32; // arrival-table ring (power of two; lives in the upper half of the 64 LDS scalars)
// ---- LDS layout: byte offsets, all compile-time -------------------------------------------------------------------
namespace lo {
constexpr int GEV = 0;                          // int2 [LEV] gathered events
constexpr int SIDX = GEV + 8 * LEV;             // int [LSC]
constexpr int RCL = SIDX + 4 * LSC;             // int [RCCAP]
constexpr int SCAL = RCL + 4 * RCCAP;           // int [64]
constexpr int EVOFF = SCAL + 4 * 64;            // int [64 + 16]
constexpr int SCHED = SCAL + 4 * 32;            // int [SCHR] ring over the arrival table (activeCheck of the next SCHR / 2 .. SCHR access slots): the upper half of SCAL
constexpr int BINS = EVOFF + 4 * 80;            // int [LGB]
constexpr int WTOT = BINS + 4 * LGB;            // int [NW]
constexpr int PAR = WTOT + 4 * NW;              // per subframe parity: HIST, MLOC, MLOCS, CANDN, each [NPCL]
constexpr int PARSZ = 4 * 4 * NPCL;
constexpr int P_HIST = 0, P_MLOC = 4 * NPCL, P_MLOCS = 8 * NPCL, P_CANDN = 12 * NPCL;
constexpr int RB = PAR + 2 * PARSZ;             // per subframe parity, the resolver's tables: FCALL, LCALL, TOTAL, NLV, FIE, each [NPCL]
constexpr int RBSZ = 5 * 4 * NPCL;
constexpr int FCALL = RB, LCALL = RB + 4 * NPCL, TOTAL = RB + 8 * NPCL, NLV = RB + 12 * NPCL, FIE = RB + 16 * NPCL; // (+ parity * RBSZ)
constexpr int QUEUE = RB + 2 * RBSZ;            // int [LQ] slots of the UEs that have an event in this subframe
constexpr int LCAND = QUEUE + 4 * LQ;           // int2 [LCC]
constexpr int RQ = BINS;                        // int [LRQ] refill list (phase B .. round 1; BINS is only used by the grant selection)
constexpr int TAIL = LCAND + 8 * LCC;           // int4 lrec[lslots]; int2 ldraw[lslots]; unsigned lnd[lslots]
static_assert(LRQ <= LGB, "the refill list shares the grant bins");
static_assert(LQ <= 4 * WG_THREADS, "at most four queue batches per wavefront (the glibc mode remembers its slots in four registers)");
static_assert(SIDX % 16 == 0 && TAIL % 16 == 0 && LCAND % 8 == 0, "alignment");
} // namespace lo

#define LI(off) (reinterpret_cast<int *>(smem + (off)))
#define LU(off) (reinterpret_cast<unsigned *>(smem + (off)))
#define LI2(off) (reinterpret_cast<int2 *>(smem + (off)))

// ---- exchange granules (as in prach_cluster.hip): ONE naturally aligned 8-byte write-through store {20-bit value | tag[11:0]}
// {20-bit value | tag[15:12]}; the consumer re-reads until the tag (subframe + 1) matches: no flag, no fence (guide G16 / R2)
constexpr unsigned GRL_NONE = 0xFFFFFu;
__device__ __forceinline__ long long lmk(unsigned lo20, unsigned hi20, unsigned tag) {
    const unsigned w0 = (lo20 & 0xFFFFFu) | ((tag & 0xFFFu) << 20), w1 = (hi20 & 0xFFFFFu) | (((tag >> 12) & 0xFu) << 20);
    return (long long)(((unsigned long long)w1 << 32) | w0);
}
__device__ __forceinline__ bool lok(long long g, unsigned tag) {
    const unsigned w0 = (unsigned)g, w1 = (unsigned)((unsigned long long)g >> 32);
    return (w0 >> 20) == (tag & 0xFFFu) && ((w1 >> 20) & 0xFu) == ((tag >> 12) & 0xFu);
}
__device__ __forceinline__ long long lld(const PRACH_G long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lst(PRACH_G long long *p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A cluster whose workgroups have VERIFIED (handshake before the step loop) that they all run on one XCD shares that XCD's L2: its
// granule stores may then stay in L2 (workgroup-scope store: no write-through to the fabric), where the peers' `sc1` loads — which
// bypass only the per-CU L1 — find them after an L2 round trip instead of a fabric one (guide: `sc1` stores DROP the line from the
// XCD's L2, plain / `sc0` stores KEEP it).  Any other placement keeps the write-through stores.
__device__ __forceinline__ void lstx(const bool same_xcd, PRACH_G long long *p, long long v) {
    if (same_xcd) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long lwait(const PRACH_G long long *p, unsigned tag, char *smem) {
    long long g = lld(p);
    unsigned spins = 0;
    while (!lok(g, tag)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > LSPIN) { LI(lo::SCAL)[S_STATUS] = PRACH_ERR_TIMEOUT; break; } // peer not resident? the engine reruns the trial
        g = lld(p);
    }
    return g;
}

// per-trial constants of the step loop (wave-uniform)
struct LK {
    int nUE, nP, aT, maxRar, maxMsg2, variant, b, G;
    bool withnoma, sx;
    unsigned seed_lo, seed_hi, rarlim;
    FastMod fmP, fmB, fmA, fm5;
    PRACH_G int *ptc, *ftt, *stt, *fcnt;
};

__device__ __forceinline__ int l_idx_of(const LK &K, const int slot) { return (K.b + K.G * (slot >> 6)) * 64 + (slot & 63); }

// contending with a RAR window that stays open (prach_cluster.hip light_case)
__device__ __forceinline__ bool l_light(const unsigned pk, const int rx, const int rz, const int t, const unsigned rarlim) {
    const unsigned pg = pk >> PK_PEND_SHIFT;
    const bool contend = (pk & 3u) == (unsigned)ACT_M1 && (pk & (0xffu << PK_PRE_SHIFT)) != 0u && rz <= t;
    const int age = pg == (unsigned)PEND_STAY ? t - 1 - rx : 0;
    const unsigned rarnow = (pk & (0xffu << PK_RAR_SHIFT)) + ((unsigned)age << PK_RAR_SHIFT);
    return pg < 3u && contend && rarnow < rarlim;
}

// ---------------------------------------------------------------------------------------------------------------------
// The full per-UE body for one queued UE per lane (prach_cluster.hip ue_step<0>: deferred outcome of subframe t-1, activation,
// selectPreamble / requestResourceAllocation on own state, bucket bookkeeping, special events).  pc: byte offset of this
// subframe's parity block; fb: byte offset of the previous subframe's first / last caller tables inside FCALL / LCALL.
// ---------------------------------------------------------------------------------------------------------------------
// MODE 0: Philox, fused (apply + activate + select).  The reference's own rand() stream (GLIBC) splits it as prach_cluster.hip does: MODE 1 apply +
// activate + this UE's rand() calls into its group's total and two per-group lane masks (the count follows from the pre-step state);
// MODE 2, behind the exchange of the counts: select with the draws at stream[sbase + gpre[group] + draws of the lower lanes of the group].
struct LG { int *gsum; const int *gpre; unsigned *gmask; const PRACH_G int *stream; unsigned long long sbase; }; // (GLIBC only)
template <int MODE>
__device__ __forceinline__ void l_step(char *smem, const LK &K, int4 *lrec, unsigned *lnd, const int2 *ldraw, const LG &GL, const int pc, const int fb, const int t, const int prevAC,
                                       PRACH_G long long *mbev, const unsigned tag, const int lane, const int i, const int slot, const bool valid, const int4 r,
                                       unsigned ndc, int &c_succ, int &c_contf) {
    constexpr bool COUNT = MODE == 1, SELECT = MODE == 2;
    const int nUE = K.nUE;
    const int tp = t - 1;
    const int tmod = fastmod(t, K.fmA);
    int *const hist = LI(lo::PAR + pc + lo::P_HIST), *const mloc = LI(lo::PAR + pc + lo::P_MLOC), *const candn = LI(lo::PAR + pc + lo::P_CANDN);
    const UeK UK{K.maxRar, K.maxMsg2, K.aT, K.withnoma, K.fmP, K.fmB, K.fmA, K.fm5};
    ColdGlobal cold{K.ptc, K.ftt, K.stt, K.fcnt};
    bool nd_dirty = false, dirty = false;
    bool rdy = (ndc & ND_READY) != 0u;
    ndc &= ~ND_READY;
    UeState u = unpack(r);

    // ---- deferred outcome of subframe t-1 (prach_ue_body.h ue_apply) ----
    if (!SELECT && u.pend != PEND_NONE) {
        if (u.pend == PEND_STAY) { u.rar += tp - u.tx; u.tx = tp; } // (the record dates from subframe u.tx: compact phase A)
        dirty = ue_apply(u, ((unsigned)r.w & PK_GRANT_BIT) != 0u, i, tp, K.fmA, CallTables{LI(lo::FCALL + fb), LI(lo::LCALL + fb)});
    }
    // ---- activation (Beta.c:136-146; activateUEs' two draws, WithNOMA:393-394, are never looked at: a WithNOMA UE's draw index STARTS at 2) ----
    if (!SELECT && valid && i >= prevAC) { ue_activate(u, i, t, cold); dirty = true; }
    const UePlan pl = ue_plan(u, t, K.maxRar, K.maxMsg2);
    const int need = pl.need;
    const int jl = slot >> 6, ln = slot & 63; // this workgroup's local group of the UE, its lane in the group
    if (COUNT) {
        if (need >= 1) {
            atomicAdd(&GL.gsum[i >> 6], need);
            atomicOr(&GL.gmask[4 * jl + (ln >> 5)], 1u << (ln & 31));
            if (need == 2) atomicOr(&GL.gmask[4 * jl + 2 + (ln >> 5)], 1u << (ln & 31));
        }
        if (dirty) lrec[slot] = pack(u);
        return;
    }

    // The next two draws of a UE (Philox counters nd, nd + 1) were computed AHEAD, off the subframe's critical chain (refill in
    // the exchange window), and wait in LDS; a UE whose refill did not fit the list draws in place.
    int d1 = 0, d2 = 0;
    if (SELECT) { // the reference's own stream: the UE's position inside its group from the two lane masks
        if (need > 0) {
            const unsigned lo_ = ln < 32 ? (1u << ln) - 1u : 0xffffffffu, hi_ = ln < 32 ? 0u : (1u << (ln - 32)) - 1u;
            const int before = __popc(GL.gmask[4 * jl] & lo_) + __popc(GL.gmask[4 * jl + 1] & hi_) + __popc(GL.gmask[4 * jl + 2] & lo_) + __popc(GL.gmask[4 * jl + 3] & hi_);
            const unsigned long long o = GL.sbase + (unsigned long long)GL.gpre[i >> 6] + (unsigned long long)before;
            d1 = GL.stream[o];
            if (need > 1) d2 = GL.stream[o + 1];
        }
    } else if (__any(need > 0)) {
        const unsigned k = ndc;
        if (need > 0 && rdy) { const int2 dd = ldraw[slot]; d1 = dd.x; d2 = dd.y; }
        if (__any(need > 0 && !rdy)) {
            const int e1 = philox_draw31(K.seed_lo, K.seed_hi, (unsigned)i, k, (unsigned)nUE, (unsigned)K.variant);
            const int e2 = philox_draw31(K.seed_lo, K.seed_hi, (unsigned)i, k + 1, (unsigned)nUE, (unsigned)K.variant);
            if (!rdy) { d1 = e1; d2 = e2; }
        }
        if (need > 0) { ndc = k + (unsigned)need; nd_dirty = true; rdy = false; } // (its refill is listed below)
    }

    // ---- selectPreamble / requestResourceAllocation on own state (prach_ue_body.h ue_select) ----
    const UeOut o = ue_select(u, pl, d1, d2, i, t, tmod, UK, cold, c_succ, c_contf);
    dirty = dirty || o.dirty;
    const int oldp = o.oldp, evtype = o.evtype, evp = o.evp;
    const bool member_pre = o.member_pre, eclass = o.eclass;

    // ---- bucket bookkeeping (workgroup-level LDS atomics) ----
    if (member_pre) atomicAdd(&hist[oldp], 1);
    if (u.pend == PEND_STAY) { if (__atomic_load_n(&mloc[oldp], __ATOMIC_RELAXED) > i) atomicMin(&mloc[oldp], i); }
    if (evtype == EVL_CALLER) atomicMin(&mloc[evp], i);
    {
        // special events -> this workgroup's mailbox, early-leaver candidates and refills -> their lists: the three list positions are
        // taken by lane 0 with three returning LDS atomics issued back to back — ONE wait instead of three dependent round trips
        const unsigned long long em = __ballot(evtype != 0), cm = __ballot(eclass), rm = SELECT ? 0ull : __ballot(need > 0);
        if (em | cm | rm) {
            int b_ev = 0, b_cd = 0, b_rq = 0;
            if (lane == 0) {
                if (em) b_ev = atomicAdd(&LI(lo::SCAL)[S_NEV + (pc ? 2 : 0)], __popcll(em));
                if (cm) b_cd = atomicAdd(&LI(lo::SCAL)[S_NCAND + (pc ? 2 : 0)], __popcll(cm));
                if (rm) b_rq = atomicAdd(&LI(lo::SCAL)[S_NRQ], __popcll(rm));
            }
            b_ev = __builtin_amdgcn_readfirstlane(b_ev); b_cd = __builtin_amdgcn_readfirstlane(b_cd); b_rq = __builtin_amdgcn_readfirstlane(b_rq);
            if (evtype != 0) {
                const int es = b_ev + __popcll(em & lanemask_lt(lane));
                const int info = ue_event_info(o);
                if (es < CLUSTER_EVW) lstx(K.sx, mbev + es, lmk((unsigned)i, (unsigned)info, tag));
            }
            if (eclass) {
                const int cs = b_cd + __popcll(cm & lanemask_lt(lane));
                if (cs < LCC) LI2(lo::LCAND)[cs] = make_int2(i, oldp);
                else LI(lo::SCAL)[S_STATUS] = PRACH_ERR_INTERNAL; // (engine: exact rerun on the general kernels)
                atomicAdd(&candn[oldp], 1);
            }
            if (!SELECT && need > 0) {
                const int rs = b_rq + __popcll(rm & lanemask_lt(lane));
                if (rs < LRQ) LI(lo::RQ)[rs] = slot;
            }
        }
    }
    if (!SELECT && nd_dirty) lnd[slot] = ndc | (rdy ? ND_READY : 0u);
    if (dirty) lrec[slot] = pack(u);
}

// One gathered event against the lowest DEFINITE caller of every bucket (complete after round 1).
__device__ __forceinline__ void l_classify(char *smem, const int fa, const int k, const int2 ev) {
    const int *const fcallA = LI(lo::FCALL + fa);
    const int type = ev.y & 7, p = (ev.y >> 4) & 0xff;
    if (type == EVL_RESETCAND) {
        if (fcallA[(ev.y >> 12) & 0xff] < ev.x) LI2(lo::GEV)[k].y = 0; // bumped on its old bucket before its turn: cannot re-join
        else { const int s = atomicAdd(&LI(lo::SCAL)[S_NRC], 1); if (s < RCCAP) LI(lo::RCL)[s] = k; }
    } else if (type == EVL_RJOIN) {
        atomicAdd(&LI(lo::SCAL)[S_NRJ], 1);
    } else if (type == EVL_LEAVER) {
        if (ev.x < fcallA[p]) atomicAdd(&LI(lo::NLV + fa)[p], 1);
    } else if (type == EVL_CALLER) {
        if (ev.x == fcallA[p]) LI(lo::FIE + fa)[p] = 1;
    }
}

// Reset-cycle re-join candidates, strictly in index order, by ONE wavefront with the first-caller table in registers
// (lane = bucket; nPreamble <= 64).  prach_cluster.hip resolve_reset_candidates.
__device__ __forceinline__ void l_resolve_reset_candidates(char *smem, const int fa, const int nrc_in, const int nP) {
    const int lane = threadIdx.x & 63;
    const int n = __builtin_amdgcn_readfirstlane(nrc_in);
    int *const fcall = LI(lo::FCALL + fa);
    int *const rcl = LI(lo::RCL), *const sidx = LI(lo::SIDX);
    int2 *const gev = LI2(lo::GEV);
    int f0 = lane < nP ? fcall[lane] : INT_MAX;
    for (int c = lane; c < n; c += 64) { // rank-sort the candidate list by UE index into SIDX (free at this point of the subframe)
        const int myidx = gev[rcl[c]].x;
        int rank = 0;
        for (int j = 0; j < n; j++) rank += gev[rcl[j]].x < myidx ? 1 : 0;
        sidx[rank] = rcl[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int base = 0; base < n; base += 64) {
        const int m = min(64, n - base);
        int es = 0, cidx = 0, cinfo = 0;
        if (lane < m) { es = sidx[base + lane]; const int2 e = gev[es]; cidx = e.x; cinfo = e.y; }
        int cancelled = 0;
        for (int s_ = 0; s_ < m; s_++) {
            const int idx = __builtin_amdgcn_readlane(cidx, s_), info = __builtin_amdgcn_readlane(cinfo, s_);
            const int p = (info >> 4) & 0xff, q = (info >> 12) & 0xff;
            if (__builtin_amdgcn_readlane(f0, q & 63) < idx) { // bumped before its turn: does not re-join
                if (lane == s_) cancelled = 1;
            } else if (idx < __builtin_amdgcn_readlane(f0, p & 63)) { // its call becomes the first one on p
                if (lane == (p & 63)) f0 = idx;
            }
        }
        if (lane < m && cancelled) gev[es].y = 0;
    }
    if (lane < nP) fcall[lane] = f0;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// GLIBC: the reference's own rand() stream (window PD->stream, generated on the device before this launch); gcap = groups the LDS count tables hold
template <bool GLIBC>
__global__ __launch_bounds__(WG_THREADS) void lcluster_kernel(const TrialDev *__restrict__ params, const int G, const int lslots, const int xpack, const int ntrials, const int gcap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int T = blockIdx.x / G, b = blockIdx.x % G; // a cluster = consecutive blocks (in-order dispatch completes whole clusters)
    if (xpack) {
        // XCD-packed launch: blocks bx and bx + 8 are dealt to the same XCD (observed round-robin dispatch, for speed only — the
        // handshake below checks it), so a cluster is made of the blocks of equal bx % 8 of one chunk of 8 G blocks, and eight
        // clusters — one per XCD — share a chunk.  Blocks of trials past the last one leave at once.
        const int chunk = blockIdx.x / (8 * G), within = blockIdx.x % (8 * G);
        T = chunk * 8 + (within & 7); b = within >> 3;
        if (T >= ntrials) return;
    }
    const TrialDev *const PD = params + T;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    LK K;
    K.nUE = PD->nUE; K.nP = PD->nP; K.aT = PD->aT; K.maxRar = PD->maxRarWindow; K.maxMsg2 = PD->maxMsg2; K.variant = PD->variant;
    K.b = b; K.G = G; K.sx = false;
    K.withnoma = K.variant == PRACH_VARIANT_WITHNOMA_C;
    K.seed_lo = PD->seed_lo; K.seed_hi = PD->seed_hi;
    K.rarlim = (unsigned)(K.maxRar - 1) << PK_RAR_SHIFT; // window still open after this subframe: rar + 1 < maxRarWindow
    K.fmP = make_fastmod(K.nP); K.fmB = make_fastmod(PD->backoff); K.fmA = make_fastmod(K.aT); K.fm5 = make_fastmod(5);
    K.ptc = (PRACH_G int *)PD->ptc; K.ftt = (PRACH_G int *)PD->ftt; K.stt = (PRACH_G int *)PD->stt; K.fcnt = (PRACH_G int *)PD->fcnt;
    // The event body's constants (divisor magics, retransmission limits, the cold-field pointers) are needed in ONE phase of the
    // subframe; as wave-uniform values they would sit in scalar registers across the whole step loop, where the budget of 102 is
    // already spent — and get spilled and reloaded (v_readlane) on the critical chain.  Held in vector registers instead (the
    // wavefront has ~25 to spare at 4 waves per SIMD) they cost nothing to use: a VALU operand either way.
#define LK_TO_VGPR(x) asm volatile("" : "+v"(x))
    LK_TO_VGPR(K.fmP.d); LK_TO_VGPR(K.fmP.M); LK_TO_VGPR(K.fmB.d); LK_TO_VGPR(K.fmB.M); LK_TO_VGPR(K.fm5.d); LK_TO_VGPR(K.fm5.M);
    LK_TO_VGPR(K.maxMsg2); LK_TO_VGPR(K.seed_lo); LK_TO_VGPR(K.seed_hi);
    LK_TO_VGPR(K.ptc); LK_TO_VGPR(K.ftt); LK_TO_VGPR(K.stt); LK_TO_VGPR(K.fcnt);
#undef LK_TO_VGPR
    const int nUE = K.nUE, nP = K.nP, aT = K.aT;
    const int stop = PD->stop, nGrantUL = PD->nGrantUL, binshift = PD->binshift;
    const PRACH_G int *const sched = (const PRACH_G int *)PD->sched;
    const bool withnoma = K.withnoma;

    int4 *const lrec = reinterpret_cast<int4 *>(smem + lo::TAIL);
    // Philox: draw index + the next two draws per slot.  GLIBC: instead, rand() calls of every 64-UE group of the trial in this subframe
    // (gsum), their exclusive prefix in index order (gpre), and per OWN group the lanes that draw once or twice / twice (gmask)
    int2 *const ldraw = GLIBC ? nullptr : reinterpret_cast<int2 *>(smem + lo::TAIL + 16 * lslots);
    unsigned *const lnd = GLIBC ? nullptr : reinterpret_cast<unsigned *>(smem + lo::TAIL + 24 * lslots);
    int *const gsum = reinterpret_cast<int *>(smem + lo::TAIL + 16 * lslots), *const gpre = gsum + gcap;
    unsigned *const gmask = reinterpret_cast<unsigned *>(gpre + gcap); // [4 * lslots / 64]
    const PRACH_G int *const stream = (const PRACH_G int *)PD->stream;
    const unsigned long long stream_len = PD->stream_len;
    unsigned long long base = 0; // GLIBC: rand() calls consumed so far (relative to the stream window)
    int zs[4] = {-1, -1, -1, -1}, nzs = 0; // GLIBC: the slots this lane ran through the select pass (their groups' counts are zeroed behind S1)
    int *const scal = LI(lo::SCAL);
    int *const queue = LI(lo::QUEUE);
    int2 *const gev = LI2(lo::GEV);

    const int totgroups = (nUE + 63) >> 6;
    const int lgroups = (totgroups + G - 1) / G; // local groups of any workgroup (upper bound)

    // mailboxes: [2 parities][G workgroups][mbs granules]: header, nP bucket granules, CLUSTER_EVW event granules
    const unsigned mbs = (unsigned)PD->mbstride >> 1; // granules per mailbox
    PRACH_G long long *const mbox = (PRACH_G long long *)PD->mbox;
    const unsigned parstride = (unsigned)G * mbs;       // granules per parity
    // this thread's fixed roles in the exchange, as granule offsets inside a parity block
    unsigned r1off[3];
    int r1p[3];
#pragma unroll
    for (int u = 0; u < 3; u++) { // round 1: up to 3072 bucket granules (G = 48 x 64 preambles); more: the tail loop below
        const int k = tid + u * WG_THREADS;
        const int wg = k / max(nP, 1), p = k - wg * nP;
        r1p[u] = k < G * nP ? p : -1;
        r1off[u] = (unsigned)wg * mbs + 1u + (unsigned)p;
    }
    unsigned r2off[2]; // round 1 also fetches the first LEPF event granules of every mailbox: this thread's (workgroup, slot)
    int r2wg[2], r2es[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int k = tid + u * WG_THREADS;
        r2wg[u] = k < G * LEPF ? k / LEPF : -1;
        r2es[u] = k % LEPF;
        r2off[u] = (unsigned)(k / LEPF) * mbs + 1u + (unsigned)nP + (unsigned)(k % LEPF);
    }
    const int hl = tid - (WG_THREADS - 64);            // the last wavefront reads the headers: lane = workgroup
    const unsigned hoff = (unsigned)max(hl, 0) * mbs;
    const unsigned myoff = (unsigned)b * mbs;           // own mailbox

    // calloc + initialUE (Beta.c:78-83) for the groups this workgroup owns
    const unsigned nd0 = K.withnoma ? 2u : 0u; // activateUEs draws twice before the first preamble draw (WithNOMA:393-394)
    for (int x = tid; x < lslots; x += WG_THREADS) {
        lrec[x] = make_int4(-1, 0, 0, 0); // (every slot, also past the last UE)
        if (!GLIBC) lnd[x] = nd0;
        if (x < lgroups * 64) {
            const int g = b + G * (x >> 6), i = g * 64 + (x & 63);
            if (g < totgroups && i < nUE) {
                K.ptc[i] = 0; K.ftt[i] = 0; K.stt[i] = 0; K.fcnt[i] = 0;
                if (!GLIBC) {
                    ldraw[x] = make_int2(philox_draw31(K.seed_lo, K.seed_hi, (unsigned)i, nd0, (unsigned)K.nUE, (unsigned)K.variant),
                                         philox_draw31(K.seed_lo, K.seed_hi, (unsigned)i, nd0 + 1u, (unsigned)K.nUE, (unsigned)K.variant));
                    lnd[x] = nd0 | ND_READY;
                }
            }
        }
    }
    if (GLIBC) {
        for (int g = tid; g < gcap; g += WG_THREADS) { gsum[g] = 0; gpre[g] = 0; }
        for (int x = tid; x < 4 * (lslots >> 6); x += WG_THREADS) gmask[x] = 0u;
    }
    if (tid < NPCL) {
#pragma unroll
        for (int par = 0; par < 2; par++) {
            LI(lo::PAR + par * lo::PARSZ + lo::P_HIST)[tid] = 0; LI(lo::PAR + par * lo::PARSZ + lo::P_MLOC)[tid] = INT_MAX;
            LI(lo::PAR + par * lo::PARSZ + lo::P_MLOCS)[tid] = INT_MAX; LI(lo::PAR + par * lo::PARSZ + lo::P_CANDN)[tid] = 0;
            LI(lo::FCALL + par * lo::RBSZ)[tid] = INT_MAX; LI(lo::LCALL + par * lo::RBSZ)[tid] = -1;
            LI(lo::TOTAL + par * lo::RBSZ)[tid] = 0; LI(lo::NLV + par * lo::RBSZ)[tid] = 0; LI(lo::FIE + par * lo::RBSZ)[tid] = 0;
        }
    }
    if (tid < 32) scal[tid] = 0;
    if (tid < SCHR) LI(lo::SCHED)[tid] = ((const PRACH_G int *)PD->sched)[min(tid, PD->maxTime / K.aT + 1)];
    __syncthreads();

    // ---- same-XCD handshake: every workgroup publishes the id of the XCD it runs on (write-through granule, tag 0xFFFF, in the
    // header of its parity-1 mailbox — first used by subframe 1, which no workgroup reaches before every peer is past this point,
    // because subframe 0's exchange needs every peer's subframe-0 granules) and reads all G of them: the cluster keeps its granules
    // in L2 only if they are all equal.  Every workgroup reads the same G values, so all decide alike.
    if (xpack && G > 1 && G <= 64) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xfu;
        PRACH_G long long *const hs = mbox + parstride;
        if (tid == 0) lst(hs + myoff, lmk(xcc, 0u, 0xFFFFu));
        if (tid < 64) {
            bool same = true;
            if (tid < G) same = ((unsigned)lwait(hs + (unsigned)tid * mbs, 0xFFFFu, smem) & 0xFFFFFu) == xcc;
            const bool all = __ballot(!same) == 0ull;
            if (tid == 0) scal[S_SX] = all ? 1 : 0;
        }
        __syncthreads();
        K.sx = scal[S_SX] != 0 && scal[S_STATUS] == PRACH_OK;
    }
    const bool sx = K.sx;

    int activeCheck = 0, grantCheck = 0, tlast = -1, time_exit = stop;
    unsigned long long steps = 0;
    int status = (lgroups * 64 > lslots || lslots > LQ || nP > NPCL || (GLIBC && totgroups > gcap) || G > 64 || lgroups > 64 * NWA || stop >= 0xFFFE) ? PRACH_ERR_UNSUPPORTED : PRACH_OK;
    unsigned long long deadmask = 0; // bit m: this wavefront's m-th group (local group w + NWA * m) is finished for good
    int t5 = 0, tA = 0, slotA = 0;   // t mod 5, t mod accessTime, t / accessTime: kept incrementally
    // arrival table entry of the NEXT access slot: fetched one slot ahead by a VECTOR load whose result is only made scalar
    // (v_readfirstlane) when the slot begins — a scalar use right behind the load would park the wave for a full L2 round trip
    // on the subframe's critical chain once per access slot
    // — and through an LDS ring over the table (refilled by half every SCHR / 2 slots): a global load here would make the compiler wait
    // with vmcnt(0), i.e. also for the exchange's granule loads issued just before phase A of the next subframe
    int *const ring = LI(lo::SCHED);
    const int nsched = PD->maxTime / aT + 2; // entries of the table (prach_engine.hip)
    int acNextV = ring[0];
#ifdef PRACH_STAMPS
    unsigned long long *const fstamps = reinterpret_cast<unsigned long long *>(smem + lo::RCL + 4 * (RCCAP - 56)); // [24] + dstat[4]: the reset-candidate list's space (a trial that has such candidates garbles its profile)
    unsigned long long *const dstat = fstamps + 24; // thread 0: sum of queue lengths, round-1 bucket / event granules read again, refills
    if (tid < 28) fstamps[tid] = 0;
    unsigned long long fprev = __builtin_readcyclecounter();
#define LSTAT(k, v) do { if (threadIdx.x == PRACH_STAMP_TID) dstat[k] += (unsigned long long)(v); } while (0)
#else
#define LSTAT(k, v) do { } while (0)
#endif

    // ---- phase A of subframe ta over this wavefront's groups (prach_cluster.hip compact_phase_a).  SPEC: ahead of the resolver
    // of subframe ta - 1 (its grants are not known: a new caller waits for phase B; a matched UE that gets one is taken out again
    // by the granting thread).  pcA: byte offset of subframe ta's parity block.
    auto phase_a = [&](const bool SPEC, const int ta, const int prevA, const int acA, const int pcA) __attribute__((always_inline)) {
        int *const hist = LI(lo::PAR + pcA + lo::P_HIST), *const mloc = LI(lo::PAR + pcA + lo::P_MLOC), *const mlocs = LI(lo::PAR + pcA + lo::P_MLOCS);
        const int ngroups = (acA + 63) >> 6;
        if (w >= NWA) return; // (the last wavefront is the exchange's: bucket publish, headers, event offsets — see NWA)
        for (int m = 0;; m++) {
            const int j = w + NWA * m, g = b + G * j;
            if (g >= ngroups) break;
            if ((deadmask >> m) & 1ull) continue;
            const int sl = j * 64 + lane, i = g * 64 + lane;
            const int4 r = lrec[sl];
            const unsigned pk = (unsigned)r.w;
            const unsigned pg = pk >> PK_PEND_SHIFT; // deferred outcome | grant bit << 3
            const PassMasks M = SPEC ? pass_masks<true>(pk, r.x, r.z, ta, K.rarlim, g * 64 + 64 > prevA, i, acA, prevA, nUE)
                                     : pass_masks<false>(pk, r.x, r.z, ta, K.rarlim, g * 64 + 64 > prevA, i, acA, prevA, nUE);
            const unsigned long long mHeavy = ~(M.light | M.quiet);
            if ((M.light | mHeavy) == 0ull) {
                if (M.done == ~0ull) deadmask |= 1ull << m; // nothing will ever happen in this group again
                continue;
            }
            const bool lightc = __builtin_amdgcn_inverse_ballot_w64(M.light), heavy = __builtin_amdgcn_inverse_ballot_w64(mHeavy);
            const bool trig = __builtin_amdgcn_inverse_ballot_w64(M.trig);
            if (lightc) { // Beta.c:245 + the txTime++ of Beta.c:346,358
                const bool bump = pg != 0u;
                const bool member = bump || trig; // matched by a preambleCollision scan in this subframe
                if (pg != (unsigned)PEND_STAY) { // (steady contention is not rewritten: it follows from the record's age)
                    const unsigned npk = ((pk & 0x0FFFFFFFu) + (1u << PK_RAR_SHIFT)) | (member ? (unsigned)PEND_STAY << PK_PEND_SHIFT : 0u);
                    lrec[sl] = make_int4(bump ? ta : r.x, r.y, r.z, (int)npk);
                }
                if (member) {
                    const int p1 = (int)((pk >> PK_PRE_SHIFT) & 0xffu) - 1;
                    atomicAdd(&hist[p1], 1);
                    int *const ml = pg == (unsigned)PEND_STAY ? mlocs : mloc;
                    if (__atomic_load_n(&ml[p1], __ATOMIC_RELAXED) > i) atomicMin(&ml[p1], i);
                }
            }
            const unsigned long long hm = mHeavy;
            if (hm) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&scal[S_QN + (ta & 1)], __builtin_amdgcn_readfirstlane(__popcll(hm)));
                base = __builtin_amdgcn_readlane(base, 0);
                if (heavy) queue[base + __popcll(hm & lanemask_lt(lane))] = sl; // (the queue has room for every owned slot)
            }
        }
    };

    // ---- the calls of one subframe (resolver tables at byte offset fa_, N_ gathered events): scan counts and counters; `now`: also
    // the last-caller table and the singleton list (needed by the grant selection / by PEND_RJOIN of the very next pass)
    auto calls = [&](const int fa_, const int N_, const bool now) __attribute__((always_inline)) {
        const int *const fcallA = LI(lo::FCALL + fa_);
        int *const lcallA = LI(lo::LCALL + fa_);
        const int nrj = now ? scal[S_NRJ] : 0; // (a deferred subframe has no Msg3 re-entry)
        int my_coll = 0, my_txop = 0;
        for (int k = tid; k < N_ + nP; k += WG_THREADS) {
            int idx = 0, p = 0, ispre = 0;
            bool caller = false;
            if (k < N_) {
                const int2 e = gev[k];
                const int type = e.y & 7;
                if (type == EVL_CALLER || type == EVL_RESETCAND) { caller = true; idx = e.x; p = (e.y >> 4) & 0xff; ispre = (e.y >> 3) & 1; }
            } else {
                p = k - N_;
                if (fcallA[p] != INT_MAX && !LI(lo::FIE + fa_)[p]) { caller = true; idx = fcallA[p]; ispre = 1; } // a STAY pre-member calls first
            }
            if (!caller) continue;
            const bool first = idx == fcallA[p];
            int rj = 0;
            if (nrj > 0) { // Msg3-timeout re-entries that stayed matched since the previous call on this bucket (rare)
                int prev = (!first) ? fcallA[p] : -1;
                for (int j = 0; j < N_; j++) {
                    const int2 ej = gev[j];
                    const int tj = ej.y & 7;
                    if ((tj == EVL_CALLER || tj == EVL_RESETCAND) && ((ej.y >> 4) & 0xff) == p && ej.x < idx && ej.x > prev) prev = ej.x;
                }
                for (int j = 0; j < N_; j++) {
                    const int2 ej = gev[j];
                    if ((ej.y & 7) == EVL_RJOIN && ((ej.y >> 4) & 0xff) == p && ej.x < idx && ej.x > prev) rj++;
                }
            }
            const int check = 1 + (first ? LI(lo::TOTAL + fa_)[p] - ispre - LI(lo::NLV + fa_)[p] : 0) + rj;
            if (now && lcallA[p] < idx) atomicMax(&lcallA[p], idx);
            if (check == 1) {
                if (now) {
                    const int s_ = atomicAdd(&scal[S_NS], 1);
                    if (s_ < LSC) LI(lo::SIDX)[s_] = idx;
                }
                my_txop += 1;
            } else if (withnoma) { // WithNOMA:650-652
                my_coll += check; my_txop += check;
            } else { // Beta.c:349-351
                my_coll += 1; my_txop += 1;
            }
        }
        if (__any((my_coll | my_txop) != 0)) {
            const int sc_ = wave_sum(my_coll), st_ = wave_sum(my_txop);
            if (lane == 0) { if (sc_) atomicAdd(&scal[S_COLL], sc_); if (st_) atomicAdd(&scal[S_TXOP], st_); }
        }
    };
    int pendN = -1, pendFa = 0; // deferred calls of the previous subframe (pendN < 0: none)

    for (int t = 0; t < stop && status == PRACH_OK; t++) {
        // (an opaque copy of the thread index: role predicates such as `tid < nP` are then recomputed where they are used — one v_cmp —
        //  instead of being hoisted out of the step loop as 64-bit lane masks that live in, and spill from, scalar registers)
        int tl = tid;
        asm volatile("" : "+v"(tl));
        steps++;
        tlast = t;
        if (t5 == 0) grantCheck = 0; // Beta.c:112 (hard-coded 5)
        const int prevAC = activeCheck;
        if (tA == 0 && activeCheck != nUE) { // Beta.c:121-134: this access slot's arrivals (fetched a slot ago)
            activeCheck = __builtin_amdgcn_readfirstlane(acNextV);
            acNextV = ring[(slotA + 1) & (SCHR - 1)];
            if (slotA > 0 && (slotA & (SCHR / 2 - 1)) == 0 && tid < SCHR / 2) // the half of the ring that has just been used up
                ring[(slotA + SCHR / 2 + tid) & (SCHR - 1)] = sched[min(slotA + SCHR / 2 + tid, nsched - 1)];
        }
        const int parity = t & 1;
        const unsigned tag = (unsigned)(t + 1);
        const int pc = parity * lo::PARSZ, pn = (parity ^ 1) * lo::PARSZ; // parity blocks of this subframe / of the next one
        const int fa = parity * lo::RBSZ, fb = (parity ^ 1) * lo::RBSZ;   // resolver tables: this subframe's resolver fills [A]; the pass reads [B]
        PRACH_G long long *const mbpar = mbox + (size_t)parity * parstride;
        PRACH_G long long *const mygr = mbpar + myoff;
        PRACH_G long long *const mbev = mygr + 1 + nP;
        int *const fcallA = LI(lo::FCALL + fa);
        LSTAMP(0); // loop head

        if (t == 0) { // (every later subframe's phase A has run ahead, inside the exchange of the subframe before)
            phase_a(false, 0, prevAC, activeCheck, pc);
            __syncthreads();
        }
        { // ---- phase B: the queued UEs through the full body, 64 at a time ----
            int c_succ = 0, c_contf = 0;
            const int sl_first = queue[w * 64 + lane]; // (read together with the queue length: one LDS round trip, not two)
            const int qn = scal[S_QN + parity];
            LSTAT(0, qn);
            const LG GL0{nullptr, nullptr, nullptr, nullptr, 0ull};
            if (!GLIBC) {
                for (int q0 = w * 64; q0 < qn; q0 += NW * 64) {
                    const bool v = q0 + lane < qn;
                    const int sl = v ? (q0 == w * 64 ? sl_first : queue[q0 + lane]) : 0;
                    const int i = l_idx_of(K, sl);
                    int4 r = make_int4(-1, 0, 0, 0);
                    unsigned ndc = 0;
                    if (v) { r = lrec[sl]; ndc = lnd[sl]; }
                    // (an idle lane keeps the idle record: l_step leaves it alone)
                    l_step<0>(smem, K, lrec, lnd, ldraw, GL0, pc, fb, t, prevAC, mbev, tag, lane, i, sl, v && i < activeCheck, r, ndc, c_succ, c_contf);
                }
            } else {
                // The reference's own stream: every draw's position = rand() calls of earlier subframes + this subframe's activation draws
                // (activateUEs, WithNOMA:393-394) + the index-ordered prefix of per-UE call counts.  Count pass over the queued UEs, exchange of
                // the per-group counts (two per granule, behind the event granules of the mailbox), block-wide prefix, select pass.
                const unsigned long long actdraws = withnoma ? 2ull * (unsigned long long)(activeCheck - prevAC) : 0ull;
                const int ngroups_t = (activeCheck + 63) >> 6;
                const LG GC{gsum, gpre, gmask, stream, 0ull};
                for (int q0 = w * 64; q0 < qn; q0 += NW * 64) {
                    const bool v = q0 + lane < qn;
                    const int sl = v ? (q0 == w * 64 ? sl_first : queue[q0 + lane]) : 0;
                    const int i = l_idx_of(K, sl);
                    int4 r = make_int4(-1, 0, 0, 0);
                    if (v) r = lrec[sl];
                    l_step<1>(smem, K, lrec, lnd, ldraw, GC, pc, fb, t, prevAC, mbev, tag, lane, i, sl, v && i < activeCheck, r, 0u, c_succ, c_contf);
                }
                __syncthreads(); // the counts of this workgroup's groups are complete
                {
                    const int nq = (lgroups + 4) / 5;
                    PRACH_G long long *const mine = mygr + 1 + nP + CLUSTER_EVW;
                    for (int q = tl; q < nq; q += WG_THREADS) {
                        unsigned c5[5];
#pragma unroll
                        for (int u_ = 0; u_ < 5; u_++) { const int g = b + G * (5 * q + u_); c5[u_] = g < totgroups ? (unsigned)gsum[g] : 0u; }
                        lstx(sx, mine + q, lmk(c5[0] | (c5[1] << 8) | ((c5[2] & 0xFu) << 16), (c5[2] >> 4) | (c5[3] << 4) | (c5[4] << 12), tag));
                    }
                    // (five 8-bit counts per granule — a 64-UE group makes at most 128 calls: fewer granules, fewer pollers.  Gathering them directly
                    //  into the scanning threads' registers, four polls per thread and no staging barrier, measured 2.5 ms SLOWER per trial:
                    //  4096 pollers per workgroup get in the way of the stores they wait for)
                    for (int k = tl; k < G * nq; k += WG_THREADS) {
                        const int wg = k / nq, q = k - wg * nq;
                        if (wg == b) continue;
                        const long long g_ = lwait(mbpar + (unsigned)wg * mbs + 1u + (unsigned)nP + (unsigned)CLUSTER_EVW + (unsigned)q, tag, smem);
                        const unsigned lo_ = (unsigned)g_ & 0xFFFFFu, hi_ = (unsigned)((unsigned long long)g_ >> 32) & 0xFFFFFu;
                        const unsigned c5[5] = {lo_ & 0xFFu, (lo_ >> 8) & 0xFFu, (lo_ >> 16) | ((hi_ & 0xFu) << 4), (hi_ >> 4) & 0xFFu, (hi_ >> 12) & 0xFFu};
#pragma unroll
                        for (int u_ = 0; u_ < 5; u_++) { const int g = wg + G * (5 * q + u_); if (g < totgroups) gsum[g] = (int)c5[u_]; }
                    }
                    __syncthreads();
                    int vv[4], sum = 0;
#pragma unroll
                    for (int u_ = 0; u_ < 4; u_++) { const int g = tl * 4 + u_; vv[u_] = g < ngroups_t ? gsum[g] : 0; sum += vv[u_]; }
                    const int x = wave_scan_incl(sum);
                    if (lane == 63) LI(lo::WTOT)[w] = x;
                    __syncthreads();
                    int run = x - sum;
                    for (int k = 0; k < w; k++) run += LI(lo::WTOT)[k];
#pragma unroll
                    for (int u_ = 0; u_ < 4; u_++) { const int g = tl * 4 + u_; if (g < gcap) gpre[g] = run; run += vv[u_]; }
                    if (tl == WG_THREADS - 1) scal[S_NCROSS] = run; // (free here: only the grant selection uses it)
                }
                __syncthreads();
                const unsigned long long tot = actdraws + (unsigned long long)scal[S_NCROSS];
                if (scal[S_STATUS] != PRACH_OK) { status = scal[S_STATUS]; time_exit = t; break; }
                if (base + tot > stream_len) { status = PRACH_ERR_STREAM; time_exit = t; break; } // (the engine retries with a larger window)
                const LG GS{gsum, gpre, gmask, stream, base + actdraws};
                nzs = 0;
                for (int q0 = w * 64; q0 < qn; q0 += NW * 64) {
                    const bool v = q0 + lane < qn;
                    const int sl = v ? queue[q0 + lane] : 0;
                    const int i = l_idx_of(K, sl);
                    int4 r = make_int4(-1, 0, 0, 0);
                    if (v) r = lrec[sl];
                    l_step<2>(smem, K, lrec, lnd, ldraw, GS, pc, fb, t, prevAC, mbev, tag, lane, i, sl, v && i < activeCheck, r, 0u, c_succ, c_contf);
                    if (nzs < 4) zs[nzs] = v ? sl : -1; // (remembered in registers: the queue's entries are overwritten by phase A of t + 1 behind S1)
                    nzs++;
                }
                base += tot;
            }
            if (__any((c_succ | c_contf) != 0)) {
                const int ss_ = wave_sum(c_succ), sf_ = wave_sum(c_contf);
                if (lane == 0) {
                    if (ss_) atomicAdd(&scal[S_NSUCC], ss_);
                    if (sf_) atomicAdd(&scal[S_CONTF], sf_);
                }
            }
        }
        LSTAMP(1); // phase B
        __syncthreads(); // S1: histogram / lowest callers / candidate list of this workgroup are complete; [B] is free
        LSTAMP(2);
        if (GLIBC) {
            // every draw of this subframe has been read: the counts of the groups the queued UEs are in can go (own groups only ever get counts
            // from this workgroup; nobody looks at them again before the count pass of the next subframe, several barriers from here)
#pragma unroll
            for (int r_ = 0; r_ < 4; r_++) { // (the queue holds at most LQ = 4096 entries: four batches per wavefront)
                if (r_ < nzs && zs[r_] >= 0) {
                    const int sl = zs[r_], jl = sl >> 6;
                    gsum[l_idx_of(K, sl) >> 6] = 0; gmask[4 * jl] = 0u; gmask[4 * jl + 1] = 0u; gmask[4 * jl + 2] = 0u; gmask[4 * jl + 3] = 0u;
                }
            }
            nzs = 0;
            // (pulling the part of the stream window the next subframes draw from into L2 ahead of time, one slice per workgroup: 0.6 ms slower)
        }
        // publish, first part: per bucket {histogram, lowest caller} — complete since S1; the header (event count) follows the leaver
        // filter.  Self-validating granules: the earlier they leave, the fewer of them the other workgroups have to read twice.
        if (tl >= WG_THREADS - 64 && tl - (WG_THREADS - 64) < nP) { // (the last wavefront: the first ones run the leaver filter)
            const int k = tl - (WG_THREADS - 64);
            const int ml = min(LI(lo::PAR + pc + lo::P_MLOC)[k], LI(lo::PAR + pc + lo::P_MLOCS)[k]);
            lstx(sx, mygr + 1 + k, lmk((unsigned)LI(lo::PAR + pc + lo::P_HIST)[k], ml == INT_MAX ? GRL_NONE : (unsigned)ml, tag));
        }

        // early leavers below this workgroup's lowest caller are the only ones a rank can need.  No candidate (the usual case outside
        // the overloaded part of a trial): the event count is final since S1 and the header leaves at once, without the barrier S2.
        {
            int *const nevp = &scal[S_NEV + 2 * parity]; // (this subframe's pair of counters)
            const int2 nc = *reinterpret_cast<const int2 *>(nevp); // {events so far, early-leaver candidates}
            int nevraw = nc.x;
            if (tl == 64) { scal[S_NS] = 0; scal[S_NRC] = 0; scal[S_NRJ] = 0; scal[S_QN + parity] = 0; } // (this subframe's queue has been consumed)
            if (nc.y > 0) {
                const int *const mloc = LI(lo::PAR + pc + lo::P_MLOC), *const mlocs = LI(lo::PAR + pc + lo::P_MLOCS);
                for (int k = tl; k < nc.y; k += WG_THREADS) {
                    const int2 c = LI2(lo::LCAND)[k];
                    if (c.x < min(mloc[c.y], mlocs[c.y])) {
                        const int es = atomicAdd(nevp, 1);
                        if (es < CLUSTER_EVW) lstx(sx, mbev + es, lmk((unsigned)c.x, (unsigned)(EVL_LEAVER | (c.y << 4)), tag));
                    }
                }
                LSTAMP(3); // leaver filter
                __syncthreads(); // S2
                LSTAMP(4);
                if (tl == 64) nevraw = *nevp;
            }
            // publish: header {#events, overflow, #successes} (the pair of counters is reset behind S3 — every thread has read it by
            // then — and used again by the pass of subframe t + 2, two unconditional barriers later)
            // (the overflow bit also carries a capacity this workgroup ALONE has exceeded in its pass — the candidate list — so that every
            //  workgroup of the cluster leaves at the same S3 with PRACH_ERR_INTERNAL instead of spinning for a peer that has left)
            if (tl == 64) lstx(sx, mygr, lmk((unsigned)min(nevraw, CLUSTER_EVW) | ((nevraw > CLUSTER_EVW || scal[S_STATUS] == PRACH_ERR_INTERNAL) ? (1u << 13) : 0u), (unsigned)scal[S_NSUCC], tag));
        }
        LSTAMP(5); // publish
        // Phase A of the NEXT subframe runs while the other workgroups' granules are on their way; then round 1 is issued — the bucket
        // granules of every workgroup, the first event granules and, on the last wavefront, the headers —, the refill and the deferred
        // calls run while it is in flight, and every granule is checked: one whose tag is still the old one is re-read until it arrives.
        // (Round 1 in front of phase A reads the slower workgroups' granules before they are written: same time within 0.5 %.  Re-reading
        // ALL late granules of a thread together instead of one by one: 57-60 ms instead of 52.5 — polls get in the way of the stores
        // they wait for.)
        const bool ahead = t + 1 < stop;
        if (ahead) {
            // subframe t+1: arrivals of its access slot (Beta.c:121-134), then phase A on this workgroup's records as the pass of
            // subframe t left them
            const int acN = (tA + 1 == aT && activeCheck != nUE) ? __builtin_amdgcn_readfirstlane(acNextV) : activeCheck;
            phase_a(true, t + 1, activeCheck, acN, pn);
        }
        LSTAMP(16);
        long long gv[3] = {0, 0, 0}, hv = 0;
#pragma unroll
        for (int u = 0; u < 3; u++)
            if (r1p[u] >= 0) gv[u] = lld(mbpar + r1off[u]);
        if (hl >= 0 && hl < G) hv = lld(mbpar + hoff);
        long long ev2[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (r2wg[u] >= 0) ev2[u] = lld(mbpar + r2off[u]);
        LSTAMP(17);
        if (!GLIBC) { // refill: the next two draws of every UE that drew in this subframe's phase B (off the chain: the exchange is in flight)
            const int nrq = min(scal[S_NRQ], LRQ);
            LSTAT(3, scal[S_NRQ]);
            for (int k = (tl + WG_THREADS / 2) & (WG_THREADS - 1); k < nrq; k += WG_THREADS) { // (from wavefront 8 on: wavefront 0 has the deferred calls)
                const int sl = LI(lo::RQ)[k];
                const int i = l_idx_of(K, sl);
                const unsigned nd = lnd[sl] & ~ND_READY;
                ldraw[sl] = make_int2(philox_draw31(K.seed_lo, K.seed_hi, (unsigned)i, nd, (unsigned)nUE, (unsigned)K.variant),
                                      philox_draw31(K.seed_lo, K.seed_hi, (unsigned)i, nd + 1u, (unsigned)nUE, (unsigned)K.variant));
                lnd[sl] = nd | ND_READY;
            }
        }
        LSTAMP(18);
        if (pendN >= 0) { calls(pendFa, pendN, false); pendN = -1; } // the previous subframe's deferred calls (its event list is intact until S3)
        LSTAMP(6); // phase A ahead
        {
            int *const total = LI(lo::TOTAL + fa);
#pragma unroll
            for (int u = 0; u < 3; u++) {
                if (r1p[u] >= 0) {
                    long long g_ = gv[u];
                    if (!lok(g_, tag)) { LSTAT(1, 1); g_ = lwait(mbpar + r1off[u], tag, smem); }
                    const unsigned h = (unsigned)g_ & 0xFFFFFu, ml = (unsigned)((unsigned long long)g_ >> 32) & 0xFFFFFu;
                    if (h) atomicAdd(&total[r1p[u]], (int)h);
                    if (ml != GRL_NONE) atomicMin(&fcallA[r1p[u]], (int)ml);
                }
            }
            for (int k = tl + 3 * WG_THREADS; k < G * nP; k += WG_THREADS) { // (more than 3072 bucket granules)
                const int wg = k / nP, p = k - wg * nP;
                const long long g_ = lwait(mbpar + (unsigned)wg * mbs + 1u + (unsigned)p, tag, smem);
                const unsigned h = (unsigned)g_ & 0xFFFFFu, ml = (unsigned)((unsigned long long)g_ >> 32) & 0xFFFFFu;
                if (h) atomicAdd(&total[p], (int)h);
                if (ml != GRL_NONE) atomicMin(&fcallA[p], (int)ml);
            }
            LSTAMP(19);
            if (hl >= 0) {
                int nev = 0, nsuc = 0, ovf = 0;
                if (hl < G) {
                    long long g_ = hv;
                    LSTAT(2, __popcll(__ballot(!lok(g_, tag))) ? 1 : 0); // (subframes in which a header was late)
                    if (!lok(g_, tag)) g_ = lwait(mbpar + hoff, tag, smem);
                    const unsigned w0 = (unsigned)g_ & 0xFFFFFu;
                    nev = (int)(w0 & 0x1FFFu); ovf = (int)((w0 >> 13) & 1u); nsuc = (int)((unsigned)((unsigned long long)g_ >> 32) & 0xFFFFFu);
                }
                const int x = wave_scan_incl(nev); // (the whole wavefront is here: hl = lane)
                LI(lo::EVOFF)[hl] = x - nev;
                const int rem = wave_sum(max(nev - LEPF, 0)); // events beyond the granules fetched in round 1
                nsuc = wave_sum(nsuc);
                ovf = __ballot(ovf != 0) != 0ull;
                if (hl == 63) { scal[S_NTOT] = x | (ovf ? 1 << 30 : 0); LI(lo::EVOFF)[64] = x; }
                if (hl == 0) { scal[S_NREM] = rem; scal[S_NSUCCTOT] = nsuc; }
            }
        }
        LSTAMP(7); // round-1 granules taken
        __syncthreads(); // S3: totals, lowest definite callers, event offsets (and the next subframe's phase A)
        LSTAMP(8);
        if (tl < NPCL) { // this parity is used again in two subframes
            LI(lo::PAR + pc + lo::P_HIST)[tl] = 0; LI(lo::PAR + pc + lo::P_MLOC)[tl] = INT_MAX;
            LI(lo::PAR + pc + lo::P_MLOCS)[tl] = INT_MAX; LI(lo::PAR + pc + lo::P_CANDN)[tl] = 0;
            // the resolver tables of the subframe before (read by this subframe's pass and by its deferred calls, both done): next subframe's [A]
            LI(lo::FCALL + fb)[tl] = INT_MAX; LI(lo::LCALL + fb)[tl] = -1; LI(lo::TOTAL + fb)[tl] = 0; LI(lo::NLV + fb)[tl] = 0; LI(lo::FIE + fb)[tl] = 0;
        }
        if (tl == 64) { scal[S_NRQ] = 0; scal[S_NEV + 2 * parity] = 0; scal[S_NCAND + 2 * parity] = 0; } // (the refill list and this subframe's event / candidate counts have been consumed)
        // {status, gathered events | overflow << 30, events beyond round 1, successes}.  The status word is looked at HERE only: an error
        // raised behind S3 (a capacity of the resolver, a time-out of round 2) ends the trial one subframe later — the engine reruns it anyway
        const int4 ctl = *reinterpret_cast<const int4 *>(&scal[S_STATUS]);
        if (ctl.x != PRACH_OK) { status = ctl.x; time_exit = t; break; }
        const int N = ctl.y & 0x3FFFFFFF;
        if ((ctl.y >> 30) || N > LEV) { status = PRACH_ERR_INTERNAL; time_exit = t; break; } // engine: exact rerun on the general kernels
        // the event granules: the first LEPF of every mailbox came with round 1 (each by the thread that fetched it), classified
        // against the lowest definite callers; a mailbox with more has the rest read now (a second round trip, rare)
        {
            const int *const evoff = LI(lo::EVOFF);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (r2wg[u] >= 0) {
                    const int o0 = evoff[r2wg[u]], nev = evoff[r2wg[u] + 1] - o0;
                    if (r2es[u] < nev) {
                        long long e = ev2[u];
                        if (!lok(e, tag)) { e = lwait(mbpar + r2off[u], tag, smem); }
                        const int2 ev = make_int2((int)((unsigned)e & 0xFFFFFu), (int)((unsigned)((unsigned long long)e >> 32) & 0xFFFFFu));
                        gev[o0 + r2es[u]] = ev;
                        l_classify(smem, fa, o0 + r2es[u], ev);
                    }
                }
            }
            if (ctl.z > 0) {
                for (int k = tl; k < N; k += WG_THREADS) {
                    int lo_ = 0, hi_ = G; // workgroup whose segment holds event k
                    while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if (evoff[mid] <= k) lo_ = mid; else hi_ = mid; }
                    const int es = k - evoff[lo_];
                    if (es < LEPF) continue;
                    const long long e = lwait(mbpar + (unsigned)lo_ * mbs + 1u + (unsigned)nP + (unsigned)es, tag, smem);
                    const int2 ev = make_int2((int)((unsigned)e & 0xFFFFFu), (int)((unsigned)((unsigned long long)e >> 32) & 0xFFFFFu));
                    gev[k] = ev;
                    l_classify(smem, fa, k, ev);
                }
            }
        }
        LSTAMP(9); // round 2
        if (N > 0) __syncthreads(); // S4: events gathered and classified against the lowest DEFINITE callers (N is uniform)
        LSTAMP(10);

        // ---- resolve (identical on every workgroup of the cluster) ----
        const int nrc = scal[S_NRC];
        if (nrc > 0) { // rare: reset cycles that may re-join — decided strictly in index order, then recount
            if (nrc > RCCAP) { status = PRACH_ERR_INTERNAL; time_exit = t; break; }
            if (tl < 64) l_resolve_reset_candidates(smem, fa, nrc, nP);
            else if (tl < 64 + NPCL) { LI(lo::NLV + fa)[tl - 64] = 0; LI(lo::FIE + fa)[tl - 64] = 0; }
            __syncthreads();
            for (int k = tl; k < N; k += WG_THREADS) {
                const int2 e = gev[k];
                const int type = e.y & 7, p = (e.y >> 4) & 0xff;
                if (type == EVL_LEAVER) { if (e.x < fcallA[p]) atomicAdd(&LI(lo::NLV + fa)[p], 1); }
                else if ((type == EVL_CALLER || type == EVL_RESETCAND) && e.x == fcallA[p]) LI(lo::FIE + fa)[p] = 1;
            }
            __syncthreads();
        }
        // every call: scan count `check` (Beta.c:321-330), counters (Beta.c:334,349-351 / WithNOMA:650-652).  With no UL grant left in
        // this 5 ms window and no Msg3 re-entry in this subframe nothing the calls produce is needed by the next pass — the
        // singleton list only feeds the grant selection, the last-caller table only PEND_RJOIN — so they are DEFERRED into the
        // next subframe's exchange window (off the critical chain; the counters are the same whenever they are added up).
        const int Gr = max(0, nGrantUL - 1 - grantCheck); // Beta.c:336-347
        const bool defer = Gr == 0 && nrc == 0 && scal[S_NRJ] == 0 && t + 1 < stop;
        if (defer) { pendN = N; pendFa = fa; }
        else calls(fa, N, true);
        LSTAMP(11); // calls
        if (!defer) __syncthreads(); // S5: calls done; singles listed
        LSTAMP(12);
        const int ns = defer ? 0 : scal[S_NS]; // (deferred: the window's grants are used up, grantCheck no longer matters: Beta.c:336)
        if (ns > LSC) { status = PRACH_ERR_INTERNAL; time_exit = t; break; }
        // An UL grant: the grant bit into the UE's record.  Phase A of subframe t+1 has already run: it took a matched UE
        // (PEND_STAY) for steadily contending, counted it into its bucket and did not queue it — take it out again and queue it.
        // A granted UE was the only caller of a bucket nobody else stayed matched in, so it is the only one phase A counted there
        // as matched before (MLOCS).
        auto grant = [&](const int my) {
            const unsigned x = (unsigned)((my >> 6) - b);
            const unsigned q = x / (unsigned)G; // (owner check done by the caller: x is a multiple of G)
            const int sl = (int)(q * 64u) + (my & 63);
            atomicOr(reinterpret_cast<unsigned *>(&lrec[sl].w), PK_GRANT_BIT);
            if (ahead) {
                const int4 r = lrec[sl];
                const unsigned pk = (unsigned)r.w & ~PK_GRANT_BIT;
                if ((pk >> PK_PEND_SHIFT) == (unsigned)PEND_STAY && l_light(pk, r.x, r.z, t + 1, K.rarlim)) {
                    const int p1 = (int)((pk >> PK_PRE_SHIFT) & 0xffu) - 1;
                    atomicSub(&LI(lo::PAR + pn + lo::P_HIST)[p1], 1);
                    LI(lo::PAR + pn + lo::P_MLOCS)[p1] = INT_MAX;
                    queue[atomicAdd(&scal[S_QN + (parity ^ 1)], 1)] = sl; // ... and its grant is applied by phase B of subframe t+1
                }
            }
        };
        auto mine = [&](const int my) { return ((unsigned)(my >> 6) % (unsigned)G) == (unsigned)b; };
        if (Gr > 0 && ns > 0 && ns <= 64) {
            // up to one wavefront of singleton callers: every lane ranks its own index against the others through v_readlane
            if (tl < 64) {
                const int nsu = __builtin_amdgcn_readfirstlane(ns);
                const int my = tl < nsu ? LI(lo::SIDX)[tl] : INT_MAX;
                int rank = 0;
                for (int s_ = 0; s_ < nsu; s_++) rank += __builtin_amdgcn_readlane(my, s_) < my ? 1 : 0;
                if (tl < nsu && rank < Gr && mine(my)) grant(my);
            }
        } else if (Gr > 0 && ns > 0) { // (most subframes of an overloaded 5 ms window have no grant left: nothing to select)
            // the Gr lowest-index singleton callers, in O(ns): counts per index bin, block-wide exclusive prefix, whole bins below the
            // crossing bin are granted, the crossing bin is ranked exactly
            int *const bins = LI(lo::BINS), *const sidx = LI(lo::SIDX), *const rcl = LI(lo::RCL), *const wtot = LI(lo::WTOT);
            bins[tl] = 0;
            if (tl == 0) scal[S_NCROSS] = 0;
            __syncthreads();
            for (int j = tl; j < ns; j += WG_THREADS) atomicAdd(&bins[sidx[j] >> binshift], 1);
            __syncthreads();
            {
                const int c = bins[tl];
                const int x = wave_scan_incl(c);
                if (lane == 63) wtot[w] = x;
                __syncthreads();
                int add = 0;
                for (int k = 0; k < w; k++) add += wtot[k];
                bins[tl] = x - c + add; // exclusive prefix
            }
            __syncthreads();
            for (int j = tl; j < ns; j += WG_THREADS) {
                const int my = sidx[j];
                const int bin = my >> binshift;
                const int before = bins[bin];
                if (before >= Gr) continue;
                const int cnt = (bin + 1 < LGB ? bins[bin + 1] : ns) - before;
                if (before + cnt <= Gr) { if (mine(my)) grant(my); }
                else { const int s_ = atomicAdd(&scal[S_NCROSS], 1); if (s_ < RCCAP) rcl[s_] = my; }
            }
            __syncthreads();
            const int ncross = scal[S_NCROSS];
            if (ncross > RCCAP) { status = PRACH_ERR_INTERNAL; time_exit = t; break; }
            if (tl < ncross) {
                const int my = rcl[tl];
                int rank = bins[my >> binshift];
                for (int m = 0; m < ncross; m++) rank += rcl[m] < my ? 1 : 0;
                if (rank < Gr && mine(my)) grant(my);
            }
        }
        grantCheck += ns;
        const int nsucc_tot = ctl.w;
        LSTAMP(13); // grants
        if (Gr > 0 && ns > 0) __syncthreads(); // S6: the grants are in the records before the next pass reads them
        LSTAMP(14);
        if (nsucc_tot == nUE) { time_exit = t; break; } // Beta.c:180
        if (++t5 == 5) t5 = 0;
        if (++tA == aT) { tA = 0; slotA++; }
    }
    __syncthreads();
    if (status == PRACH_OK && scal[S_STATUS] != PRACH_OK) status = scal[S_STATUS]; // (raised behind the last S3)
    if (pendN >= 0 && status == PRACH_OK) { calls(pendFa, pendN, false); pendN = -1; }
    __syncthreads();

    // ---- the deferred outcome of the last subframe (prach_cluster.hip cluster_pass<3>), end-of-trial sums (Beta.c:185-197)
    // and the logged fields (Beta.c:501-508) of the owned UEs
    const int tend = tlast + 1;
    {
        const int fl = (tlast & 1) * lo::RBSZ;
        const int *const fcall = LI(lo::FCALL + fl), *const lcall = LI(lo::LCALL + fl);
        PRACH_G int *const timers = (PRACH_G int *)PD->timers;
        PRACH_G v4i_t *const logs = (PRACH_G v4i_t *)PD->logs;
        long long sumT = 0;
        int ptcS = 0, fcS = 0;
        unsigned long long ndS = 0;
        for (int x = tid; x < lgroups * 64 && x < lslots; x += WG_THREADS) {
            const int g = b + G * (x >> 6), i = g * 64 + (x & 63);
            if (g >= totgroups || i >= nUE) continue;
            const int4 r = lrec[x];
            UeState u = unpack(r);
            if (status == PRACH_OK && tlast >= 0 && u.pend != PEND_NONE) { // (as in l_step, with tp = tlast)
                if (u.pend == PEND_STAY) { u.rar += tlast - u.tx; u.tx = tlast; }
                ue_apply(u, ((unsigned)r.w & PK_GRANT_BIT) != 0u, i, tlast, K.fmA, CallTables{fcall, lcall});
            }
            const int timer = u.act == ACT_IDLE ? -1 : (u.act == ACT_DONE ? u.tb : tend - u.tb);
            const int ptc = K.ptc[i], fc = K.fcnt[i];
            if (u.act == ACT_DONE) { sumT += timer; ptcS += ptc; fcS += fc; }
            if (!GLIBC && u.act != ACT_IDLE) ndS += lnd[x] & ~ND_READY; // (a UE that never arrived drew nothing)
            timers[i] = u.act == ACT_DONE ? timer : INT_MIN;
            if (logs) {
                prach_ue_log o;
                o.idx = i; o.timer = timer; o.active = u.act - 1; o.txTime = u.tx; o.firstTxTime = K.ftt[i];
                o.secondTxTime = K.stt[i]; o.nowBackoff = now_backoff(u.bo, tend); o.preamble = u.pre - 1;
                o.preambleChange = u.pre != 0; o.rarWindow = u.rar; o.maxRarCounter = u.mrc; o.preambleTxCounter = ptc;
                o.msg2Flag = (u.act == ACT_M3 || u.act == ACT_DONE); o.connectionRequest = u.conn == 2 ? 48 : u.conn;
                o.msg4Flag = u.act == ACT_DONE; o.failCount = fc;
                store_log(logs, i, o);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            sumT += __shfl_down(sumT, d); ptcS += __shfl_down(ptcS, d); fcS += __shfl_down(fcS, d); ndS += __shfl_down(ndS, d);
        }
        if (lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&scal[S_SUMT]), (unsigned long long)sumT);
            atomicAdd(reinterpret_cast<unsigned long long *>(&scal[S_ND]), ndS);
            atomicAdd(&scal[S_PTC], ptcS);
            atomicAdd(&scal[S_FC], fcS);
        }
    }
    __syncthreads();
    if (tid == 0) { // DevResult was zeroed by the engine before the launch
        PRACH_G DevResult *o = (PRACH_G DevResult *)PD->out;
        gadd(&o->sumTimer, *reinterpret_cast<long long *>(&scal[S_SUMT]));
        if (GLIBC) { if (b == 0) gadd(&o->draws, base); } // (every workgroup has counted the same stream position)
        else gadd(&o->draws, *reinterpret_cast<unsigned long long *>(&scal[S_ND]));
        gadd(&o->ptcSum, scal[S_PTC]);
        gadd(&o->fcSum, scal[S_FC]);
        gadd(&o->nSuccess, scal[S_NSUCC]);
        gadd(&o->finalSuccess, scal[S_NSUCC]);
        gadd(&o->continueFailed, scal[S_CONTF]);
        if (status != PRACH_OK) gmin(&o->status, status);
        if (status == PRACH_ERR_INTERNAL) o->hard_error = 1; // (a capacity was exceeded: that, not a peer's time-out, is what the engine must act on)
#ifdef PRACH_STAMPS
        if (PD->diag) for (int k = 0; k < 28; k++) ((PRACH_G unsigned long long *)PD->diag)[b * 32 + k] = fstamps[k]; // (every workgroup's own clock: dstat follows the 24 stamps)
#endif
        if (b == 0) {
#ifdef PRACH_STAMPS
            for (int k = 0; k < 24; k++) o->fstamps[k] = fstamps[k];
            for (int k = 0; k < 4; k++) o->dbg[k] = dstat[k];
#endif
            o->time_exit = time_exit;
            o->collisionPreambles = scal[S_COLL];
            o->totalPreambleTxop = scal[S_TXOP];
            o->activeCheck = activeCheck;
            o->steps = steps;
        }
    }
}

// glibc (the reference's own rand() stream): no draws computed ahead (16 B per slot), but the per-group call counts of the whole trial and
// their prefix (8 B per group, `groups` rounded up) and four mask words per own group
size_t lcluster_kernel_lds_bytes(int lslots, bool glibc, int groups) {
    if (!glibc) return (size_t)lo::TAIL + (size_t)lslots * 28;
    return (size_t)lo::TAIL + (size_t)lslots * 16 + (size_t)lcluster_group_capacity(groups) * 8 + (size_t)(lslots / 64) * 16;
}
int lcluster_group_capacity(int groups) { return (groups + 63) / 64 * 64; }
int lcluster_max_groups_glibc() { return 4 * WG_THREADS; } // (the block-wide prefix takes 4 groups per thread)
int lcluster_max_preambles() { return NPCL; }

hipError_t launch_lcluster_kernel(const TrialDev *params, int ntrials, int G, int lslots, int xpack, bool glibc, int groups, hipStream_t stream) {
    const size_t lds = lcluster_kernel_lds_bytes(lslots, glibc, groups);
    const void *fn = glibc ? reinterpret_cast<const void *>(&lcluster_kernel<true>) : reinterpret_cast<const void *>(&lcluster_kernel<false>);
    hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return rc;
    const int grid = xpack ? ((ntrials + 7) / 8) * 8 * G : ntrials * G;
    const int gcap = lcluster_group_capacity(groups);
    if (glibc) hipLaunchKernelGGL(lcluster_kernel<true>, dim3(grid), dim3(WG_THREADS), lds, stream, params, G, lslots, xpack, ntrials, gcap);
    else hipLaunchKernelGGL(lcluster_kernel<false>, dim3(grid), dim3(WG_THREADS), lds, stream, params, G, lslots, xpack, ntrials, gcap);
    return hipGetLastError();
}

} // namespace prach
