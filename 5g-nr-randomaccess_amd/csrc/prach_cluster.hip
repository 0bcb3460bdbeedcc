// prach_cluster.hip — the production (Philox) trial kernel: a CLUSTER of G workgroups per trial.
//
// Why: one (seed, nUE) trial is 10 000 dependent subframes; a single workgroup (one CU) is bound by
// its own instruction issue over <=1563 wave-groups per subframe.  Here G workgroups (G CUs) share a
// trial; G = 1 degenerates to one workgroup per trial for batched sweeps (no exchange at all).
//
//  * Ownership is STATIC and INTERLEAVED: 64-UE group g belongs to workgroup g % G, wave
//    (g / G) % 16.  A UE's record is only ever touched by its owner CU, so UE state needs no
//    cross-CU coherence (per-XCD L2s are not coherent, MI355X_MICROARCH.md), and the live band of
//    UEs (arrived, not finished) is spread over all CUs at every moment.  Finished groups are
//    skipped through an LDS bitmap.
//  * ONE exchange per subframe: every workgroup publishes a small record (per-preamble histogram of
//    pre-members, lowest-index would-be caller per preamble, a header, the special events) into its mailbox as
//    8-byte SELF-VALIDATING GRANULES — {20-bit value | tag} twice, tag = subframe + 1, one write-through (sc1)
//    store each — and every workgroup gathers all mailboxes with relaxed sc1 loads, re-reading a granule until
//    its tag is the current one (cdna_hip_programming.md G16 / R2: no counter, no flag, no fence), then runs the
//    same resolver redundantly.  Mailboxes are double-buffered by subframe parity; every spin is bounded.
//  * Interleaving rules out index-ordered prefix sums, so the scan count of the first caller of a
//    bucket is computed set-wise: total - [caller is a pre-member] - #"early leavers" below it, where
//    only the early leavers below the workgroup's own lowest caller are published (validated against
//    the prefix formulation in oracle/phase_model.c).
//
//  * The Philox pass is COMPACTED (compact_phase_a / _b): a lean sweep settles "nothing to do" and the steady
//    contention cycle in place — without rewriting the record: a matched UE's state follows from the record's age —
//    and queues every other UE for the full per-UE body (ue_step), run 64 queued UEs at a time.
//  * A cluster PIPELINES subframes: phase A of subframe t+1 runs while the exchange of subframe t is in flight
//    (see the kernel's step loop); a UE that the resolver then grants is taken out of its bucket again.
//  * The glibc modes (the reference's own rand() stream) split the pass into count / exchange of the counts /
//    select; clusters run both halves compacted as well.
//
// Reference semantics: RandomAccessSimulatorBeta.c:111-197 / RandomAccessWithNOMA.c:267-351; the
// decomposition is DESIGN.md §3.
#include "prach_device.h"
#include "prach_device_fn.h"
#include "prach_ue_body.h"
#include <limits.h>

namespace prach {

namespace {

// Diagnostic build only (make DIAG=1 -> libprach_hip_diag.so): per-phase cycle shares of workgroup 0,
// written to DevResult.dbg (never to an output the simulation reads).  The shipped library has no stamps.
#ifdef PRACH_STAMPS
#define STAMP(k)                                                                                       \
    do {                                                                                               \
        if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); stamps[k] += now_ - tprev; tprev = now_; } \
    } while (0)
#define FSTAMP(k)                                                                                      \
    do {                                                                                               \
        if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); fstamps[k] += now_ - fprev; fprev = now_; } \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#define FSTAMP(k) do { } while (0)
#endif

constexpr int EVC_CALLER = UEV_CALLER, EVC_RESETCAND = UEV_RESETCAND, EVC_RJOIN = UEV_RJOIN, EVC_LEAVER = 4;
constexpr int EVCAPC = 4096; // gathered events per subframe held in LDS
constexpr int SCAPC = 2048;  // singleton callers per subframe held in LDS
constexpr int DEADW = 512;   // dead-group bitmap words (16384 local groups): one run of DEADW / NW words per wavefront, see dead_skip
constexpr unsigned SPIN_LIMIT = 1u << 22;

enum { C_NSUCC = 0, C_COLL, C_TXOP, C_CONTF, C_NS, C_NRC, C_NRJ, C_STATUS, C_NEV, C_NCAND, C_OVF, C_NSUCCTOT, C_NTOT,
       C_PTC, C_FC, C_SUMT = 16, C_ND = 18, C_NCROSS = 20, C_GTOT = 21, C_QN = 22, C_QEND = 23, C_VISITS = 24, C_EVENTS = 25, C_SX = 27 };

struct CLds {
    int2 *gev;    // [EVCAPC] gathered events of all workgroups
    int *sidx;    // [SCAPC]
    int *rclist;  // [RCCAP]
    int *scal;    // [64]
    unsigned *dead; // [DEADW]
    int *evoff;   // [MAXG+1]
    int *bins;    // [GBINS] grant selection: singleton callers per index bin (then exclusive prefix)
    int *wtot;    // [NW]
    int *hist, *mloc, *mloc_stay, *cand_n; // [2][nP] by subframe parity (the kernel hands the pass a view of one parity): pre-members, lowest
                               // caller (mloc_stay: among the UEs that were already matched in the previous subframe), early-leaver
                               // candidates per bucket of this workgroup
    int *total, *fcall, *lcall, *nlv, *fie; // [nP] each ([2][nP]: fcall, lcall)
    int *queue;   // [QCAP] Philox pass: UEs of this workgroup that have an event in this subframe
    int *stage;   // [NW][stage entries] global-record kernels: what a wavefront collects before it takes queue slots (compact_phase_a)
    int *gsum;    // [GSCAP] glibc mode: rand() calls of every 64-UE group in this subframe's UE loop
    int *gpre;    // [GSCAP] their exclusive prefix in index order
    unsigned *gmask; // [4 * GSCAP / 2] compacted glibc pass: per OWN group, lanes that draw once or twice (2 words) / twice (2 words)
    // LDS-resident clusters (REC_L16): the hot record and the Philox draw index of every owned UE (slot = local group * 64 + lane)
    int4 *lrec;      // [lslots]
    unsigned *lnd;   // [lslots]
    int2 *lcand;     // [LCANDCAP] early-leaver candidates of this subframe (more: global scratch)
};
constexpr int LCANDCAP = 1024;
constexpr int GBINS = 1024;
#ifndef PRACH_QCAP
#define PRACH_QCAP 8192
#endif
constexpr int QCAP = PRACH_QCAP; // event queue of the compacted pass (more: the overflowing wavefront works in place)
constexpr int GSCAP = 4096; // glibc mode on the cluster kernel: at most 4096 groups (262 144 UEs)
constexpr int MAXG = 64; // the gather's header phase is one wavefront: lane = workgroup

// Finished groups.  Local group j of a workgroup belongs to wavefront j % NW.  One workgroup per trial (G == 1: up to 1024
// groups per wavefront, and late in a trial or in a lightly loaded one nearly all of them finished): the bit sits in that
// wavefront's own run of words at position j / NW, so a wavefront walking its groups in order finds the next live one with
// one LDS read and a find-first-set per 32 groups.  A cluster (a few groups per wavefront): plain bit j, tested one by one
// (the shorter code path; it measured 1 % faster on the single-trial bench).
constexpr int DEADWW = DEADW / NW; // words per wavefront
__device__ __forceinline__ void dead_mark(const CLds &L, const int j, const int G) {
    if (G > 1) atomicOr(&L.dead[(j >> 5) & (DEADW - 1)], 1u << (j & 31));
    else atomicOr(&L.dead[(j % NW) * DEADWW + ((j / NW) >> 5)], 1u << ((j / NW) & 31));
}
// One workgroup per trial: a wavefront walks its own run of bitmap words forward, and only it ever writes them — so the word it
// is in stays in a scalar register (DeadCache) and the LDS is read once per 32 groups, not once per group visit (a dependent LDS
// round trip on the path of every one of the ~98 visits per wavefront and subframe at nUE = 100 000).
struct DeadCache { int widx; unsigned word; };
// next live local group >= j of wavefront j % NW whose global group (b + G * j) is below ngroups, or -1 (wave-uniform)
__device__ __forceinline__ int dead_skip(const CLds &L, const int j, const int b, const int G, const int ngroups, DeadCache &dc) {
    if (G > 1) {
        for (int jj = j;; jj += NW) {
            if (b + G * jj >= ngroups) return -1;
            const unsigned word = __builtin_amdgcn_readfirstlane(L.dead[(jj >> 5) & (DEADW - 1)]);
            if (!((word >> (jj & 31)) & 1u)) return jj;
        }
    }
    const int w = j & (NW - 1);
    int m = j / NW;
    if (j >= ngroups) return -1; // (G == 1, b == 0: local group == global group)
    // (the cached word and its index are wave-uniform and read back as such: kept in scalar registers, compared with s_cmp)
    if ((m >> 5) != __builtin_amdgcn_readfirstlane(dc.widx)) { dc.widx = m >> 5; dc.word = __builtin_amdgcn_readfirstlane(L.dead[w * DEADWW + (m >> 5)]); }
    unsigned word = __builtin_amdgcn_readfirstlane(dc.word);
    if (!((word >> (m & 31)) & 1u)) return j; // the common case while a trial is busy: the very next group is live
    for (;;) { // skip finished groups a word at a time
        const unsigned live = ~word >> (m & 31); // bit k: group m + k is live (zeros shifted in from the top = "not in this word")
        if (live) {
            m += __builtin_ctz(live);
            return w + NW * m >= ngroups ? -1 : w + NW * m;
        }
        m = (m | 31) + 1;
        if (w + NW * m >= ngroups) return -1;
        dc.widx = m >> 5;
        word = dc.word = __builtin_amdgcn_readfirstlane(L.dead[w * DEADWW + (m >> 5)]);
    }
}

// LDS layout: EVERY array sits at a compile-time offset (the per-bucket tables have the fixed stride NPC, whatever nPreamble
// is), so that an LDS address is an immediate in the ds_* instruction instead of a live scalar register: with run-time
// offsets the ~25 table pointers alone took a quarter of the wave's SGPR budget and the step loop spilled ~250 SGPRs
// (v_writelane / v_readlane on the critical path of every phase).  Only the tail of an LDS-resident cluster (records, draw
// indices: sized by the launch) is addressed through a run-time base.
constexpr int NPC = 256; // stride of the per-bucket tables (nPreamble <= 254)
constexpr int LQCAP = CLUSTER_LQCAP; // LDS-resident clusters: event queue = at most every owned UE slot
template <int NPC_, int QCAP_, int EVC_, int STG_>
struct LdsOff {
    static constexpr int GEV = 0;
    static constexpr int SIDX = GEV + 8 * EVC_;
    static constexpr int RCLIST = SIDX + 4 * SCAPC;
    static constexpr int SCAL = RCLIST + 4 * RCCAP;
    static constexpr int DEAD = SCAL + 4 * 64;
    static constexpr int EVOFF = DEAD + 4 * DEADW;
    static constexpr int BINS = EVOFF + 4 * (MAXG + 16);
    static constexpr int WTOT = BINS + 4 * GBINS;
    static constexpr int HIST = WTOT + 4 * NW;       // [2][NPC] by subframe parity, like the next three
    static constexpr int MLOC = HIST + 4 * 2 * NPC_;
    static constexpr int MLOC_STAY = MLOC + 4 * 2 * NPC_;
    static constexpr int CAND_N = MLOC_STAY + 4 * 2 * NPC_;
    static constexpr int TOTAL = CAND_N + 4 * 2 * NPC_;
    static constexpr int FCALL = TOTAL + 4 * NPC_;    // [2][NPC]
    static constexpr int LCALL = FCALL + 4 * 2 * NPC_; // [2][NPC]
    static constexpr int NLV = LCALL + 4 * 2 * NPC_;
    static constexpr int FIE = NLV + 4 * NPC_;
    static constexpr int QUEUE = FIE + 4 * NPC_;
    static constexpr int STAGE = QUEUE + 4 * QCAP_;   // global-record kernels: [NW][STG_] per-wavefront stage of the compacted pass (compact_phase_a)
    static constexpr int TAIL_G = STAGE + 4 * NW * STG_; // ... then (glibc) gsum, gpre, gmask
    static constexpr int LCAND = QUEUE + 4 * LQCAP;  // LDS-resident kernels: LQCAP-entry queue, candidate list, then the launch-sized tail
    static constexpr int TAIL_L = LCAND + 8 * LCANDCAP;
    static_assert(SIDX % 16 == 0 && TAIL_L % 16 == 0, "16-byte alignment of the event and record arrays");
};
// stage entries per wavefront: 64 left over from the round before + 64 per record slot of a round
template <int REC_, bool GLIBC_> struct CtxT;
using lds_off = LdsOff<NPC, QCAP, EVCAPC, 0>;      // (offsets in front of the stage do not depend on it)

template <class O>
__device__ __forceinline__ CLds ccarve(char *smem, bool glibc, int lslots) {
    CLds L;
    L.gev = reinterpret_cast<int2 *>(smem + O::GEV);
    L.sidx = reinterpret_cast<int *>(smem + O::SIDX);
    L.rclist = reinterpret_cast<int *>(smem + O::RCLIST);
    L.scal = reinterpret_cast<int *>(smem + O::SCAL);
    L.dead = reinterpret_cast<unsigned *>(smem + O::DEAD);
    L.evoff = reinterpret_cast<int *>(smem + O::EVOFF);
    L.bins = reinterpret_cast<int *>(smem + O::BINS);
    L.wtot = reinterpret_cast<int *>(smem + O::WTOT);
    L.hist = reinterpret_cast<int *>(smem + O::HIST); L.mloc = reinterpret_cast<int *>(smem + O::MLOC);
    L.mloc_stay = reinterpret_cast<int *>(smem + O::MLOC_STAY); L.cand_n = reinterpret_cast<int *>(smem + O::CAND_N);
    L.total = reinterpret_cast<int *>(smem + O::TOTAL); L.fcall = reinterpret_cast<int *>(smem + O::FCALL);
    L.lcall = reinterpret_cast<int *>(smem + O::LCALL); L.nlv = reinterpret_cast<int *>(smem + O::NLV); L.fie = reinterpret_cast<int *>(smem + O::FIE);
    L.queue = reinterpret_cast<int *>(smem + O::QUEUE);
    L.stage = reinterpret_cast<int *>(smem + O::STAGE); // (the LDS-resident layouts overlay this space and never use it)
    L.gsum = reinterpret_cast<int *>(smem + O::TAIL_G); L.gpre = L.gsum + GSCAP; L.gmask = reinterpret_cast<unsigned *>(L.gsum + 2 * GSCAP); // only touched in glibc mode
    (void)glibc;
    L.lrec = nullptr; L.lnd = nullptr; L.lcand = nullptr;
    if (lslots > 0) { // (never together with glibc mode)
        L.lcand = reinterpret_cast<int2 *>(smem + O::LCAND);
        L.lrec = reinterpret_cast<int4 *>(smem + O::TAIL_L);
        L.lnd = reinterpret_cast<unsigned *>(smem + O::TAIL_L + 16 * lslots);
    }
    return L;
}

// shared words: every access is a device-scope relaxed atomic == global_load/store ... sc1
__device__ __forceinline__ long long ld_sc1_64(const PRACH_G long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1_64(PRACH_G long long *p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// granule store of a cluster: write-through, or — once the cluster has VERIFIED that all its workgroups run on one XCD (handshake in the
// kernel prologue, as in prach_lcluster.hip) — resident in that XCD's L2, where the peers' sc1 loads find it an L2 round trip later
__device__ __forceinline__ void st_gr(const bool same_xcd, PRACH_G long long *p, long long v) {
    if (same_xcd) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct FastMods { FastMod nP, backoff, aT, five; };

// Event info word (20 bits): type[2:0] ispre[3] bucket p[11:4] old bucket q[19:12].
// Exchange granule: ONE naturally aligned 8-byte write-through store {20-bit value | tag[11:0]} {20-bit value | tag[15:12]}.
// Every granule carries the subframe tag (t+1 <= 60001 fits 16 bits), so it validates itself: the consumer
// re-reads until the tag matches — no drain, no flag, no fence (cdna_hip_programming.md G16, R2).
constexpr unsigned GR_NONE = 0xFFFFFu;
__device__ __forceinline__ long long mk_granule(unsigned lo20, unsigned hi20, unsigned tag) {
    const unsigned w0 = (lo20 & 0xFFFFFu) | ((tag & 0xFFFu) << 20), w1 = (hi20 & 0xFFFFFu) | (((tag >> 12) & 0xFu) << 20);
    return (long long)(((unsigned long long)w1 << 32) | w0);
}
__device__ __forceinline__ bool granule_ok(long long g, unsigned tag) {
    const unsigned w0 = (unsigned)g, w1 = (unsigned)((unsigned long long)g >> 32);
    return (w0 >> 20) == (tag & 0xFFFu) && ((w1 >> 20) & 0xFu) == ((tag >> 12) & 0xFu);
}
// bounded re-read of one granule until it carries `tag`
__device__ __forceinline__ long long wait_granule(const PRACH_G long long *p, unsigned tag, int *status_word) {
    long long g = ld_sc1_64(p);
    unsigned spins = 0;
    while (!granule_ok(g, tag)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { *status_word = PRACH_ERR_TIMEOUT; break; } // peer not resident? the engine reruns the trial
        g = ld_sc1_64(p);
    }
    return g;
}

// Where the hot records of a trial live:
//   REC_G16  global memory, 16 bytes per UE (clusters whose owned UEs do not fit LDS; the glibc modes; the dense pass)
//   REC_H8   global memory, 8 + 4 bytes per UE (one workgroup per trial: the streaming regime)
//   REC_L16  LDS: a cluster workgroup keeps the 16-byte record AND the Philox draw index of every UE it owns resident for the
//            whole trial (49 groups x 64 UEs x 20 bytes = 63 KB at nUE = 100 000, G = 32): the pass, the event body and the grant
//            never touch L2 for a record; global memory only sees the cold per-UE fields and the final state
constexpr int REC_G16 = 0, REC_H8 = 1, REC_L16 = 2;
template <int REC_, bool GLIBC_ = false>
struct CtxT {
    static constexpr int REC = REC_;
    static constexpr int QCAPX = QCAP;     // event queue entries
    static constexpr int EVCX = EVCAPC;    // gathered events held in LDS
    static constexpr bool H8 = REC_ == REC_H8;   // 8 + 4 byte hot record (one workgroup per trial, streaming)
    static constexpr bool LREC = REC_ == REC_L16; // LDS-resident records
    static constexpr int PFD = 2;          // record slots of the compacted pass's walk (the records of a cluster come from L2)
    // the compacted pass collects its event UEs on a per-wavefront stage (64 + 64 per record slot) before it takes queue slots for them —
    // where LDS has room for it: not beside LDS-resident records, not beside the glibc modes' 64 KB of group sums (158.6 of 160 KB already)
    static constexpr bool STAGED = REC_ != REC_L16 && !GLIBC_;
    static constexpr int STG = STAGED ? 64 + 64 * PFD : 0;
    int b, G, evw, mbstride;
    bool sx;             // the cluster shares one XCD (verified): granules stay in its L2
    FastMod fmG;         // LREC: slot of an owned UE = ((idx >> 6) - b) / G * 64 + (idx & 63)
    int4 *lrec;          // LREC: [lslots]
    unsigned *lnd;       // LREC: [lslots] Philox draw index
    int2 *lcand;         // LREC: [LCANDCAP]
    PRACH_G int *mbox;   // [2][G][mbstride ints]: per workgroup 1 header + nP bucket + evw event granules
    PRACH_G v2i_t *cand; // this workgroup's private early-leaver candidate scratch
    PRACH_G v4i_t *rec;  // [nUE] the 16-byte record (H8 == false)
    PRACH_G v2i_t *hot;  // [nUE] {packed word, (txTime + 1) | nowBackoff expiry << 16}: all the pass needs of a UE, 8 bytes
    PRACH_G int *tbase;  // [nUE] timer base (only read when something happens to the UE)
    int *status_word;    // LDS: a value that does not fit the 8-byte form stops the trial (the engine reruns it on trial_kernel)
};

template <class CX>
__device__ __forceinline__ PRACH_G long long *gr_of(const CX &C, int parity, int wg) {
    return reinterpret_cast<PRACH_G long long *>(C.mbox + ((size_t)parity * C.G + wg) * C.mbstride);
}

// With one workgroup per trial (the streaming, batched regime) the hot record (prach_device.h) is kept as 8 + 4 bytes inside
// the 16 bytes the engine lays out per UE: txTime and the backoff expiry are subframe numbers below 2^16 (the engine checks;
// longer trials keep the 16-byte form) and the timer base is not needed to decide what a UE does: half the bytes per UE
// visit.  A cluster (G > 1) keeps the 16-byte record: its state is L2-resident and one load per event is shorter than two.
constexpr int HOT_BO_BIAS = 2048; // nowBackoff is a literal <= 0 or an expiry subframe; a stale txTime can make the literal negative (Beta.c:266)
__device__ __forceinline__ int4 hot_decode(const v2i_t h, const int tb) {
    const unsigned w1 = (unsigned)h.y;
    return make_int4((int)(w1 & 0xffffu) - 1, tb, (int)(w1 >> 16) - HOT_BO_BIAS, h.x);
}
__device__ __forceinline__ v2i_t hot_encode(const int tx, const int bo, const int pk) {
    v2i_t h;
    h.x = pk;
    h.y = (int)(((unsigned)(tx + 1) & 0xffffu) | ((unsigned)(bo + HOT_BO_BIAS) << 16));
    return h;
}
__device__ __forceinline__ bool hot_fits(const int tx, const int bo) { return (unsigned)(tx + 1) <= 0xffffu && (unsigned)(bo + HOT_BO_BIAS) <= 0xffffu; }
// LREC: the LDS slot of UE i (owned by this workgroup: (i >> 6) - b is a multiple of G) and back
template <class CX>
__device__ __forceinline__ int slot_of(const CX &C, const int i) {
    const unsigned x = (unsigned)((i >> 6) - C.b);
    unsigned q = C.fmG.d == 1u ? x : __umulhi(x, C.fmG.M);
    q += (x - q * C.fmG.d) != 0u ? 1u : 0u; // the quotient estimate is at most 1 short; x is an exact multiple
    return (int)(q * 64u) + (i & 63);
}
template <class CX>
__device__ __forceinline__ int idx_of(const CX &C, const int slot) { return (C.b + C.G * (slot >> 6)) * 64 + (slot & 63); }

// (slot: only read when CX::LREC)
template <class CX>
__device__ __forceinline__ int4 hot_load_full(const CX &C, const int i, const int slot) { // non-temporal: the grant bit is set by an L2 atomic
    if (CX::LREC) return C.lrec[slot];
    if (CX::H8) return hot_decode(__builtin_nontemporal_load(C.hot + (unsigned)i), C.tbase[(unsigned)i]);
    return load_rec(C.rec + (unsigned)i);
}
template <class CX>
__device__ __forceinline__ void hot_store_full(const CX &C, const int i, const int slot, const int4 r) {
    if (CX::LREC) {
        C.lrec[slot] = r;
    } else if (CX::H8) {
        if (!hot_fits(r.x, r.z)) *C.status_word = PRACH_ERR_INTERNAL;
        C.hot[(unsigned)i] = hot_encode(r.x, r.z, r.w);
        C.tbase[(unsigned)i] = r.y;
    } else {
        store_rec(C.rec + (unsigned)i, r);
    }
}
template <class CX>
__device__ __forceinline__ void hot_grant(const CX &C, const int i, const int slot) {
    if (CX::LREC) atomicOr(reinterpret_cast<unsigned *>(&C.lrec[slot].w), PK_GRANT_BIT);
    else if (CX::H8) __hip_atomic_fetch_or(reinterpret_cast<PRACH_G unsigned *>(C.hot + (unsigned)i), PK_GRANT_BIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else grant_rec(C.rec + (unsigned)i);
}
// what phase A reads and writes: everything but the timer base
template <class CX>
__device__ __forceinline__ int4 hot_load_pass(const CX &C, const unsigned i, const int slot) {
    if (CX::LREC) return C.lrec[slot];
    if (CX::H8) return hot_decode(__builtin_nontemporal_load(C.hot + i), 0);
    return load_rec(C.rec + i);
}
template <class CX>
__device__ __forceinline__ void hot_store_pass(const CX &C, const unsigned i, const int slot, const int4 r) {
    if (CX::LREC) C.lrec[slot] = r;
    else if (CX::H8) C.hot[i] = hot_encode(r.x, r.z, r.w);
    else store_rec(C.rec + i, r);
}
template <class CX>
__device__ __forceinline__ unsigned nd_load(const TrialG &P, const CX &C, const int i, const int slot) { return CX::LREC ? C.lnd[slot] : P.nd[i]; }
template <class CX>
__device__ __forceinline__ void nd_store(const TrialG &P, const CX &C, const int i, const int slot, const unsigned v) {
    if (CX::LREC) C.lnd[slot] = v; else P.nd[i] = v;
}

// ---------------------------------------------------------------------------------------------
// One subframe of the UE loop for the 64 UEs a wavefront holds (lane = UE; the UEs need not be neighbours).
// MODE 0: fused apply + activate + select (Philox); glibc mode splits it: MODE 1 apply + activate + per-group
// draw counts, MODE 2 select with stream offsets (both need the lanes to be one 64-UE group in index order);
// MODE 3: only the deferred apply of the last subframe.  MODE 4 / 5: MODE 1 / 2 for UEs out of the event queue (any UE in
// any lane): a UE's draw count goes into its group's total and into two per-group lane masks, from which MODE 5 takes
// the index-ordered prefix inside the group.
// ---------------------------------------------------------------------------------------------
template <int MODE, class CX>
__device__ __forceinline__ void ue_step(const TrialG &P, const CLds &L, const CX &C, const FastMods &FM, const int *fcall, const int *lcall,
                                        const int t, const int prevAC, PRACH_G long long *mbev, const unsigned tag, const unsigned long long stepbase,
                                        const int lane, const int g, const int jdead, const int i, const int slot, const bool valid, const int4 r, unsigned ndc,
                                        int &c_succ, int &c_contf) {
    constexpr bool FINAL = MODE == 3, COUNT = MODE == 1 || MODE == 4, SELECT = MODE == 2 || MODE == 5;
    const int aT = P.aT, nUE = P.nUE;
    const bool withnoma = P.variant == PRACH_VARIANT_WITHNOMA_C;
    const int tp = t - 1;
    const UeK K{P.maxRarWindow, P.maxMsg2, aT, withnoma, FM.nP, FM.backoff, FM.aT, FM.five};
    ColdGlobal cold{P.ptc, P.ftt, P.stt, P.fcnt};
    bool nd_dirty = false;
    UeState u = unpack(r);
    bool dirty = false;

    // ---- deferred outcome of subframe t-1 (prach_ue_body.h ue_apply) ----
    if (!SELECT && u.pend != PEND_NONE) {
        // the compacted pass leaves a UE in steady contention untouched (see compact_phase_a): its record dates from
        // subframe u.tx, since when it has been bumped and has counted one more RAR-window subframe per subframe
        if (u.pend == PEND_STAY) { u.rar += tp - u.tx; u.tx = tp; }
        dirty = ue_apply(u, ((unsigned)r.w & PK_GRANT_BIT) != 0u, i, tp, FM.aT, CallTables{fcall, lcall});
    }
    if (FINAL) {
        if (dirty) hot_store_full(C, i, slot, pack(u));
        return;
    }
    // ---- activation (Beta.c:136-146; activateUEs WithNOMA:383-394 also draws twice) ----
    if (!SELECT && valid && i >= prevAC) {
        ue_activate(u, i, t, cold);
        if (MODE == 0 && withnoma) { ndc = 2; nd_dirty = true; }
        dirty = true;
    }

    const UePlan pl = ue_plan(u, t, P.maxRarWindow, P.maxMsg2);
    const int need = pl.need;

    if (COUNT) { // glibc: this group's rand() calls in the UE loop of subframe t (own pre-step state only)
        if (MODE == 1) {
            const int gs = __popcll(__ballot(need >= 1)) + __popcll(__ballot(need == 2));
            if (lane == 0) L.gsum[g] = gs;
        } else if (need >= 1) { // (i >> 6 is the UE's group, whatever lane it sits in)
            const int gi = i >> 6, jl = (gi - C.b) / C.G, ln = i & 63;
            atomicAdd(&L.gsum[gi], need);
            atomicOr(&L.gmask[4 * jl + (ln >> 5)], 1u << (ln & 31));
            if (need == 2) atomicOr(&L.gmask[4 * jl + 2 + (ln >> 5)], 1u << (ln & 31));
        }
        if (dirty) hot_store_full(C, i, slot, pack(u));
        return;
    }

    if (!__any(pl.busy || dirty)) {
        // nothing happens in this group; retire it for good once every UE in it has finished
        if (jdead >= 0 && __all(i >= nUE || u.act == ACT_DONE) && lane == 0) dead_mark(L, jdead, C.G);
        return;
    }

    int d1 = 0, d2 = 0;
    if (MODE == 5) { // the reference's own stream, UE out of the queue: prefix inside its group from the two lane masks
        if (need > 0) {
            const int gi = i >> 6, jl = (gi - C.b) / C.G, ln = i & 63;
            const unsigned lo = ln < 32 ? (1u << ln) - 1u : 0xffffffffu, hi = ln < 32 ? 0u : (1u << (ln - 32)) - 1u;
            const int before = __popc(L.gmask[4 * jl] & lo) + __popc(L.gmask[4 * jl + 1] & hi) + __popc(L.gmask[4 * jl + 2] & lo) +
                               __popc(L.gmask[4 * jl + 3] & hi);
            const unsigned long long o = stepbase + (unsigned long long)L.gpre[gi] + (unsigned long long)before;
            d1 = P.stream[o];
            if (need > 1) d2 = P.stream[o + 1];
        }
    } else if (MODE == 2) { // the reference's own stream: position = draws before this subframe's UE loop + index-ordered prefix
        if (__any(need > 0)) {
            const int x = wave_scan_incl(need);
            const unsigned long long o = stepbase + (unsigned long long)L.gpre[g] + (unsigned long long)(x - need);
            if (need > 0) d1 = P.stream[o];
            if (need > 1) d2 = P.stream[o + 1];
        }
    } else if (__any(need > 0)) {
        const unsigned k = ndc;
        d1 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k, (unsigned)nUE, (unsigned)P.variant);
        if (__any(need > 1))
            d2 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k + 1, (unsigned)nUE, (unsigned)P.variant);
        if (need > 0) { ndc = k + (unsigned)need; nd_dirty = true; }
    }

    // ---- selectPreamble / requestResourceAllocation on own state (prach_ue_body.h ue_select) ----
    const UeOut o = ue_select(u, pl, d1, d2, i, t, t % aT, K, cold, c_succ, c_contf);
    dirty = dirty || o.dirty;
    const int oldp = o.oldp, evtype = o.evtype, evp = o.evp;
    const bool member_pre = o.member_pre, eclass = o.eclass;

    // ---- bucket bookkeeping (workgroup-level LDS atomics) ----
    if (member_pre) atomicAdd(&L.hist[oldp], 1);
    if (u.pend == PEND_STAY) { if (__atomic_load_n(&L.mloc[oldp], __ATOMIC_RELAXED) > i) atomicMin(&L.mloc[oldp], i); }
    if (evtype == EVC_CALLER) atomicMin(&L.mloc[evp], i);
    {
        const unsigned long long em = __ballot(evtype != 0);
        if (em) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&L.scal[C_NEV], __popcll(em));
            base = __builtin_amdgcn_readlane(base, 0);
            if (evtype != 0) {
                const int slot = base + __popcll(em & lanemask_lt(lane));
                const int info = ue_event_info(o);
                if (C.G == 1) { if (slot < CX::EVCX) L.gev[slot] = make_int2(i, info); }
                else if (slot < C.evw) st_gr(C.sx, mbev + slot, mk_granule((unsigned)i, (unsigned)info, tag));
            }
        }
        const unsigned long long cm = __ballot(eclass);
        if (cm) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&L.scal[C_NCAND], __popcll(cm));
            base = __builtin_amdgcn_readlane(base, 0);
            if (eclass) {
                const int cs = base + __popcll(cm & lanemask_lt(lane));
                if (CX::LREC && cs < LCANDCAP) C.lcand[cs] = make_int2(i, oldp);
                else store_i2(&C.cand[cs], i, oldp);
                atomicAdd(&L.cand_n[oldp], 1);
            }
        }
    }
    if (nd_dirty) nd_store(P, C, i, slot, ndc);
    if (dirty) hot_store_full(C, i, slot, pack(u));
}

// ---------------------------------------------------------------------------------------------
// dense pass over the groups this workgroup owns (glibc modes, the final apply)
// ---------------------------------------------------------------------------------------------
template <int MODE, class CX>
__device__ __forceinline__ void cluster_pass(const TrialG &P, const CLds &L, const CX &C, const FastMods &FM, const int *fcall, const int *lcall,
                                             const int t, const int prevAC,
                                             const int activeCheck, PRACH_G long long *mbev, const unsigned tag, const unsigned long long stepbase) {
    constexpr bool FINAL = MODE == 3;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroups = (activeCheck + 63) >> 6;
    int c_succ = 0, c_contf = 0;

    // software pipeline: the next live groups' records (and Philox draw indices) are in flight while the
    // current group is processed
    DeadCache dc{-1, 0u};
    // (wave-uniform, and SAID so: without the readfirstlane the loop-carried group index is treated as divergent and the whole
    //  loop control — bitmap walk, range checks, exits — is emitted as vector compares and exec-mask branches)
    auto next_live = [&](const int jj) -> int { return __builtin_amdgcn_readfirstlane(dead_skip(L, jj, C.b, C.G, ngroups, dc)); };
    // two groups ahead (slots A, B; three ahead measured slower).  The load itself is unconditional (record 0 is always
    // mapped) and the "nothing there" case is applied where the record is consumed: a conditional load would have to be
    // waited for on the spot to merge it with the default value, which serialises the prefetches.
    auto fetch = [&](int jj, int4 &rr, unsigned &nn, bool &ok) {
        const int in = (C.b + C.G * max(jj, 0)) * 64 + lane;
        ok = jj >= 0 && in < activeCheck;
        const int sl = max(jj, 0) * 64 + lane;
        rr = hot_load_full(C, ok ? in : 0, sl);
        nn = MODE == 0 ? nd_load(P, C, ok ? in : 0, sl) : 0u;
    };
    int jA = next_live(w), jB;
    int4 rA, rB;
    unsigned ndA, ndB;
    bool okA, okB;
    fetch(jA, rA, ndA, okA);
    jB = jA >= 0 ? next_live(jA + NW) : -1;
    fetch(jB, rB, ndB, okB);
    while (jA >= 0) {
        const int j = jA;
        const int g = C.b + C.G * j;
        const int i = g * 64 + lane;
        const int4 r = okA ? rA : make_int4(-1, 0, 0, 0);
        const unsigned ndc = okA ? ndA : 0u;
        jA = jB; rA = rB; ndA = ndB; okA = okB;
        jB = jA >= 0 ? next_live(jA + NW) : -1;
        fetch(jB, rB, ndB, okB);
        ue_step<MODE>(P, L, C, FM, fcall, lcall, t, prevAC, mbev, tag, stepbase, lane, g, j, i, j * 64 + lane, i < activeCheck, r, ndc, c_succ, c_contf);
    }
    if (!FINAL) {
        c_succ = wave_sum(c_succ); c_contf = wave_sum(c_contf); // (DPP: prach_device_fn.h)
        if (lane == 0) {
            if (c_succ) atomicAdd(&L.scal[C_NSUCC], c_succ);
            if (c_contf) atomicAdd(&L.scal[C_CONTF], c_contf);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Philox pass, compacted.  In a 64-UE group nearly every subframe SOME UE has an event (a draw, an expiry, a
// deferred outcome), so the dense pass executes every branch of ue_step for every group at a few per cent lane
// utilisation — and the pass is instruction-issue bound (rocprofv3: >90 % issue-busy).  Here phase A walks the
// groups with a short straight-line body that handles the two cheap, overwhelmingly common cases in place —
// nothing to do, and the steady contention cycle "bumped last subframe, RAR window still open" (Beta.c:245 plus the
// txTime++ of Beta.c:346,358: rarWindow++, stay matched, count into the bucket) — and queues every other UE; phase B
// runs ue_step on the queued UEs 64 at a time, all lanes busy.  ue_step never looks at a lane's neighbours in MODE 0,
// so the result is the same whichever wavefront a UE lands in.
// ---------------------------------------------------------------------------------------------
// Is a UE whose deferred outcome is none / "matched, stays" / "called" (and not granted) in the light case of phase A at
// subframe t: contending with a RAR window that stays open?  A PEND_STAY record may be `age` subframes old (written at
// subframe r.x, not touched since): rarWindow has grown by age.  Shared by phase A and by the grant fix-up of the pipeline.
__device__ __forceinline__ bool light_case(const unsigned pk, const int rx, const int rz, const int t, const unsigned rarlim) {
    const unsigned pg = pk >> PK_PEND_SHIFT;
    const bool contend = (pk & 3u) == (unsigned)ACT_M1 && (pk & (0xffu << PK_PRE_SHIFT)) != 0u && rz <= t;
    const int age = pg == (unsigned)PEND_STAY ? t - 1 - rx : 0;
    const unsigned rarnow = (pk & (0xffu << PK_RAR_SHIFT)) + ((unsigned)age << PK_RAR_SHIFT);
    return pg < 3u && contend && rarnow < rarlim;
}

// Phase A of the compacted pass.  SPEC: for the NEXT subframe, while the exchange of the current one is in flight (see
// the kernel).
template <bool SPEC, class CX, class HOOK>
__device__ __forceinline__ void compact_phase_a(const TrialG &P, const CLds &L, const CX &C, const FastMods &FM, const int *fcall,
                                                const int *lcall, const int t, const int prevAC, const int activeCheck, PRACH_G long long *mbev,
                                                const unsigned tag, HOOK &&late_hook) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroups = (activeCheck + 63) >> 6;
    const int nUE = P.nUE, maxRar = P.maxRarWindow;
    int c_succ = 0, c_contf = 0;
    DeadCache dc{-1, 0u};
    // (wave-uniform, and SAID so: without the readfirstlane the loop-carried group index is treated as divergent and the whole
    //  loop control — bitmap walk, range checks, exits — is emitted as vector compares and exec-mask branches)
    auto next_live = [&](const int jj) -> int { return __builtin_amdgcn_readfirstlane(dead_skip(L, jj, C.b, C.G, ngroups, dc)); };
    // The prefetch is an unconditional load of an in-range record (lanes past the arrived UEs re-read the last one and
    // ignore it): nothing has to be merged with a default value, so nothing is waited for before the record is used.
    unsigned lastrec = (unsigned)(max(activeCheck, 1) - 1);
    asm volatile("" : "+v"(lastrec)); // (only ever a VALU operand: in a vector register it is not a spilled scalar reloaded by v_readlane per visit)
    auto fetch = [&](int jj) -> int4 { // H8: 8 bytes per UE, the timer base (.y) is not needed here
        return hot_load_pass(C, min((unsigned)((C.b + C.G * jj) * 64 + lane), lastrec), max(jj, 0) * 64 + lane);
    };
    unsigned rarlim = (unsigned)(maxRar - 1) << PK_RAR_SHIFT; // window still open after this subframe: rar + 1 < maxRarWindow
    asm volatile("" : "+v"(rarlim));
    constexpr bool STAGED = CX::STAGED;
    int *const stage = L.stage + w * CX::STG;
    int sc = 0; // staged UEs of this wavefront (wave-uniform)
    // ---- phase A ----
    auto phase_a = [&](const int j, const int4 r) __attribute__((always_inline)) {
        const int g = C.b + C.G * j;
        const int i = g * 64 + lane;
        const unsigned pk = (unsigned)r.w;
        const unsigned pg = pk >> PK_PEND_SHIFT; // deferred outcome | grant bit << 3
        // (ahead of the resolver of the previous subframe — SPEC — its grants are not known: a new caller (PEND_CALLER) may get one and
        //  waits for phase B; a matched UE (PEND_STAY) that gets one is taken out again by the granting thread: grant_fixup)
        const PassMasks M = pass_masks<SPEC>(pk, r.x, r.z, t, rarlim, g * 64 + 64 > prevAC, i, activeCheck, prevAC, nUE);
        const unsigned long long mHeavy = ~(M.light | M.quiet);
        if ((M.light | mHeavy) == 0ull) {
            // nothing happens in this group; retire it for good once every UE in it has finished
            if (M.done == ~0ull) {
                if (lane == 0) dead_mark(L, j, C.G);
                if (C.G == 1 && ((j / NW) >> 5) == __builtin_amdgcn_readfirstlane(dc.widx)) dc.word = __builtin_amdgcn_readfirstlane(dc.word | (1u << ((j / NW) & 31))); // (the register copy of this wavefront's word)
            }
            return;
        }
        const bool lightc = __builtin_amdgcn_inverse_ballot_w64(M.light), heavy = __builtin_amdgcn_inverse_ballot_w64(mHeavy);
        const bool trig = __builtin_amdgcn_inverse_ballot_w64(M.trig);
        if (lightc) { // Beta.c:245 + the txTime++ of Beta.c:346,358
            const bool bump = pg != 0u;
            const bool member = bump || trig; // matched by a preambleCollision scan in this subframe
            // steady contention (already PEND_STAY): bumped again, one more window subframe, still matched — all of it follows
            // from the record's age, so the record is NOT rewritten (ue_step brings it up to date when something happens)
            if (pg != (unsigned)PEND_STAY) {
                const unsigned npk = ((pk & 0x0FFFFFFFu) + (1u << PK_RAR_SHIFT)) | (member ? (unsigned)PEND_STAY << PK_PEND_SHIFT : 0u);
                hot_store_pass(C, (unsigned)i, j * 64 + lane, make_int4(bump ? t : r.x, r.y, r.z, (int)npk));
            }
            if (member) {
                const int p1 = (int)((pk >> PK_PRE_SHIFT) & 0xffu) - 1;
                atomicAdd(&L.hist[p1], 1);
                int *const ml = pg == (unsigned)PEND_STAY ? L.mloc_stay : L.mloc;
                atomicMin(&ml[p1], i); // (fire and forget: a load-compare in front of it is a dependent LDS round trip per group; config 3 +1.5 %)
            }
        }
        const unsigned long long hm = mHeavy;
        if (hm) {
            const int n = __builtin_amdgcn_readfirstlane(__popcll(hm));
            if (STAGED) { // global records: onto this wavefront's stage (the walk keeps room for a whole group in front of every round)
                if (heavy) stage[sc + __popcll(hm & lanemask_lt(lane))] = i;
                sc += n;
            } else {      // no stage: queue slots per visit
                int base = 0;
                if (lane == 0) base = atomicAdd(&L.scal[C_QN], n);
                base = __builtin_amdgcn_readlane(base, 0);
                if (CX::LREC || base + n <= CX::QCAPX) { // (LDS-resident: the queue holds slots and has room for every owned UE)
                    if (heavy) L.queue[base + __popcll(hm & lanemask_lt(lane))] = CX::LREC ? j * 64 + lane : i;
                } else if (SPEC) { // cannot happen: the kernel runs ahead only if every owned UE fits the queue
                    if (lane == 0) L.scal[C_STATUS] = PRACH_ERR_INTERNAL;
                } else { // queue full: this wavefront does its events in place
                    if (lane == 0) atomicMin(&L.scal[C_QEND], base);
                    unsigned ndc = 0;
                    int4 rf = make_int4(-1, 0, 0, 0);
                    if (heavy) { ndc = P.nd[i]; rf = CX::H8 ? make_int4(r.x, C.tbase[(unsigned)i], r.z, r.w) : r; }
                    ue_step<0>(P, L, C, FM, fcall, lcall, t, prevAC, mbev, tag, 0ull, lane, g, -1, i, 0, heavy, rf, ndc, c_succ, c_contf);
                }
            }
        }
    };
    // The stage goes to the workgroup's event queue in one piece — one returning LDS atomic per ~100 event UEs instead of one per group
    // visit: the slot allocation was a quarter of a visit's instructions, in a pass bound by instruction issue.  Queue full (the peak of
    // an overloaded 100 000-UE trial on one workgroup): this wavefront runs the full body on its staged UEs right here, gathered by index
    // as phase B would — ONE inlined copy of the body, outside the unrolled walk (it used to sit inside every record slot).
    auto flush = [&]() __attribute__((always_inline)) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&L.scal[C_QN], sc);
        base = __builtin_amdgcn_readlane(base, 0);
        const bool room = base + sc <= CX::QCAPX;
        if (!room && !SPEC && lane == 0) atomicMin(&L.scal[C_QEND], base);
        if (!room && SPEC && lane == 0) L.scal[C_STATUS] = PRACH_ERR_INTERNAL; // cannot happen: the kernel runs ahead only if every owned UE fits the queue
        for (int o = 0; o < sc; o += 64) {
            const bool v = o + lane < sc;
            const int qe = v ? stage[o + lane] : 0;
            if (room) { if (v) L.queue[base + o + lane] = qe; }
            else if (!SPEC) {
                int4 rf = make_int4(-1, 0, 0, 0);
                unsigned ndc = 0;
                if (v) { rf = hot_load_full(C, qe, 0); ndc = nd_load(P, C, qe, 0); }
                ue_step<0>(P, L, C, FM, fcall, lcall, t, prevAC, mbev, tag, 0ull, lane, qe >> 6, -1, qe, 0, v, rf, ndc, c_succ, c_contf);
            }
        }
        sc = 0;
    };
    int nvisit = 0;
    // PFD record slots, refilled round-robin: the records of the next PFD - 1 live groups are in flight while one is worked on.  With one
    // workgroup per trial the pass streams from HBM (1000 trials x 5 MB) and Little's law is the bound: 16 wavefronts x (PFD - 1) x 512 B
    // in flight per CU against a loaded latency of 1-2 us; with a cluster the records come from L2 and two slots are enough.
    constexpr int PFD = CX::PFD;
    static_assert(!STAGED || CX::STG >= 64 + 64 * PFD, "a round of the walk fits the stage behind a leftover of at most 64");
    int jq[PFD];
    int4 rq[PFD];
    jq[0] = next_live(w);
    rq[0] = fetch(jq[0]);
#pragma unroll
    for (int d = 1; d < PFD; d++) { jq[d] = jq[d - 1] >= 0 ? next_live(jq[d - 1] + NW) : -1; rq[d] = fetch(jq[d]); }
    bool hooked = false; // late_hook runs once: after this wavefront's first two groups, or at the end if it has fewer
    for (bool more = true;;) {
        if (STAGED && sc > 0 && (sc > 64 || !more)) flush(); // (the ONE place the stage is emptied)
        if (!more) break;
#pragma unroll
        for (int d = 0; d < PFD; d++) {
            if (jq[d] < 0) { more = false; break; }
            const int ja = jq[d];
            const int4 ra = rq[d];
            const int jlast = jq[(d + PFD - 1) % PFD]; // the group fetched last
            jq[d] = jlast >= 0 ? next_live(jlast + NW) : -1;
            rq[d] = fetch(jq[d]);
            phase_a(ja, ra); nvisit++;
            if (d == 1 && !hooked) { late_hook(); hooked = true; }
        }
    }
    if (!hooked) late_hook();
    if (lane == 0 && nvisit) atomicAdd(&L.scal[C_VISITS], nvisit); // (reported, never read by the simulation)
    if (!SPEC && __any((c_succ | c_contf) != 0)) { // (only the in-place overflow path counts here)
        c_succ = wave_sum(c_succ); c_contf = wave_sum(c_contf); // (DPP: prach_device_fn.h)
        if (lane == 0) {
            if (c_succ) atomicAdd(&L.scal[C_NSUCC], c_succ);
            if (c_contf) atomicAdd(&L.scal[C_CONTF], c_contf);
        }
    }
}

// Phase B: the queued UEs through the full body, 64 at a time, all wavefronts (MODE 0: Philox; 4 / 5: the two glibc passes).
template <int MODE, class CX>
__device__ __forceinline__ void compact_phase_b(const TrialG &P, const CLds &L, const CX &C, const FastMods &FM, const int *fcall,
                                                const int *lcall, const int t, const int prevAC, PRACH_G long long *mbev, const unsigned tag,
                                                const unsigned long long stepbase) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int c_succ = 0, c_contf = 0;
    // ---- phase B ----
    const int qn = CX::LREC ? L.scal[C_QN] : min(L.scal[C_QN], L.scal[C_QEND]);
    for (int q0 = w * 64; q0 < qn; q0 += NW * 64) {
        const bool v = q0 + lane < qn;
        const int qe = v ? L.queue[q0 + lane] : 0;
        const int slot = CX::LREC ? qe : 0, i = CX::LREC ? idx_of(C, qe) : qe;
        int4 r = make_int4(-1, 0, 0, 0);
        unsigned ndc = 0;
        if (v) { r = hot_load_full(C, i, slot); if (MODE == 0) ndc = nd_load(P, C, i, slot); }
        ue_step<MODE>(P, L, C, FM, fcall, lcall, t, prevAC, mbev, tag, stepbase, lane, i >> 6, -1, i, slot, v, r, ndc, c_succ, c_contf);
    }
    if (__any((c_succ | c_contf) != 0)) { // (most wavefronts, most subframes: nothing to add)
        c_succ = wave_sum(c_succ); c_contf = wave_sum(c_contf); // (DPP: prach_device_fn.h)
        if (lane == 0) {
            if (c_succ) atomicAdd(&L.scal[C_NSUCC], c_succ);
            if (c_contf) atomicAdd(&L.scal[C_CONTF], c_contf);
        }
    }
}


// One gathered event against the lowest DEFINITE caller of every bucket (complete after round 1).
__device__ __forceinline__ void classify_event(const CLds &L, int *fcallA, const int k, const int2 ev) {
    const int type = ev.y & 7, p = (ev.y >> 4) & 0xff;
    if (type == EVC_RESETCAND) {
        // a call on its old bucket by a definite caller with a lower index bumps it: cannot re-join (99.7 % of
        // them); only the survivors need the index-ordered treatment
        if (fcallA[(ev.y >> 12) & 0xff] < ev.x) L.gev[k].y = 0;
        else { const int s = atomicAdd(&L.scal[C_NRC], 1); if (s < RCCAP) L.rclist[s] = k; }
    } else if (type == EVC_RJOIN) {
        atomicAdd(&L.scal[C_NRJ], 1);
    } else if (type == EVC_LEAVER) {
        if (ev.x < fcallA[p]) atomicAdd(&L.nlv[p], 1);
    } else if (type == EVC_CALLER) {
        if (ev.x == fcallA[p]) L.fie[p] = 1;
    }
}

// Reset-cycle candidates (Beta.c:250-281 with tmp == 0 on a subframe = 1 mod accessTime): candidate i
// re-joins (and calls on its NEW preamble) iff nobody called on its OLD preamble before it, and a
// re-join is itself a call that later candidates must see.  Inherently sequential in index order,
// but tiny: ONE wavefront keeps the per-bucket first-caller table in registers (lane = bucket, up to
// 4 x 64 buckets) and walks the index-sorted candidates with v_readlane — no LDS round trip per step.
__device__ __forceinline__ int fc_get(int f0, int f1, int f2, int f3, int q) {
    const int l = q & 63;
    switch (q >> 6) {
    case 0: return __builtin_amdgcn_readlane(f0, l);
    case 1: return __builtin_amdgcn_readlane(f1, l);
    case 2: return __builtin_amdgcn_readlane(f2, l);
    default: return __builtin_amdgcn_readlane(f3, l);
    }
}
__device__ __forceinline__ void resolve_reset_candidates(const CLds &L, int *fcall, const int nrc_in, const int nP) {
    const int lane = threadIdx.x & 63;
    const int n = __builtin_amdgcn_readfirstlane(nrc_in);
    int f0 = lane < nP ? fcall[lane] : INT_MAX, f1 = lane + 64 < nP ? fcall[lane + 64] : INT_MAX,
        f2 = lane + 128 < nP ? fcall[lane + 128] : INT_MAX, f3 = lane + 192 < nP ? fcall[lane + 192] : INT_MAX;
    // rank-sort the candidate list by UE index into L.sidx (free at this point of the subframe)
    for (int c = lane; c < n; c += 64) {
        const int myidx = L.gev[L.rclist[c]].x;
        int rank = 0;
        for (int j = 0; j < n; j++) rank += L.gev[L.rclist[j]].x < myidx ? 1 : 0;
        L.sidx[rank] = L.rclist[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int base = 0; base < n; base += 64) {
        const int m = min(64, n - base);
        int slot = 0, cidx = 0, cinfo = 0;
        if (lane < m) { slot = L.sidx[base + lane]; const int2 e = L.gev[slot]; cidx = e.x; cinfo = e.y; }
        int cancelled = 0;
        for (int s_ = 0; s_ < m; s_++) {
            const int idx = __builtin_amdgcn_readlane(cidx, s_), info = __builtin_amdgcn_readlane(cinfo, s_);
            const int p = (info >> 4) & 0xff, q = (info >> 12) & 0xff;
            if (fc_get(f0, f1, f2, f3, q) < idx) { // bumped before its turn: does not re-join
                if (lane == s_) cancelled = 1;
            } else if (idx < fc_get(f0, f1, f2, f3, p)) { // its call becomes the first one on p
                if (lane == (p & 63)) {
                    switch (p >> 6) { case 0: f0 = idx; break; case 1: f1 = idx; break; case 2: f2 = idx; break; default: f3 = idx; break; }
                }
            }
        }
        if (lane < m && cancelled) L.gev[slot].y = 0;
    }
    if (lane < nP) fcall[lane] = f0;
    if (lane + 64 < nP) fcall[lane + 64] = f1;
    if (lane + 128 < nP) fcall[lane + 128] = f2;
    if (lane + 192 < nP) fcall[lane + 192] = f3;
}

} // namespace

// ---------------------------------------------------------------------------------------------
template <bool GLIBC, int REC>
__global__ __launch_bounds__(WG_THREADS, 4) void cluster_kernel(const TrialDev *__restrict__ params, const int Garg, const int lslots, const int xpack, const int ntrials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using LO = LdsOff<NPC, QCAP, EVCAPC, CtxT<REC, GLIBC>::STG>;
    // the 8 + 4 byte record form is only ever launched with ONE workgroup per trial (the streaming regime): there the cluster size is a
    // compile-time 1 and every exchange / mailbox / pipeline path of this kernel is dead code the compiler drops (half the code, fewer
    // live scalars in the pass)
    const int G = REC == REC_H8 ? 1 : Garg;
    // the workgroups of a cluster are CONSECUTIVE blocks: in-order dispatch completes whole clusters even when not every
    // block of the grid is resident at once (the engine keeps G x trials within the occupancy query's answer anyway)
    int T = blockIdx.x / G, b = blockIdx.x % G;
    if (REC != REC_H8 && xpack) { // XCD-packed launch (prach_lcluster.hip): a cluster = the blocks of equal blockIdx % 8 of a chunk of 8 G blocks
        const int chunk = blockIdx.x / (8 * G), within = blockIdx.x % (8 * G);
        T = chunk * 8 + (within & 7); b = within >> 3;
        if (T >= ntrials) return;
    }
    const TrialG P(params[T]);
    const CLds L = ccarve<LO>(smem, GLIBC, REC == REC_L16 ? lslots : 0);
    const int tid = threadIdx.x;
    const int nUE = P.nUE, nP = P.nP, aT = P.aT;
    const bool withnoma = P.variant == PRACH_VARIANT_WITHNOMA_C;

    FastMods FM;
    FM.nP = make_fastmod(P.nP); FM.backoff = make_fastmod(P.backoff); FM.aT = make_fastmod(P.aT); FM.five = make_fastmod(5);
    CtxT<REC, GLIBC> C;
    C.b = b; C.G = G; C.evw = P.evw; C.mbstride = P.mbstride; C.mbox = P.mbox; C.sx = false;
    C.fmG = make_fastmod(G); C.lrec = L.lrec; C.lnd = L.lnd; C.lcand = L.lcand;
    C.status_word = &L.scal[C_STATUS];
    C.rec = P.rec;
    C.hot = reinterpret_cast<PRACH_G v2i_t *>(P.rec);
    C.tbase = reinterpret_cast<PRACH_G int *>(P.rec) + 2 * (size_t)P.nUE;
    const int totgroups = (nUE + 63) >> 6;
    const int lgroups = (totgroups + G - 1) / G; // local groups of any workgroup (upper bound)
    C.cand = P.cand + (size_t)b * lgroups * 64;

    // calloc + initialUE (Beta.c:78-83) for the groups this workgroup owns
    for (int x = tid; x < lgroups * 64; x += WG_THREADS) {
        const int g = b + G * (x >> 6), i = g * 64 + (x & 63);
        if (REC == REC_L16) { L.lrec[x] = make_int4(-1, 0, 0, 0); L.lnd[x] = 0u; } // (every slot, also past the last UE)
        if (g < totgroups && i < nUE) {
            if (REC != REC_L16) { hot_store_full(C, i, 0, make_int4(-1, 0, 0, 0)); P.nd[i] = 0; }
            P.ptc[i] = 0; P.ftt[i] = 0; P.stt[i] = 0; P.fcnt[i] = 0;
        }
    }
    for (int k = tid; k < nP; k += WG_THREADS) {
        L.hist[k] = 0; L.mloc[k] = INT_MAX; L.mloc_stay[k] = INT_MAX; L.cand_n[k] = 0;
        L.hist[NPC + k] = 0; L.mloc[NPC + k] = INT_MAX; L.mloc_stay[NPC + k] = INT_MAX; L.cand_n[NPC + k] = 0;
        L.total[k] = 0; L.fcall[k] = INT_MAX; L.lcall[k] = -1; L.fcall[NPC + k] = INT_MAX; L.lcall[NPC + k] = -1;
        L.nlv[k] = 0; L.fie[k] = 0;
    }
    if (tid < 64) L.scal[tid] = tid == C_QEND ? QCAP : 0;
    for (int k = tid; k < DEADW; k += WG_THREADS) L.dead[k] = 0;
    if (GLIBC) for (int k = tid; k < GSCAP; k += WG_THREADS) { L.gsum[k] = 0; L.gpre[k] = 0; }
    __syncthreads();
    if (REC != REC_H8 && xpack && G > 1 && G <= 64) {
        // same-XCD handshake (prach_lcluster.hip): every workgroup publishes the XCD it runs on (write-through granule, tag 0xFFFF, header of
        // its parity-1 mailbox: first used by subframe 1, which nobody reaches before every peer is past this point) and reads all G
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xfu;
        if (tid == 0) st_sc1_64(gr_of(C, 1, b), mk_granule(xcc, 0u, 0xFFFFu));
        if (tid < 64) {
            bool same = true;
            if (tid < G) same = ((unsigned)wait_granule(gr_of(C, 1, tid), 0xFFFFu, &L.scal[C_STATUS]) & 0xFFFFFu) == xcc;
            const bool all = __ballot(!same) == 0ull;
            if (tid == 0) L.scal[C_SX] = all ? 1 : 0;
        }
        __syncthreads();
        C.sx = L.scal[C_SX] != 0 && L.scal[C_STATUS] == PRACH_OK;
    }

    int activeCheck = 0, grantCheck = 0, tlast = -1, time_exit = P.stop;
    unsigned long long steps = 0;
    int status = (lgroups > DEADW * 32 || (GLIBC && totgroups > GSCAP) || (REC == REC_L16 && (lgroups * 64 > lslots || lslots > LQCAP)) || nP > NPC || P.stop >= 0xFFFE) ? PRACH_ERR_UNSUPPORTED : PRACH_OK;
    unsigned long long base = 0; // glibc: rand() calls consumed so far (relative to the stream window)
#ifdef PRACH_STAMPS
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
    unsigned long long fstamps[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fprev = tprev;
    unsigned long long statN = 0, maxN = 0, statRC = 0, statNS = 0, maxNS = 0;
#endif

    // Software pipeline of a cluster (G > 1, Philox): phase A of subframe t+1 runs on half of the wavefronts while the other
    // half gathers the exchange of subframe t — it needs nothing of that exchange's outcome except the grants, and the UEs a
    // grant could go to are left for phase B (compact_phase_a<true>).  Needs every owned UE to fit the event queue.
    const bool pipelined = !GLIBC && !P.dense_pass && P.pipeline && G > 1 && (REC == REC_L16 || lgroups * 64 <= QCAP);
    int ahead_for = -1; // subframe whose phase A has already run
    auto parity_view = [&](const int par) { CLds V = L; V.hist = L.hist + par * NPC; V.mloc = L.mloc + par * NPC; V.mloc_stay = L.mloc_stay + par * NPC; V.cand_n = L.cand_n + par * NPC; return V; };

    const unsigned rarlim_k = (unsigned)(P.maxRarWindow - 1) << PK_RAR_SHIFT;

    for (int t = 0; t < P.stop && status == PRACH_OK; t++) {
        steps++;
        tlast = t;
        if (t % 5 == 0) grantCheck = 0; // Beta.c:112 (hard-coded 5)
        const int prevAC = activeCheck;
        if (t % aT == 0 && activeCheck != nUE) activeCheck = P.sched[t / aT]; // Beta.c:121-134
        const int parity = t & 1;
        const unsigned tag = (unsigned)(t + 1);
        PRACH_G long long *const mygr = G > 1 ? gr_of(C, parity, b) : nullptr;
        PRACH_G long long *const mbev = G > 1 ? mygr + 1 + nP : nullptr;
        // first / last caller tables are double-buffered by subframe parity: this subframe's resolver fills [A],
        // the pass (apply of the previous subframe) reads [B]
        int *const fcallA = L.fcall + parity * NPC, *const lcallA = L.lcall + parity * NPC;
        int *const fcallB = L.fcall + (parity ^ 1) * NPC, *const lcallB = L.lcall + (parity ^ 1) * NPC;
        const CLds Lc = parity_view(parity), Ln = parity_view(parity ^ 1); // per-bucket counts of this subframe / of the next one
        // An UL grant: the grant bit into the UE's record.  If phase A of subframe t+1 has already run it took a matched UE
        // (PEND_STAY) for steadily contending, counted it into its bucket and did not queue it: take it out again and queue
        // it.  A granted UE was the only caller of a bucket nobody else stayed matched in, so it is the only one that phase A
        // counted there as matched before (mloc_stay).  (Every owned UE fits the queue when the kernel runs ahead.)
        auto grant = [&](const int my) {
            const int myslot = REC == REC_L16 ? slot_of(C, my) : 0;
            hot_grant(C, my, myslot);
            if (ahead_for == t + 1) {
                const int4 r = hot_load_pass(C, (unsigned)my, myslot);
                const unsigned pk = (unsigned)r.w & ~PK_GRANT_BIT;
                if ((pk >> PK_PEND_SHIFT) == (unsigned)PEND_STAY && light_case(pk, r.x, r.z, t + 1, rarlim_k)) {
                    const int p1 = (int)((pk >> PK_PRE_SHIFT) & 0xffu) - 1;
                    atomicSub(&Ln.hist[p1], 1);
                    Ln.mloc_stay[p1] = INT_MAX;
                    L.queue[atomicAdd(&L.scal[C_QN], 1)] = REC == REC_L16 ? myslot : my; // ... and its grant is applied by phase B of subframe t+1
                }
            }
        };

        if (GLIBC) {
            // glibc mode: every draw's position in the reference's rand() stream = draws of earlier subframes + the
            // activation draws of this one (WithNOMA:393-394) + an index-ordered prefix of per-UE draw counts.
            // Pass X counts per 64-UE group; the counts of all groups are exchanged (2 per granule) and scanned.
            const unsigned long long actdraws = withnoma ? 2ull * (unsigned long long)(activeCheck - prevAC) : 0ull;
            const int ngroups_t = (activeCheck + 63) >> 6;
            // compacted (clusters whose owned UEs all fit the event queue): phase A settles the UEs that draw nothing, the
            // queued ones are counted here (MODE 4) and select behind the count exchange (MODE 5)
            const bool gcompact = G > 1 && !P.dense_pass && lgroups * 64 <= QCAP;
            if (gcompact) {
                for (int x = tid; x < lgroups; x += WG_THREADS) {
                    const int g = b + G * x;
                    if (g < totgroups) L.gsum[g] = 0;
                    L.gmask[4 * x] = 0; L.gmask[4 * x + 1] = 0; L.gmask[4 * x + 2] = 0; L.gmask[4 * x + 3] = 0;
                }
                compact_phase_a<false>(P, Lc, C, FM, fcallB, lcallB, t, prevAC, activeCheck, mbev, tag, [] {});
                __syncthreads(); // the queue is complete, the counts are zeroed
                compact_phase_b<4>(P, Lc, C, FM, fcallB, lcallB, t, prevAC, mbev, tag, 0ull);
            } else {
                cluster_pass<1>(P, Lc, C, FM, fcallB, lcallB, t, prevAC, activeCheck, mbev, tag, 0ull);
            }
            __syncthreads();
            if (G > 1) {
                const int nq = (lgroups + 1) >> 1;
                PRACH_G long long *const mine = mygr + 1 + nP + C.evw;
                for (int q = tid; q < nq; q += WG_THREADS) {
                    const int g0 = b + G * (2 * q), g1 = b + G * (2 * q + 1);
                    st_gr(C.sx, mine + q, mk_granule(g0 < totgroups ? (unsigned)L.gsum[g0] : 0u, g1 < totgroups ? (unsigned)L.gsum[g1] : 0u, tag));
                }
                for (int k = tid; k < G * nq; k += WG_THREADS) {
                    const int wg = k / nq, q = k - wg * nq;
                    if (wg == b) continue;
                    const long long g_ = wait_granule(gr_of(C, parity, wg) + 1 + nP + C.evw + q, tag, &L.scal[C_STATUS]);
                    const int g0 = wg + G * (2 * q), g1 = wg + G * (2 * q + 1);
                    if (g0 < totgroups) L.gsum[g0] = (int)((unsigned)g_ & 0xFFFFFu);
                    if (g1 < totgroups) L.gsum[g1] = (int)((unsigned)((unsigned long long)g_ >> 32) & 0xFFFFFu);
                }
                __syncthreads();
            }
            { // block-wide exclusive prefix over the arrived groups (4 consecutive groups per thread)
                int v[4], sum = 0;
#pragma unroll
                for (int u_ = 0; u_ < 4; u_++) { const int g = tid * 4 + u_; v[u_] = g < ngroups_t ? L.gsum[g] : 0; sum += v[u_]; }
                const int x = wave_scan_incl(sum);
                if ((tid & 63) == 63) L.wtot[tid >> 6] = x;
                __syncthreads();
                int add = 0;
                for (int k = 0; k < (tid >> 6); k++) add += L.wtot[k];
                int run = x - sum + add;
#pragma unroll
                for (int u_ = 0; u_ < 4; u_++) { const int g = tid * 4 + u_; if (g < GSCAP) L.gpre[g] = run; run += v[u_]; }
                if (tid == WG_THREADS - 1) L.scal[C_GTOT] = run;
            }
            __syncthreads();
            if (L.scal[C_STATUS] != PRACH_OK) { status = L.scal[C_STATUS]; time_exit = t; break; }
            const unsigned long long tot = actdraws + (unsigned long long)L.scal[C_GTOT];
            if (base + tot > P.stream_len) { status = PRACH_ERR_STREAM; time_exit = t; break; } // engine retries with a larger window
            if (gcompact) compact_phase_b<5>(P, Lc, C, FM, fcallB, lcallB, t, prevAC, mbev, tag, base + actdraws);
            else cluster_pass<2>(P, Lc, C, FM, fcallB, lcallB, t, prevAC, activeCheck, mbev, tag, base + actdraws);
            base += tot;
        } else {
            FSTAMP(0); // loop head
            if (P.dense_pass) {
                cluster_pass<0>(P, Lc, C, FM, fcallB, lcallB, t, prevAC, activeCheck, mbev, tag, 0ull);
            } else {
                if (ahead_for != t) {
                    compact_phase_a<false>(P, Lc, C, FM, fcallB, lcallB, t, prevAC, activeCheck, mbev, tag, [] {});
                    FSTAMP(15); // phase A (not run ahead)
                    __syncthreads(); // the queue is complete
                    FSTAMP(16); // its barrier
                }
                compact_phase_b<0>(P, Lc, C, FM, fcallB, lcallB, t, prevAC, mbev, tag, 0ull);
            }
            FSTAMP(1); // phase B body
        }
        __syncthreads(); // S1: histogram / lowest callers / candidate list of this workgroup are complete; [B] is free
        FSTAMP(2); // S1
        STAMP(0);

        // early leavers below this workgroup's lowest caller are the only ones a rank can need
        const int ncand = L.scal[C_NCAND];
        for (int k = tid; k < ncand; k += WG_THREADS) {
            v2i_t c;
            if (REC == REC_L16 && k < LCANDCAP) { const int2 cl = L.lcand[k]; c.x = cl.x; c.y = cl.y; }
            else c = C.cand[k];
            if (c.x < min(Lc.mloc[c.y], Lc.mloc_stay[c.y])) {
                const int slot = atomicAdd(&L.scal[C_NEV], 1);
                const int info = EVC_LEAVER | (c.y << 4);
                if (G == 1) { if (slot < EVCAPC) L.gev[slot] = make_int2(c.x, info); }
                else if (slot < C.evw) st_gr(C.sx, mbev + slot, mk_granule((unsigned)c.x, (unsigned)info, tag));
            }
        }
        for (int k = tid; k < nP; k += WG_THREADS) { fcallB[k] = INT_MAX; lcallB[k] = -1; L.total[k] = 0; L.nlv[k] = 0; L.fie[k] = 0; } // ready for the gathers
        if (tid == 0) {
            L.scal[C_EVENTS] += min(L.scal[C_QN], (REC == REC_L16 ? INT_MAX : L.scal[C_QEND])); // (reported, never read by the simulation)
            L.scal[C_NS] = 0; L.scal[C_NRC] = 0; L.scal[C_NRJ] = 0; L.scal[C_QN] = 0; L.scal[C_QEND] = QCAP; // (the queue has been consumed)
        }
        FSTAMP(3); // leaver filter + table reset
        __syncthreads(); // S2
        FSTAMP(4); // S2
        int N;
        if (G == 1) {
            // one workgroup owns the whole trial: its histogram / lowest callers ARE the totals; events are in LDS
            const int nevraw = L.scal[C_NEV];
            for (int k = tid; k < nP; k += WG_THREADS) {
                L.total[k] = Lc.hist[k]; fcallA[k] = min(Lc.mloc[k], Lc.mloc_stay[k]);
                Lc.hist[k] = 0; Lc.mloc[k] = INT_MAX; Lc.mloc_stay[k] = INT_MAX; Lc.cand_n[k] = 0;
            }
            if (nevraw > EVCAPC) { status = PRACH_ERR_INTERNAL; time_exit = t; break; } // engine falls back to trial_kernel
            N = nevraw;
            __syncthreads(); // S3
            if (tid == 0) { L.scal[C_NEV] = 0; L.scal[C_NCAND] = 0; L.scal[C_NSUCCTOT] = L.scal[C_NSUCC]; }
            STAMP(1); STAMP(2);
            // classify the events: reset-cycle candidates, Msg3 re-entries, early leavers below / callers at the first call
            for (int k = tid; k < N; k += WG_THREADS) classify_event(L, fcallA, k, L.gev[k]);
        } else {
            // publish: per bucket {histogram, lowest caller}, header {#events, overflow, #successes}: self-validating granules
            for (int k = tid; k < nP; k += WG_THREADS) {
                const int ml = min(Lc.mloc[k], Lc.mloc_stay[k]);
                st_gr(C.sx, mygr + 1 + k, mk_granule((unsigned)Lc.hist[k], ml == INT_MAX ? GR_NONE : (unsigned)ml, tag));
            }
            if (tid == 0) {
                const int nevraw = L.scal[C_NEV];
                st_gr(C.sx, mygr, mk_granule((unsigned)min(nevraw, C.evw) | (nevraw > C.evw ? (1u << 13) : 0u), (unsigned)L.scal[C_NSUCC], tag));
                L.scal[C_NEV] = 0; L.scal[C_NCAND] = 0;
            }
            STAMP(1);
            FSTAMP(5); // publish stores
            // round 1: the bucket granules of every workgroup and, on the last wavefront, the headers.  The loads are issued,
            // then (pipelined) phase A of the NEXT subframe runs while they and the other workgroups' stores are in flight,
            // then every granule is checked and, if its tag is still the old one, re-read until it arrives.
            const bool ahead = pipelined && t + 1 < P.stop;
            long long gv[4] = {0, 0, 0, 0}, hv = 0; // the first four sweeps cover 4096 bucket granules (G = 64 with 54 preambles: 3456)
            auto issue_loads = [&]() {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int k = tid + u * WG_THREADS;
                    if (k < G * nP) { const int wg = k / nP, p = k - wg * nP; gv[u] = ld_sc1_64(gr_of(C, parity, wg) + 1 + p); }
                }
                if (tid >= WG_THREADS - 64 && tid - (WG_THREADS - 64) < G) hv = ld_sc1_64(gr_of(C, parity, tid - (WG_THREADS - 64)));
            };
            if (ahead) {
                // subframe t+1: arrivals of its access slot (Beta.c:121-134), then phase A on this workgroup's records as the
                // pass of subframe t left them; the round-1 loads go out from inside it (late: the other workgroups' stores have
                // had time to land, so that few granules have to be read twice)
                const int acNext = ((t + 1) % aT == 0 && activeCheck != nUE) ? P.sched[(t + 1) / aT] : activeCheck;
                compact_phase_a<true>(P, Ln, C, FM, nullptr, nullptr, t + 1, activeCheck, acNext, nullptr, 0u, issue_loads);
            } else {
                issue_loads();
            }
            FSTAMP(6); // phase A ahead + round-1 loads issued
            auto take_bucket = [&](const int k, long long g_, const bool fetched) {
                const int wg = k / nP, p = k - wg * nP;
                if (!fetched || !granule_ok(g_, tag)) g_ = wait_granule(gr_of(C, parity, wg) + 1 + p, tag, &L.scal[C_STATUS]);
                const unsigned h = (unsigned)g_ & 0xFFFFFu, ml = (unsigned)((unsigned long long)g_ >> 32) & 0xFFFFFu;
                if (h) atomicAdd(&L.total[p], (int)h);
                if (ml != GR_NONE) atomicMin(&fcallA[p], (int)ml);
            };
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = tid + u * WG_THREADS;
                if (k < G * nP) take_bucket(k, gv[u], true);
            }
            for (int k = tid + 4 * WG_THREADS; k < G * nP; k += WG_THREADS) take_bucket(k, 0ll, false); // (more than 4096 bucket granules)
            if (tid >= WG_THREADS - 64) {
                const int l = tid - (WG_THREADS - 64);
                int nev = 0, nsuc = 0, ovf = 0;
                if (l < G) {
                    long long g_ = hv;
                    if (!granule_ok(g_, tag)) g_ = wait_granule(gr_of(C, parity, l), tag, &L.scal[C_STATUS]);
                    const unsigned w0 = (unsigned)g_ & 0xFFFFFu;
                    nev = (int)(w0 & 0x1FFFu); ovf = (int)((w0 >> 13) & 1u); nsuc = (int)((unsigned)((unsigned long long)g_ >> 32) & 0xFFFFFu);
                }
                const int x = wave_scan_incl(nev); // (the whole last wavefront is here)
                L.evoff[l] = x - nev;
                nsuc = wave_sum(nsuc);
                ovf = __ballot(ovf != 0) != 0ull;
                if (l == 63) L.scal[C_NTOT] = x;
                if (l == 0) { L.scal[C_NSUCCTOT] = nsuc; L.scal[C_OVF] = ovf; }
            }
            if (ahead) ahead_for = t + 1;
            FSTAMP(7); // round-1 granules taken
            __syncthreads(); // S3: totals, lowest definite callers, event offsets (and the next subframe's phase A)
            FSTAMP(8); // S3
            STAMP(2);
            for (int k = tid; k < nP; k += WG_THREADS) { Lc.hist[k] = 0; Lc.mloc[k] = INT_MAX; Lc.mloc_stay[k] = INT_MAX; Lc.cand_n[k] = 0; } // this parity is used again in two subframes
            if (L.scal[C_STATUS] != PRACH_OK) { status = L.scal[C_STATUS]; time_exit = t; break; }
            N = L.scal[C_NTOT];
            if (L.scal[C_OVF] || N > EVCAPC) { status = PRACH_ERR_INTERNAL; time_exit = t; break; } // engine falls back to trial_kernel
            // round 2: every event granule, classified on arrival
            for (int k = tid; k < N; k += WG_THREADS) {
                int lo = 0, hi = G; // workgroup whose segment holds event k
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.evoff[mid] <= k) lo = mid; else hi = mid; }
                const long long e = wait_granule(gr_of(C, parity, lo) + 1 + nP + (k - L.evoff[lo]), tag, &L.scal[C_STATUS]);
                const int2 ev = make_int2((int)((unsigned)e & 0xFFFFFu), (int)((unsigned)((unsigned long long)e >> 32) & 0xFFFFFu));
                L.gev[k] = ev;
                classify_event(L, fcallA, k, ev);
            }
        }
        FSTAMP(9); // round 2
        if (N > 0) __syncthreads(); // S4: events gathered and classified against the lowest DEFINITE callers (N is uniform)
        FSTAMP(10); // S4
        STAMP(3);

        // ---- resolve (identical on every workgroup of the cluster) ----
        const int nrc = L.scal[C_NRC];
        if (nrc > 0) { // rare: reset cycles that may re-join — decided strictly in index order, then recount
            if (nrc > RCCAP) { status = PRACH_ERR_INTERNAL; time_exit = t; break; }
            if (tid < 64) resolve_reset_candidates(L, fcallA, nrc, nP);
            for (int k = 64 + tid; k < 64 + nP; k += WG_THREADS) { L.nlv[k - 64] = 0; L.fie[k - 64] = 0; }
            __syncthreads();
            for (int k = tid; k < N; k += WG_THREADS) {
                const int2 e = L.gev[k];
                const int type = e.y & 7, p = (e.y >> 4) & 0xff;
                if (type == EVC_LEAVER) { if (e.x < fcallA[p]) atomicAdd(&L.nlv[p], 1); }
                else if ((type == EVC_CALLER || type == EVC_RESETCAND) && e.x == fcallA[p]) L.fie[p] = 1;
            }
            __syncthreads();
        }
        STAMP(4); STAMP(5);
#ifdef PRACH_STAMPS
        if (tid == 0) { statN += (unsigned long long)N; if ((unsigned long long)N > maxN) maxN = N; statRC += (unsigned long long)nrc; }
#endif
        // every call: scan count `check` (Beta.c:321-330), counters (Beta.c:334,349-351 / WithNOMA:650-652)
        const int nrj = L.scal[C_NRJ];
        int my_coll = 0, my_txop = 0;
        for (int k = tid; k < N + nP; k += WG_THREADS) {
            int idx = 0, p = 0, ispre = 0;
            bool caller = false;
            if (k < N) {
                const int2 e = L.gev[k];
                const int type = e.y & 7;
                if (type == EVC_CALLER || type == EVC_RESETCAND) { caller = true; idx = e.x; p = (e.y >> 4) & 0xff; ispre = (e.y >> 3) & 1; }
            } else {
                p = k - N;
                if (fcallA[p] != INT_MAX && !L.fie[p]) { caller = true; idx = fcallA[p]; ispre = 1; } // a STAY pre-member calls first
            }
            if (!caller) continue;
            const bool first = idx == fcallA[p];
            int rj = 0;
            if (nrj > 0) { // Msg3-timeout re-entries that stayed matched since the previous call on this bucket (rare)
                int prev = (!first) ? fcallA[p] : -1;
                for (int j = 0; j < N; j++) {
                    const int2 ej = L.gev[j];
                    const int tj = ej.y & 7;
                    if ((tj == EVC_CALLER || tj == EVC_RESETCAND) && ((ej.y >> 4) & 0xff) == p && ej.x < idx && ej.x > prev) prev = ej.x;
                }
                for (int j = 0; j < N; j++) {
                    const int2 ej = L.gev[j];
                    if ((ej.y & 7) == EVC_RJOIN && ((ej.y >> 4) & 0xff) == p && ej.x < idx && ej.x > prev) rj++;
                }
            }
            const int check = 1 + (first ? L.total[p] - ispre - L.nlv[p] : 0) + rj;
            if (lcallA[p] < idx) atomicMax(&lcallA[p], idx);
            if (check == 1) {
                const int s = atomicAdd(&L.scal[C_NS], 1);
                if (s < SCAPC) L.sidx[s] = idx;
                my_txop += 1;
            } else if (withnoma) { // WithNOMA:650-652
                my_coll += check; my_txop += check;
            } else { // Beta.c:349-351
                my_coll += 1; my_txop += 1;
            }
        }
        my_coll = wave_sum(my_coll); my_txop = wave_sum(my_txop);
        if ((tid & 63) == 0) { if (my_coll) atomicAdd(&L.scal[C_COLL], my_coll); if (my_txop) atomicAdd(&L.scal[C_TXOP], my_txop); }
        FSTAMP(11); // calls
        __syncthreads(); // S5: calls done; singles listed
        FSTAMP(12); // S5
        STAMP(6);
        const int ns = L.scal[C_NS];
        if (ns > SCAPC) { status = PRACH_ERR_INTERNAL; time_exit = t; break; }
        const int Gr = max(0, P.nGrantUL - 1 - grantCheck); // Beta.c:336-347
        if (Gr > 0 && ns > 0 && ns <= 64) {
            // up to one wavefront of singleton callers: every lane ranks its own index against the others through
            // v_readlane — no LDS round trip, no workgroup barrier
            if (tid < 64) {
                const int nsu = __builtin_amdgcn_readfirstlane(ns);
                const int my = tid < nsu ? L.sidx[tid] : INT_MAX;
                int rank = 0;
                for (int s_ = 0; s_ < nsu; s_++) rank += __builtin_amdgcn_readlane(my, s_) < my ? 1 : 0;
                if (tid < nsu && rank < Gr && ((my >> 6) % G) == b) grant(my);
            }
        } else if (Gr > 0 && ns > 0) { // (most subframes of an overloaded 5 ms window have no grant left: nothing to select)
            // the Gr lowest-index singleton callers, in O(ns): counts per index bin (1024 bins over [0,nUE)),
            // block-wide exclusive prefix, whole bins below the crossing bin are granted, the crossing bin is
            // ranked exactly.  The grant itself is ONE atomicOr into the UE's record by the UE's owner.
            const int binshift = P.binshift;
            L.bins[tid] = 0;
            if (tid == 0) L.scal[C_NCROSS] = 0;
            __syncthreads();
            for (int j = tid; j < ns; j += WG_THREADS) atomicAdd(&L.bins[L.sidx[j] >> binshift], 1);
            __syncthreads();
            {
                const int c = L.bins[tid];
                const int x = wave_scan_incl(c);
                if ((tid & 63) == 63) L.wtot[tid >> 6] = x;
                __syncthreads();
                int add = 0;
                for (int k = 0; k < (tid >> 6); k++) add += L.wtot[k];
                L.bins[tid] = x - c + add; // exclusive prefix
            }
            __syncthreads();
            // whole bins below the crossing bin: granted; members of the (single) crossing bin: compacted, then
            // ranked among themselves
            for (int j = tid; j < ns; j += WG_THREADS) {
                const int my = L.sidx[j];
                const int bin = my >> binshift;
                const int before = L.bins[bin];
                if (before >= Gr) continue;
                const int cnt = (bin + 1 < GBINS ? L.bins[bin + 1] : ns) - before;
                if (before + cnt <= Gr) { if (((my >> 6) % G) == b) grant(my); }
                else { const int s_ = atomicAdd(&L.scal[C_NCROSS], 1); if (s_ < RCCAP) L.rclist[s_] = my; }
            }
            __syncthreads();
            const int ncross = L.scal[C_NCROSS];
            if (ncross > RCCAP) { status = PRACH_ERR_INTERNAL; time_exit = t; break; }
            if (tid < ncross) {
                const int my = L.rclist[tid];
                int rank = L.bins[my >> binshift];
                for (int m = 0; m < ncross; m++) rank += L.rclist[m] < my ? 1 : 0;
                if (rank < Gr && ((my >> 6) % G) == b) grant(my);
            }
        }
        grantCheck += ns;
        const int nsucc_tot = L.scal[C_NSUCCTOT];
        FSTAMP(13); // grants
        if (Gr > 0 && ns > 0) __syncthreads(); // S6: the grants are in the records before the next pass reads them
        FSTAMP(14); // S6
        STAMP(7);
#ifdef PRACH_STAMPS
        if (tid == 0) { statNS += (unsigned long long)ns; if ((unsigned long long)ns > maxNS) maxNS = ns; }
#endif
        if (L.scal[C_STATUS] != PRACH_OK) { status = L.scal[C_STATUS]; time_exit = t; break; }
        if (nsucc_tot == nUE) { time_exit = t; break; } // Beta.c:180
    }
    __syncthreads();
    if (status == PRACH_OK && tlast >= 0) cluster_pass<3>(P, L, C, FM, L.fcall + (tlast & 1) * NPC, L.lcall + (tlast & 1) * NPC, tlast + 1, activeCheck, activeCheck, nullptr, 0u, 0ull);
    __syncthreads();

    // end-of-trial sums (Beta.c:185-197) and the logged fields (Beta.c:501-508) of the owned UEs
    const int tend = tlast + 1;
    long long sumT = 0;
    int ptcS = 0, fcS = 0;
    unsigned long long ndS = 0;
    for (int x = tid; x < lgroups * 64; x += WG_THREADS) {
        const int g = b + G * (x >> 6), i = g * 64 + (x & 63);
        if (g >= totgroups || i >= nUE) continue;
        const UeState u = unpack(hot_load_full(C, i, x));
        const int timer = u.act == ACT_IDLE ? -1 : (u.act == ACT_DONE ? u.tb : tend - u.tb);
        const int ptc = P.ptc[i], fc = P.fcnt[i];
        if (u.act == ACT_DONE) { sumT += timer; ptcS += ptc; fcS += fc; }
        ndS += nd_load(P, C, i, x);
        P.timers[i] = u.act == ACT_DONE ? timer : INT_MIN;
        if (P.logs) {
            prach_ue_log o;
            o.idx = i; o.timer = timer; o.active = u.act - 1; o.txTime = u.tx; o.firstTxTime = P.ftt[i];
            o.secondTxTime = P.stt[i]; o.nowBackoff = now_backoff(u.bo, tend); o.preamble = u.pre - 1;
            o.preambleChange = u.pre != 0; o.rarWindow = u.rar; o.maxRarCounter = u.mrc; o.preambleTxCounter = ptc;
            o.msg2Flag = (u.act == ACT_M3 || u.act == ACT_DONE); o.connectionRequest = u.conn == 2 ? 48 : u.conn;
            o.msg4Flag = u.act == ACT_DONE; o.failCount = fc;
            store_log(P.logs, i, o);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sumT += __shfl_down(sumT, d); ptcS += __shfl_down(ptcS, d); fcS += __shfl_down(fcS, d); ndS += __shfl_down(ndS, d);
    }
    if ((tid & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&L.scal[C_SUMT]), (unsigned long long)sumT);
        atomicAdd(reinterpret_cast<unsigned long long *>(&L.scal[C_ND]), ndS);
        atomicAdd(&L.scal[C_PTC], ptcS);
        atomicAdd(&L.scal[C_FC], fcS);
    }
    __syncthreads();
    if (tid == 0) { // DevResult was zeroed by the engine before the launch
        PRACH_G DevResult *o = P.out;
        gadd(&o->sumTimer, *reinterpret_cast<long long *>(&L.scal[C_SUMT]));
        if (GLIBC) { if (b == 0) gadd(&o->draws, base); }
        else gadd(&o->draws, *reinterpret_cast<unsigned long long *>(&L.scal[C_ND]));
        gadd(&o->ptcSum, L.scal[C_PTC]);
        gadd(&o->fcSum, L.scal[C_FC]);
        gadd(&o->nSuccess, L.scal[C_NSUCC]);
        gadd(&o->finalSuccess, L.scal[C_NSUCC]);
        gadd(&o->continueFailed, L.scal[C_CONTF]);
        if (status != PRACH_OK) gmin(&o->status, status);
        if (status == PRACH_ERR_INTERNAL) o->hard_error = 1; // (a capacity was exceeded: that, not a peer's time-out, is what the engine must act on)
        if (b == 0) {
#ifdef PRACH_STAMPS
            for (int k = 0; k < 8; k++) o->stamps6[k] = stamps[k];
            for (int k = 0; k < 24; k++) o->fstamps[k] = fstamps[k];
            o->dbg[0] = statN; o->dbg[1] = maxN; o->dbg[2] = statRC; o->dbg[3] = (statNS << 20) | maxNS;
#endif
            o->time_exit = time_exit;
            o->collisionPreambles = L.scal[C_COLL];
            o->totalPreambleTxop = L.scal[C_TXOP];
            o->activeCheck = activeCheck;
            o->steps = steps;
        }
        if (L.scal[C_VISITS] | L.scal[C_EVENTS]) { // own-traffic accounting of the compacted pass (< 2^31 per workgroup and trial)
            gadd(&o->visits, (unsigned long long)(unsigned)L.scal[C_VISITS]);
            gadd(&o->events, (unsigned long long)(unsigned)L.scal[C_EVENTS]);
        }
    }
}

size_t cluster_kernel_lds_bytes(int nP, bool glibc, int lslots) {
    (void)nP; // the per-bucket tables have a fixed stride
    if (lslots > 0) return (size_t)lds_off::TAIL_L + (size_t)lslots * 20;
    if (glibc) return (size_t)lds_off::TAIL_G + sizeof(int) * 4 * GSCAP; // (no stage: CtxT::STAGED)
    return (size_t)lds_off::TAIL_G + sizeof(int) * NW * CtxT<REC_G16, false>::STG;
}

using cluster_kernel_t = void (*)(const TrialDev *, int, int, int, int);
// rec_mode: REC_G16 / REC_H8 (the reference's rand() stream with one workgroup per trial) / REC_L16 (clusters, Philox: LDS-resident records).
// Philox with one workgroup per trial is prach_batch.hip's; outside its limits (nPreamble > 64, ...) the 16-byte form runs here.
static cluster_kernel_t pick_cluster_kernel(int rng_mode, int rec_mode) {
    if (rng_mode == PRACH_RNG_GLIBC) return rec_mode == REC_H8 ? cluster_kernel<true, REC_H8> : cluster_kernel<true, REC_G16>;
    return rec_mode == REC_L16 ? cluster_kernel<false, REC_L16> : cluster_kernel<false, REC_G16>;
}

hipError_t launch_cluster_kernel(const TrialDev *params, int ntrials, int G, int maxP, int rng_mode, int rec_mode, int lslots, int xpack, hipStream_t stream) {
    if (rec_mode != REC_L16) lslots = 0;
    if (rec_mode == REC_H8 && rng_mode != PRACH_RNG_GLIBC) rec_mode = REC_G16;
    const size_t lds = cluster_kernel_lds_bytes(maxP, rng_mode == PRACH_RNG_GLIBC, lslots);
    const cluster_kernel_t fn = pick_cluster_kernel(rng_mode, rec_mode);
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return rc;
    if (rec_mode == REC_H8 || G <= 1) xpack = 0;
    const int grid = xpack ? ((ntrials + 7) / 8) * 8 * G : ntrials * G;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(WG_THREADS), lds, stream, params, G, lslots, xpack, ntrials);
    return hipGetLastError();
}

// workgroups of this kernel (with its dynamic LDS) the runtime admits per CU: what a cooperative launch would be checked against
int cluster_kernel_blocks_per_cu(int maxP, int rng_mode, int rec_mode, int lslots) {
    if (rec_mode != REC_L16) lslots = 0;
    if (rec_mode == REC_H8 && rng_mode != PRACH_RNG_GLIBC) rec_mode = REC_G16;
    const size_t lds = cluster_kernel_lds_bytes(maxP, rng_mode == PRACH_RNG_GLIBC, lslots);
    const cluster_kernel_t fn = pick_cluster_kernel(rng_mode, rec_mode);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(fn), WG_THREADS, lds) != hipSuccess || nb < 1) return 1;
    return nb;
}

} // namespace prach
