// prach_batch.hip — the BATCHED regime: one workgroup per trial, thousands of trials in flight (BASELINE configs[2] / [4]: the
// --times x nUE sweep).  The state of the in-flight trials streams through HBM every subframe, so what a UE costs is what is READ of it
// and how many instructions are spent on it — in every subframe it merely waits in.
//
// Round 3 walked one 32-bit pass word per live UE per subframe (79 % -> 30 % of a subframe; profiles/r03_config3.md).  Round 4: NO WALK.
// A UE in steady contention is bumped every subframe (Beta.c:346,358) and counts one RAR-window subframe each time (Beta.c:245), so when the
// event body schedules a UE its whole trajectory until its next event is known: matched by preambleCollision scans in [tj, tj + dur), an
// event at tj + dur (prach_ue_body.h: pw_schedule).  The trial keeps two CALENDARS (time-indexed lists in global memory, their fill counts in LDS):
//   * the JOIN calendar: at tj the UE adds itself to the per-bucket histogram and lowest-index tables of the subframes tj .. tj + dur - 1 —
//     a ring of HRING subframes in LDS (dur x 2 LDS atomics, ONCE per contention window instead of two per subframe of the window);
//   * the EVENT calendar: at tj + dur (window expiry, Msg3, a deferred outcome) the UE's index is in the list of the subframe — that list IS the
//     event queue of the subframe: complete when the subframe begins (no pass, no barrier in front of the event body), dense (64 UEs per batch).
// What a subframe costs is therefore its EVENTS (one 32-byte record in, one out, two 4-byte calendar entries in and out), not its live UEs.
//   * An UL grant (the resolver, Beta.c:336-347) takes a UE out of contention early: the UE goes into the next subframe's event list (bit 31),
//     into the workgroup's granted set of this subframe (LDS bitmap + list: its other calendar entries of the next subframe are skipped, a later
//     one is recognised by its generation count), and a granted mid-window UE — necessarily the only member of its bucket — is taken out of the
//     ring's later subframes by the granting thread (its remaining window rides in the low bits of the lowest-index word).
//   * Everything about a UE — the 16-byte hot record of prach_device.h plus draw index, preambleTxCounter, failCount, first / second TxTime, the
//     schedule word — is ONE 32-byte record, read and written only by the event body.
//   * The event body is prach_ue_body.h (shared with every other kernel); the resolver is prach_cluster.hip's for one workgroup.
//   * No per-subframe capacity on the resolver's event list or the leaver candidates (they continue in global memory); a calendar list holds
//     PD->calcap entries (the engine sizes it at nUE / 4, at nUE for small trials) — beyond that the trial is rerun on trial_kernel, reported.
//   * batch_kernel<16, true>: the same in the reference's own rand() stream (what `prach_sim -t 100` issues per sweep point) — the event body
//     runs as a count pass + a block-wide prefix over the 64-UE groups + a select pass, see the kernel's head.
//   * PRACH_FLAG_SECTOR_GRANTS (WithNOMA:626-637): six grant budgets in the grant phase.
// Limits (the engine falls back to prach::cluster_kernel): nPreamble <= 64, maxRarWindow <= 64 (<= 16 in the two-trials-per-CU shape), < 65 000
// subframes, backoff + accessTime + maxRarWindow + 70 <= 256 (the calendar's horizon); the reference-stream form: 131 072 UEs.
// Reference semantics: RandomAccessSimulatorBeta.c:111-197 / RandomAccessWithNOMA.c:267-351; decomposition: DESIGN.md section 3.
#include "prach_device.h"
#include "prach_device_fn.h"
#include "prach_ue_body.h"
#include <limits.h>
#include <type_traits>

namespace prach {

namespace {

#ifdef PRACH_STAMPS
#define BSTAMP(k)                                                                                      \
    do {                                                                                               \
        if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); fstamps[k] += now_ - fprev; fprev = now_; } \
    } while (0)
#else
#define BSTAMP(k) do { } while (0)
#endif

#ifndef PRACH_B_K1
#define PRACH_B_K1 1   // (experiment switch) 0: no separate list / short path for "window closes, retransmit" events
#endif
#ifndef PRACH_B_W8_WAVES
#define PRACH_B_W8_WAVES 6 // (experiment switch) wavefronts per SIMD the 512-thread shape is compiled for: 4 = two workgroups per CU, 6 = three
#endif
#ifndef PRACH_B_XVALU
#define PRACH_B_XVALU 0 // (sensitivity experiment) extra Philox draws per event batch, results unused
#endif
#ifndef PRACH_B_XLDS
#define PRACH_B_XLDS 0  // (sensitivity experiment) extra returning LDS atomics per event batch (on a dummy word)
#endif
#ifndef PRACH_B_XST
#define PRACH_B_XST 0   // (sensitivity experiment) extra scattered 4-byte global stores per event batch (into the UE's own record word 7, rewritten right after)
#endif
#ifndef PRACH_B_PIPE
#define PRACH_B_PIPE 1 // (experiment switch) 0: the compiler places the waits for the next batch's loads
#endif
constexpr int NPB = 64;       // stride of the per-bucket tables (nPreamble <= 64)
constexpr int BSC = 2048;     // singleton callers per subframe
constexpr int BGB = 1024;     // grant selection bins
constexpr int CR = 256;       // calendar slots: the fill counts of the subframes t .. t + 255 (LDS); a trial uses the first calmask + 1 of them
// Workgroup shapes.  NWB wavefronts per workgroup: 16 (1024 threads, one workgroup = one trial per CU) or 8 (512 threads and an LDS
// footprint under 80 KB, so that TWO workgroups = two independent trials share a CU: while one waits at a barrier or for its event
// records, the other one issues).  What is held in LDS per subframe (event list and candidate list continue in global memory; the singleton
// list BSC, the reset-cycle / crossing-bin lists RCCAP and the granted list are capacities: beyond them the engine reruns the trial on trial_kernel):
template <int NWB> struct BCap {
    static constexpr int EV = NWB == 16 ? 4096 : 1536;    // events of a subframe held in LDS (more: global memory)
#ifdef PRACH_QCAP
    static constexpr int CAND = PRACH_QCAP;                 // (test build: nearly every subframe's candidate list continues in global memory)
#else
    static constexpr int CAND = NWB == 16 ? 4096 : 1024;  // early-leaver candidates of a subframe held in LDS (more: global memory)
#endif
    static constexpr int HRING = NWB == 16 ? 64 : 16;     // subframes ahead the histogram / lowest-index ring holds: a window lasts maxRarWindow - 1 < HRING subframes
    static constexpr int GBITS = NWB == 16 ? 32768 : 16384; // granted-UE bitmap (bit = UE index mod GBITS: exact below that many UEs, else a filter in front of the list)
    static constexpr int GL = NWB == 16 ? 1024 : 512;     // granted UEs of a subframe (list)
};

constexpr int EVB_CALLER = UEV_CALLER, EVB_RESETCAND = UEV_RESETCAND, EVB_RJOIN = UEV_RJOIN, EVB_LEAVER = 4;

// (the event / candidate / granted counts exist twice, by subframe parity: a wavefront that is already in the next subframe's body appends to the other one)
enum { B_NSUCC = 0, B_COLL, B_TXOP, B_CONTF, B_NS, B_NRC, B_NRJ, B_OVF /* a calendar list / the granted list is full: the trial leaves */, B_NEV = 8 /* [2] */, B_NCAND = 10 /* [2] */,
       B_NGL = 12 /* [2] granted UEs */, B_PTC = 14, B_FC, B_SUMT = 16, B_ND = 18, B_JOINS = 20, B_EVENTS = 21, B_NCROSS = 22, B_SGC = 24 /* [24, 30): sectorGrants[6], WithNOMA:260 (PRACH_FLAG_SECTOR_GRANTS) */ };

// ---- LDS layout: byte offsets, all compile-time ------------------------------------------------------------------------------------
template <int NWB> struct BL {
    using C = BCap<NWB>;
    static constexpr int GEV = 0;                          // int2 [EV] events of this subframe
    static constexpr int SIDX = GEV + 8 * C::EV;           // int [BSC] singleton callers: index | bucket << 20
    static constexpr int RCL = SIDX + 4 * BSC;             // int [RCCAP]
    static constexpr int SCAL = RCL + 4 * RCCAP;           // int [64]
    static constexpr int BINS = SCAL + 4 * 64;             // int [BGB]
    static constexpr int WTOT = BINS + 4 * BGB;            // int [16]
    static constexpr int TOTAL = WTOT + 4 * 16;            // int [NPB]
    static constexpr int FCALL = TOTAL + 4 * NPB;          // int [2][NPB] by subframe parity
    static constexpr int LCALL = FCALL + 8 * NPB;          // int [2][NPB]
    static constexpr int NLV = LCALL + 8 * NPB;            // int [NPB]
    static constexpr int FIE = NLV + 4 * NPB;              // int [NPB]
    static constexpr int FMINP = FIE + 4 * NPB;            // int [NPB] the lowest matched UE of every bucket as the ring held it: index << 6 | subframes of its window still ahead
    static constexpr int HR = FMINP + 4 * NPB;             // int [HRING][NPB] histogram ring: matched UEs per bucket in the subframes t .. t + HRING - 1 (slot = subframe mod HRING)
    static constexpr int MR = HR + 4 * C::HRING * NPB;     // int [HRING][NPB] lowest matched index << 6 | remaining window, same ring
    static constexpr int ECNT = MR + 4 * C::HRING * NPB;   // int [2][CR] entries in the event calendar's list of the subframes t .. (slot = subframe & calmask): [0] "window closes,
                                                           // retransmit" events, filled from the list's front; [1] every other event, filled from its back
    static constexpr int JCNT = ECNT + 8 * CR;             // int [CR] ... in the join calendar's
    static constexpr int GBM = JCNT + 4 * CR;              // unsigned [2][GBITS / 32] granted-UE bitmap, by subframe parity
    static constexpr int GLIST = GBM + 2 * C::GBITS / 8;   // int [2][GL] granted UEs, by subframe parity
    static constexpr int CANDL = GLIST + 2 * 4 * C::GL;    // int [CAND] early-leaver candidates: index | old bucket << 20
    static constexpr int END = CANDL + 4 * C::CAND;
    static_assert(SIDX % 16 == 0 && HR % 16 == 0, "alignment");
};
static_assert(BL<8>::END <= 160 * 1024 / 3, "three 512-thread workgroups per CU");
// The reference's own rand() stream (GLIBC instantiation, 1024 threads): per 64-UE group of the trial, the lanes that make at least one / two rand()
// calls in this subframe (two 64-bit masks) and the group's exclusive prefix of calls in index order.  BGG groups = 131 072 UEs at most.
constexpr int BGG = 2048;
struct BLG { static constexpr int GM = (BL<16>::END + 15) / 16 * 16, GPRE = GM + 16 * BGG, END = GPRE + 4 * BGG; };
static_assert(BLG::END <= 160 * 1024 && BLG::GM % 16 == 0, "LDS");

#define BI(off) (reinterpret_cast<int *>(smem + (off)))
#define BU(off) (reinterpret_cast<unsigned *>(smem + (off)))
#define BI2(off) (reinterpret_cast<int2 *>(smem + (off)))

// the 32-byte event record: A = the hot record of prach_device.h {txTime, timer base, nowBackoff, packed}; B = {Philox draw index (24 bits) |
// generation count of the UE's schedule << 24, preambleTxCounter | failCount << 16, secondTxTime | firstTxTime << 16, the schedule word}
struct BRec { int4 a, b; };
__device__ __forceinline__ BRec brec_load(const PRACH_G v4i_t *p) {
    const v4i_t a = p[0], b = p[1]; // (plain loads and stores: non-temporal ones cost 30 % on config 3 — a record's line is read again soon)
    BRec r;
    r.a = make_int4(a.x, a.y, a.z, a.w); r.b = make_int4(b.x, b.y, b.z, b.w);
    return r;
}
__device__ __forceinline__ void brec_store(PRACH_G v4i_t *p, const int4 a, const int4 b) {
    v4i_t va, vb;
    va.x = a.x; va.y = a.y; va.z = a.z; va.w = a.w; vb.x = b.x; vb.y = b.y; vb.z = b.z; vb.w = b.w;
    p[0] = va; p[1] = vb;
}
__device__ __forceinline__ ColdRegs cold_unpack(const int4 b) {
    ColdRegs c;
    c.ptc = b.y & 0xffff; c.fcnt = (int)((unsigned)b.y >> 16); c.stt = b.z & 0xffff; c.ftt = (int)((unsigned)b.z >> 16);
    return c;
}
__device__ __forceinline__ int4 cold_pack(const unsigned nd, const unsigned gen, const ColdRegs &c, const unsigned word) {
    return make_int4((int)((nd & 0xFFFFFFu) | (gen << 24)), (c.ptc & 0xffff) | (c.fcnt << 16), (c.stt & 0xffff) | (c.ftt << 16), (int)word);
}
// calendar entries.  Event calendar: UE index [19:0] | generation [27:20] | bit 31: an UL grant of the previous subframe (always valid).
// Join calendar: UE index [19:0] | bucket [25:20] | window length [31:26].
constexpr int CAL_GRANT = (int)0x80000000u;

} // namespace

// ------------------------------------------------------------------------------------------------------------------------------------
// GLIBC: the draws of a subframe come from the reference's own rand() stream (window PD->stream, generated on the device before the launch) at
// the positions the reference's index-ordered UE loop reaches: the event body runs twice — a COUNT pass (catch-up, activation and ue_plan only: how
// many calls each event UE makes follows from its pre-step state) that marks the calling lanes of every group, a block-wide prefix over
// the groups in index order, then the full body as the SELECT pass with d1, d2 = stream[base + prefix[group] + calls of the group's lower lanes].
template <int NWB, bool GLIBC = false>
__global__ __launch_bounds__(NWB * 64, NWB == 8 ? PRACH_B_W8_WAVES : 4) void batch_kernel(const TrialDev *__restrict__ params) {
    static_assert(!GLIBC || NWB == 16, "the reference-stream form exists in the 1024-thread shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using bl = BL<NWB>;
    constexpr int TB = NWB * 64, BEV = BCap<NWB>::EV, CCAP = BCap<NWB>::CAND, HRING = BCap<NWB>::HRING, GBITS = BCap<NWB>::GBITS, GLCAP = BCap<NWB>::GL;
    const TrialDev *const PD = params + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nUE = PD->nUE, nP = PD->nP, aT = PD->aT, stop = PD->stop, nGrantUL = PD->nGrantUL, binshift = PD->binshift;
    const int variant = PD->variant;
    const bool sectors = (PD->flags & PRACH_FLAG_SECTOR_GRANTS) != 0;
    const unsigned seed_lo = PD->seed_lo, seed_hi = PD->seed_hi;
    UeK K;
    K.maxRar = PD->maxRarWindow; K.maxMsg2 = PD->maxMsg2; K.aT = aT; K.withnoma = variant == PRACH_VARIANT_WITHNOMA_C;
    K.fmP = make_fastmod(nP); K.fmB = make_fastmod(PD->backoff); K.fmA = make_fastmod(aT); K.fm5 = make_fastmod(5);
    const bool withnoma = K.withnoma;
    PRACH_G v4i_t *const rec32 = (PRACH_G v4i_t *)PD->rec32;     // [nUE][2]
    PRACH_G int *const ecal = (PRACH_G int *)PD->ecal;           // [calmask + 1][calcap] event calendar
    PRACH_G int *const jcal = (PRACH_G int *)PD->jcal;           // [calmask + 1][calcap] join calendar
    const int calcap = PD->calcap;
    const unsigned calmask = (unsigned)PD->calmask;
    PRACH_G int *const candg = (PRACH_G int *)PD->qov;           // [nUE] early-leaver candidates of a subframe beyond their LDS part
    const PRACH_G int *const sched = (const PRACH_G int *)PD->sched;
    int *const scal = BI(bl::SCAL);
    int2 *const gev = BI2(bl::GEV);
    // the event list of a subframe: BEV entries in LDS, the rest (event storms of extreme parameter sets) in global memory — no capacity
    PRACH_G v2i_t *const evov = (PRACH_G v2i_t *)PD->evov;
    auto ev_get = [&](const int k) -> int2 { if (k < BEV) return gev[k]; const v2i_t v = evov[k - BEV]; return make_int2(v.x, v.y); };
    auto ev_set = [&](const int k, const int a, const int b) { if (k < BEV) gev[k] = make_int2(a, b); else store_i2(&evov[k - BEV], a, b); };
    auto ev_kill = [&](const int k) { if (k < BEV) gev[k].y = 0; else evov[k - BEV].y = 0; };
    int *const ecnt = BI(bl::ECNT), *const jcnt = BI(bl::JCNT);
    int *const hr = BI(bl::HR), *const mr = BI(bl::MR);
    int *const candl = BI(bl::CANDL);
    unsigned *const gm = BU(BLG::GM);  // (GLIBC only) [BGG][4]: lanes with >= 1 call (two words), lanes with 2 calls (two words)
    int *const gpre = BI(BLG::GPRE);   // (GLIBC only) [BGG]
    const PRACH_G int *const stream = (const PRACH_G int *)PD->stream;
    const unsigned long long stream_len = PD->stream_len;
    unsigned long long base = 0; // GLIBC: rand() calls consumed so far (relative to the stream window)

    const int totgroups = (nUE + 63) >> 6;
    // calloc + initialUE (Beta.c:78-83)
    for (int i = tid; i < nUE; i += TB) brec_store(rec32 + 2 * (size_t)i, make_int4(-1, 0, 0, 0), make_int4(0, 0, 0, (int)PW_IDLE));
    for (int k = tid; k < HRING * NPB; k += TB) { hr[k] = 0; mr[k] = INT_MAX; }
    for (int k = tid; k < CR; k += TB) { ecnt[k] = 0; ecnt[CR + k] = 0; jcnt[k] = 0; }
    for (int k = tid; k < 2 * GBITS / 32; k += TB) BU(bl::GBM)[k] = 0u;
    if (tid < NPB) {
        BI(bl::TOTAL)[tid] = 0; BI(bl::NLV)[tid] = 0; BI(bl::FIE)[tid] = 0; BI(bl::FMINP)[tid] = INT_MAX;
        BI(bl::FCALL)[tid] = INT_MAX; BI(bl::FCALL)[NPB + tid] = INT_MAX; BI(bl::LCALL)[tid] = -1; BI(bl::LCALL)[NPB + tid] = -1;
    }
    if (tid < 64) scal[tid] = 0;
    if (GLIBC) for (int k = tid; k < 4 * BGG; k += TB) gm[k] = 0u;
    __syncthreads();

    int activeCheck = 0, grantCheck = 0, tlast = -1, time_exit = stop;
    int acNext = sched[0]; // the arrival table's entry of the NEXT access slot: loaded a slot ahead, so that no subframe waits for it (Beta.c:121-134)
    int why = 0;   // which per-subframe capacity ended the trial (reported)
    unsigned long long steps = 0;
    int status = (nP > NPB || K.maxRar > HRING || stop > 65000 || (GLIBC && totgroups > BGG) || nUE >= (1 << 20) || calmask >= (unsigned)CR ||
                  PD->backoff + max(aT, 5) + K.maxRar + 70 > (int)calmask + 1) ? PRACH_ERR_UNSUPPORTED : PRACH_OK;
#ifdef PRACH_STAMPS
    unsigned long long fstamps[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fprev = __builtin_readcyclecounter();
#endif
    // was UE i given an UL grant in the subframe whose granted set sits at parity `par`?  (bitmap first; exact list behind it when the trial has more UEs than bits)
    auto granted_in = [&](const int par, const int i) -> bool {
        const unsigned *const bm = BU(bl::GBM) + par * (GBITS / 32);
        if (!((bm[(i & (GBITS - 1)) >> 5] >> (i & 31)) & 1u)) return false;
        if (nUE <= GBITS) return true;
        const int n = min(scal[B_NGL + par], GLCAP);
        const int *const gl = BI(bl::GLIST) + par * GLCAP;
        bool hit = false;
        for (int k = 0; k < n; k++) hit = hit || gl[k] == i;
        return hit;
    };

    for (int t = 0; t < stop && status == PRACH_OK; t++) {
        steps++;
        tlast = t;
        if (t % 5 == 0) { // Beta.c:112 (hard-coded 5); WithNOMA:268-274
            grantCheck = 0;
            if (sectors && tid < 6) scal[B_SGC + tid] = 0; // (barriers follow before the grant phase reads them)
        }
        const int prevAC = activeCheck;
        if (t % aT == 0) { // Beta.c:121-134
            if (activeCheck != nUE) activeCheck = acNext;
            acNext = sched[t / aT + 1]; // (the table has maxTime / accessTime + 2 entries: prach_engine.hip)
        }
        const int parity = t & 1;
        int *const fcallA = BI(bl::FCALL) + parity * NPB, *const lcallA = BI(bl::LCALL) + parity * NPB;
        int *const fcallB = BI(bl::FCALL) + (parity ^ 1) * NPB, *const lcallB = BI(bl::LCALL) + (parity ^ 1) * NPB;
        const int slot = (int)((unsigned)t & calmask);                       // this subframe's calendar lists
        int *const histx = hr + (t & (HRING - 1)) * NPB, *const mlocx = mr + (t & (HRING - 1)) * NPB; // ... and its histogram / lowest matched index << 6 | window left
        const int n1 = ecnt[slot], ne = ecnt[CR + slot], nj = jcnt[slot];     // complete: every entry was made in an earlier subframe (the grants' behind S6)
        const int qn = ne + (activeCheck - prevAC);                           // this access slot's arrivals (Beta.c:136-146) are events too: virtual entries behind the list's
        PRACH_G int *const elist = ecal + (size_t)slot * (size_t)calcap;      // retransmission events [0, n1) from the front, the other ne from the back
        if (n1 + ne > calcap) scal[B_OVF] = 1;                                // (the two ends have met: the trial leaves behind S2 and is rerun with longer lists)
        BSTAMP(0); // loop head

        // ================= joins: UEs whose contention window opens in this subframe enter the ring's subframes t .. t + dur - 1 =================
        for (int q0 = (NWB - 1 - w) * 64; q0 < nj; q0 += NWB * 64) { // (from the last wavefront down: the first ones have the most event batches)
            const int q = q0 + lane;
            int je = 0;
            if (q < nj) je = jcal[(size_t)slot * (size_t)calcap + (size_t)q];
            const int i = je & 0xFFFFF, p = (je >> 20) & 63;
            int dur = (int)((unsigned)je >> 26);
            if (q >= nj || granted_in(parity ^ 1, i)) dur = 0; // (granted in the subframe before: out of contention, Beta.c:338-343)
            const int dmax = wave_max(dur);
            for (int k = 0; k < dmax; k++) {
                if (k < dur) {
                    const int rs = ((t + k) & (HRING - 1)) * NPB + p;
                    atomicAdd(&hr[rs], 1);
                    atomicMin(&mr[rs], (i << 6) | (dur - 1 - k));
                }
            }
        }
        BSTAMP(1); // joins
        BSTAMP(2);

        // ================= the event body: this subframe's event list, 64 UEs at a time =================
        {
            int c_succ = 0, c_contf = 0;
            const int tmod = t % aT;
            const CallTables tab{fcallB, lcallB};
            // The loop over a list's batches is written for gfx950's ONE in-order memory counter (loads and stores complete in issue order as far as
            // s_waitcnt vmcnt can tell): at the top of a batch the records of the NEXT batch and the list entries of the one after are requested; they are
            // waited for once, right before this batch's stores are issued (pipe_sync: by then they have had the whole batch to arrive), and the stores then
            // have the next batch's compute phase to complete before anything waits again.  (Waiting at the top of the next batch instead — where the
            // compiler puts it — waits for the scattered stores as well: measured, that was most of the event body's time under load.)
            int pe_n2 = 0;      // list entry of this lane two batches ahead
            BRec pR_n;          // record of this lane's UE in the next batch
            pR_n.a = pR_n.b = make_int4(0, 0, 0, 0);
            auto pipe_sync = [&]() __attribute__((always_inline)) {
                if (PRACH_B_PIPE) asm volatile("" :: "v"(pR_n.a.x), "v"(pR_n.a.y), "v"(pR_n.a.z), "v"(pR_n.a.w), "v"(pR_n.b.x), "v"(pR_n.b.y), "v"(pR_n.b.z), "v"(pR_n.b.w), "v"(pe_n2));
            };
            auto entry_other = [&](const int q) -> int { return q < qn ? (q < ne ? elist[calcap - 1 - q] : prevAC + (q - ne)) : 0; }; // (arrivals: virtual entries)
            auto entry_retx = [&](const int q) -> int { return q < n1 ? elist[q] : 0; };
            auto rec_of = [&](const int e) -> BRec { return brec_load(rec32 + 2 * (size_t)(e & 0xFFFFF)); }; // (record 0 is always mapped: an idle lane loads it and ignores it)
            // what follows selectPreamble / requestResourceAllocation for every event UE: bucket bookkeeping, special events for the resolver, the UE's schedule
            auto finish = [&](const bool v, const int i, UeState &u, ColdRegs &cold, const unsigned nd, const unsigned gen, const UeOut &o) __attribute__((always_inline)) {
                BSTAMP(13); // (fine stamps of the diagnostic build: select logic)
                if (PRACH_B_XVALU) { // (sensitivity experiment: never changes a result)
                    int x_ = 0;
#pragma unroll
                    for (int k_ = 0; k_ < PRACH_B_XVALU; k_++) x_ ^= philox_draw31(seed_lo, seed_hi, (unsigned)i, nd + 77u + (unsigned)k_, (unsigned)nUE, (unsigned)variant);
                    if (x_ == 0x7fffffff && nd == 0xfffffffu) scal[B_OVF] = 3;
                }
                if (PRACH_B_XLDS) {
                    int y_ = 0;
#pragma unroll
                    for (int k_ = 0; k_ < PRACH_B_XLDS; k_++) y_ += atomicAdd(&scal[40 + (k_ & 7)], 1);
                    if (y_ == 0x7fffffff) scal[B_OVF] = 3;
                }
                if (PRACH_B_XST) {
#pragma unroll
                    for (int k_ = 0; k_ < PRACH_B_XST; k_++) if (v) reinterpret_cast<PRACH_G int *>(rec32)[8 * (size_t)i + 7] = k_;
                }
                // ---- bucket bookkeeping (this subframe's ring slot; an event UE has nothing in the later ones: window left = 0) ----
                if (o.member_pre) atomicAdd(&histx[o.oldp], 1);
                if (u.pend == PEND_STAY) atomicMin(&mlocx[o.oldp], i << 6);
                if (o.evtype == UEV_CALLER) atomicMin(&mlocx[o.evp], i << 6);
                {
                    const unsigned long long em = __ballot(o.evtype != UEV_NONE), cm = __ballot(o.eclass);
                    if (em | cm) {
                        int b_ev = 0, b_cd = 0;
                        if (lane == 0) {
                            if (em) b_ev = atomicAdd(&scal[B_NEV + parity], __popcll(em));
                            if (cm) b_cd = atomicAdd(&scal[B_NCAND + parity], __popcll(cm));
                        }
                        b_ev = __builtin_amdgcn_readfirstlane(b_ev); b_cd = __builtin_amdgcn_readfirstlane(b_cd);
                        if (o.evtype != UEV_NONE) {
                            const int es = b_ev + __popcll(em & lanemask_lt(lane));
                            ev_set(es, i, ue_event_info(o));
                        }
                        if (o.eclass) {
                            const int cs = b_cd + __popcll(cm & lanemask_lt(lane)), cv = i | (o.oldp << 20);
                            if (cs < CCAP) candl[cs] = cv; else candg[cs - CCAP] = cv;
                        }
                    }
                }
                BSTAMP(14); // bucket bookkeeping, special events
                // the UE's schedule from here (prach_ue_body.h): matched in [tj, tj + dur), its next event at tj + dur — into the calendars
                bool k1 = false;
                unsigned word = PW_IDLE;
                if (v) word = pw_schedule(u, t, K.maxRar, K.maxMsg2, k1);
                if (GLIBC || !PRACH_B_K1) k1 = false; // (the reference-stream form runs every event through the count / select passes)
                const unsigned gen1 = (gen + 1u) & 0xFFu;
                const unsigned tjn = word & 0xFFFFu, durn = (word >> 16) & 0x3Fu;
                const bool sched_ = v && tjn != 0xFFFFu; // (0xFFFF: finished for good, or a txTime that never comes — Beta.c:167)
                // list positions: one returning LDS atomic per lane and list, both issued before either is waited for (the UEs of a batch go to a handful of
                // subframes — txTime is aligned to the access slots — so the lanes meet on a few words: ~64 LDS cycles each, but ONE round trip; a wavefront-
                // aggregated form, one atomic per distinct subframe, measured slower: five dependent round trips)
                const int js = (int)(tjn & calmask), es_ = (int)((tjn + durn) & calmask);
                int jp = 0, ep = 0;
                if (sched_ && durn > 0u) jp = atomicAdd(&jcnt[js], 1);
                if (sched_) ep = atomicAdd(&ecnt[k1 ? es_ : CR + es_], 1);
                BSTAMP(15); // schedule, list positions
                pipe_sync(); // the next batch's records and the entries behind them have arrived: from here on only stores are issued
                BSTAMP(16); // wait for the next batch
                if (v) brec_store(rec32 + 2 * (size_t)i, pack(u), cold_pack(nd, gen1, cold, word));
                if (sched_) {
                    if (durn > 0u) {
                        if (jp < calcap) jcal[(size_t)js * (size_t)calcap + (size_t)jp] = (int)((unsigned)i | (((word >> 24) & 0x3Fu) << 20) | (durn << 26));
                        else scal[B_OVF] = 1;
                    }
                    if (ep < calcap) ecal[(size_t)es_ * (size_t)calcap + (size_t)(k1 ? ep : calcap - 1 - ep)] = (int)((unsigned)i | (gen1 << 20));
                    else scal[B_OVF] = 1;
                }
                BSTAMP(17); // stores
            };
            // MODE 0: Philox, everything in one pass.  MODE 1 (GLIBC): the count pass — nothing is stored but the calling lanes.  MODE 2 (GLIBC): the full
            // body with the draws at their stream positions (sbase: position of the subframe's first call of the UE loop).
            auto body_pass = [&](auto MODE_, const unsigned long long sbase) __attribute__((always_inline)) {
                constexpr int MODE = decltype(MODE_)::value;
                constexpr int ST = NWB * 64;
                int e_c = entry_other(w * 64 + lane), e_n = entry_other(w * 64 + ST + lane);
                BRec R_c = rec_of(e_c);
                asm volatile("" :: "v"(R_c.a.x), "v"(R_c.a.y), "v"(R_c.a.z), "v"(R_c.a.w), "v"(R_c.b.x), "v"(R_c.b.y), "v"(R_c.b.z), "v"(R_c.b.w), "v"(e_n)); // (the first batch is waited for HERE, not at every batch's top)
                for (int q0 = w * 64; q0 < qn; q0 += ST) {
                    if (q0 + ST < qn) { pR_n = rec_of(e_n); pe_n2 = entry_other(q0 + 2 * ST + lane); } // in flight while this batch is worked on
                    const int e = e_c;
                    const int i = e & 0xFFFFF;
                    const bool granted = e < 0;
                    const bool arrival = q0 + lane >= ne;
                    BRec R = R_c;
                    // (the next batch moves into place at the END of this one, behind pipe_sync: a register move of a value still in flight would wait for it here)
                    auto rotate = [&]() __attribute__((always_inline)) { e_c = e_n; e_n = pe_n2; R_c = pR_n; };
                    const unsigned gen = (unsigned)R.b.x >> 24;
                    // An entry counts if it is the UE's current one: a grant entry always; any other one not when the UE was granted in the subframe before (its
                    // grant entry does the work) and not when the UE has been rescheduled since the entry was made (a grant took it out of its window early)
                    bool v = q0 + lane < qn;
                    if (v && !granted && !arrival) v = (unsigned)((e >> 20) & 0xFF) == gen && !granted_in(parity ^ 1, i);
                    if (!v) { R.a = make_int4(-1, 0, 0, 0); R.b = make_int4(0, 0, 0, 0); }
                    UeState u = unpack(R.a);
                    ColdRegs cold = cold_unpack(R.b);
                    unsigned nd = (unsigned)R.b.x & 0xFFFFFFu;
                    if (v) pw_catch_up(u, (unsigned)R.b.w, granted, i, t, K.fmA, tab);
                    if (v && arrival && u.act == ACT_IDLE) { // arrival (Beta.c:136-146; activateUEs WithNOMA:383-394 also draws twice)
                        ue_activate(u, i, t, cold);
                        if (withnoma) nd = 2;
                        // the reference's stream has no per-UE draw index: the record's word keeps the UE's sector instead, fixed by the first of its two
                        // activation calls (WithNOMA:393-410), which sit at the head of this subframe's calls in index order (read in the select pass: the window has been checked)
                        if (GLIBC && sectors && MODE == 2) nd = (unsigned)sector_of_draw(stream[base + 2ull * (unsigned long long)(i - prevAC)]);
                    }
                    const UePlan pl = ue_plan(u, t, K.maxRar, K.maxMsg2);
                    const int g_ = i >> 6, ln = i & 63;
                    if (MODE == 1) {
                        if (v && pl.need >= 1) {
                            atomicOr(&gm[4 * g_ + (ln >> 5)], 1u << (ln & 31));
                            if (pl.need == 2) atomicOr(&gm[4 * g_ + 2 + (ln >> 5)], 1u << (ln & 31));
                        }
                        rotate();
                        continue;
                    }
                    int d1 = 0, d2 = 0;
                    if (MODE == 2) {
                        if (v && pl.need > 0) { // the UE's position inside its group from the two lane masks
                            const unsigned lo_ = ln < 32 ? (1u << ln) - 1u : 0xffffffffu, hi_ = ln < 32 ? 0u : (1u << (ln - 32)) - 1u;
                            const int before = __popc(gm[4 * g_] & lo_) + __popc(gm[4 * g_ + 1] & hi_) + __popc(gm[4 * g_ + 2] & lo_) + __popc(gm[4 * g_ + 3] & hi_);
                            const unsigned long long o_ = sbase + (unsigned long long)gpre[g_] + (unsigned long long)before;
                            d1 = stream[o_];
                            if (pl.need > 1) d2 = stream[o_ + 1];
                        }
                    } else if (__any(pl.need > 0)) {
                        d1 = philox_draw31(seed_lo, seed_hi, (unsigned)i, nd, (unsigned)nUE, (unsigned)variant);
                        if (__any(pl.need > 1)) d2 = philox_draw31(seed_lo, seed_hi, (unsigned)i, nd + 1u, (unsigned)nUE, (unsigned)variant);
                        nd += (unsigned)pl.need;
                    }
                    const UeOut o = ue_select(u, pl, d1, d2, i, t, tmod, K, cold, c_succ, c_contf);
                    finish(v, i, u, cold, nd, gen, o);
                    rotate();
                }
            };
            // "the RAR window closes, retransmit" (prach_ue_body.h ue_window_retx): the events listed at the front of the subframe's list, one short path for all 64 lanes
            if (!GLIBC) {
                constexpr int ST = NWB * 64;
                const int qs = (NWB - 1 - w) * 64; // (from the last wavefront down: the first ones have the most batches of the other list)
                int e_c = entry_retx(qs + lane), e_n = entry_retx(qs + ST + lane);
                BRec R_c = rec_of(e_c);
                asm volatile("" :: "v"(R_c.a.x), "v"(R_c.a.y), "v"(R_c.a.z), "v"(R_c.a.w), "v"(R_c.b.x), "v"(R_c.b.y), "v"(R_c.b.z), "v"(R_c.b.w), "v"(e_n));
                for (int q0 = qs; q0 < n1; q0 += ST) {
                    if (q0 + ST < n1) { pR_n = rec_of(e_n); pe_n2 = entry_retx(q0 + 2 * ST + lane); }
                    const int e = e_c;
                    const int i = e & 0xFFFFF;
                    BRec R = R_c;
                    const unsigned gen = (unsigned)R.b.x >> 24;
                    const bool v = q0 + lane < n1 && (unsigned)((e >> 20) & 0xFF) == gen && !granted_in(parity ^ 1, i); // (as in body_pass)
                    if (!v) { R.a = make_int4(0, 0, 0, ACT_M1 | (1 << PK_PRE_SHIFT)); R.b = make_int4(0, 0, 0, (int)pw_make(t, 0, 0)); }
                    UeState u = unpack(R.a);
                    ColdRegs cold = cold_unpack(R.b);
                    unsigned nd = (unsigned)R.b.x & 0xFFFFFFu;
                    const int d1 = philox_draw31(seed_lo, seed_hi, (unsigned)i, nd, (unsigned)nUE, (unsigned)variant);
                    nd += 1u;
                    bool ok;
                    UeOut o = ue_window_retx(u, (unsigned)R.b.w, d1, i, t, K, cold, ok);
                    if (v && !ok) scal[B_OVF] = 2; // (a record that is not in the state its list promises: reported, the trial is rerun on trial_kernel)
                    if (!v) { o.evtype = UEV_NONE; o.member_pre = false; o.eclass = false; u.pend = PEND_NONE; }
                    finish(v, i, u, cold, nd, gen, o);
                    e_c = e_n; e_n = pe_n2; R_c = pR_n; // (behind pipe_sync: the next batch moves into place)
                }
            }
            if (!GLIBC) body_pass(std::integral_constant<int, 0>{}, 0ull);
            else {
                // activateUEs' two rand() calls per arrival (WithNOMA:393-394) come first in the subframe, in index order; nothing reads them here
                const unsigned long long actdraws = withnoma ? 2ull * (unsigned long long)(activeCheck - prevAC) : 0ull;
                body_pass(std::integral_constant<int, 1>{}, 0ull);
                __syncthreads(); // the calling lanes of every group are marked
                {
                    constexpr int PERG = BGG / TB; // consecutive groups per thread
                    int vv[PERG], sum = 0;
#pragma unroll
                    for (int u_ = 0; u_ < PERG; u_++) {
                        const uint4 m = *reinterpret_cast<const uint4 *>(&gm[4 * (tid * PERG + u_)]);
                        vv[u_] = __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w);
                        sum += vv[u_];
                    }
                    const int x = wave_scan_incl(sum);
                    if (lane == 63) BI(bl::WTOT)[w] = x;
                    __syncthreads();
                    int run = x - sum;
                    for (int k = 0; k < w; k++) run += BI(bl::WTOT)[k];
#pragma unroll
                    for (int u_ = 0; u_ < PERG; u_++) { gpre[tid * PERG + u_] = run; run += vv[u_]; } // exclusive prefix in index order
                    if (tid == TB - 1) scal[B_NCROSS] = run; // (free here: only the grant selection uses it)
                }
                __syncthreads();
                const unsigned long long tot = actdraws + (unsigned long long)scal[B_NCROSS];
                if (base + tot > stream_len) { status = PRACH_ERR_STREAM; time_exit = t; break; } // (the engine retries with a larger window)
                body_pass(std::integral_constant<int, 2>{}, base + actdraws);
                base += tot;
            }
            if (__any((c_succ | c_contf) != 0)) {
                c_succ = wave_sum(c_succ); c_contf = wave_sum(c_contf);
                if (lane == 0) {
                    if (c_succ) atomicAdd(&scal[B_NSUCC], c_succ);
                    if (c_contf) atomicAdd(&scal[B_CONTF], c_contf);
                }
            }
        }
        BSTAMP(3); // event body
        __syncthreads(); // S1: histogram / lowest callers / candidate list are complete; the caller tables of t - 1 and the granted set of t - 1 are free
        BSTAMP(4);

        // early leavers below the bucket's lowest caller are the only ones a rank can need
        {
            const int ncand = scal[B_NCAND + parity];
            for (int k0 = w * 64; k0 < ncand; k0 += TB) {
                const int k = k0 + lane;
                const int c = k < ncand ? (k < CCAP ? candl[k] : candg[k - CCAP]) : 0;
                const int ci = c & 0xFFFFF, cp = (c >> 20) & 63;
                const unsigned long long lm = __ballot(k < ncand && ci < (mlocx[cp] >> 6));
                if (lm) {
                    int b_ev = 0;
                    if (lane == 0) b_ev = atomicAdd(&scal[B_NEV + parity], __popcll(lm));
                    b_ev = __builtin_amdgcn_readfirstlane(b_ev);
                    if ((lm >> lane) & 1ull) ev_set(b_ev + __popcll(lm & lanemask_lt(lane)), ci, EVB_LEAVER | (cp << 4));
                }
            }
            if (GLIBC) // every draw of this subframe has been read: the marks of the groups the event UEs are in can go
                for (int q = tid; q < qn; q += TB) {
                    const int i = q < ne ? (elist[calcap - 1 - q] & 0xFFFFF) : prevAC + (q - ne);
                    *reinterpret_cast<uint4 *>(&gm[4 * (i >> 6)]) = make_uint4(0u, 0u, 0u, 0u);
                }
            { // the granted set of subframe t - 1 has been used by every join and every event of this subframe: empty it (bit by bit, from its list)
                const int par = parity ^ 1;
                const int n = min(scal[B_NGL + par], GLCAP);
                unsigned *const bm = BU(bl::GBM) + par * (GBITS / 32);
                const int *const gl = BI(bl::GLIST) + par * GLCAP;
                for (int k = tid; k < n; k += TB) bm[(gl[k] & (GBITS - 1)) >> 5] = 0u;
            }
            if (tid < NPB) { // this workgroup's histogram / lowest callers ARE the totals (only read here: the filter above reads them too)
                lcallB[tid] = -1; BI(bl::NLV)[tid] = 0; BI(bl::FIE)[tid] = 0;
                const int m_ = mlocx[tid];
                BI(bl::TOTAL)[tid] = histx[tid]; fcallA[tid] = m_ == INT_MAX ? INT_MAX : (m_ >> 6); BI(bl::FMINP)[tid] = m_;
            }
            if (tid == 0) {
                scal[B_EVENTS] += qn + n1; scal[B_JOINS] += nj; // (reported, never read by the simulation)
                scal[B_NS] = 0; scal[B_NRC] = 0; scal[B_NRJ] = 0;
                ecnt[slot] = 0; ecnt[CR + slot] = 0; jcnt[slot] = 0; // (this slot is the subframe t + calmask + 1 from here on: nothing is scheduled that far ahead)
            }
        }
        BSTAMP(5); // leaver filter
        __syncthreads(); // S2
        BSTAMP(6);
        if (scal[B_OVF]) { status = PRACH_ERR_INTERNAL; why = scal[B_OVF] == 2 ? 6 : 5; time_exit = t; break; } // a calendar list (or the granted list, a subframe ago) was full
        const int N = scal[B_NEV + parity];
        if (tid < NPB) { histx[tid] = 0; mlocx[tid] = INT_MAX; } // (this ring slot is the subframe t + HRING from here on: joined at t + 2 at the earliest)
        if (tid == 64) scal[B_NGL + (parity ^ 1)] = 0;
        const int nsucc_tot = scal[B_NSUCC];
        // classify the events against the lowest DEFINITE caller of every bucket
        for (int k = tid; k < N; k += TB) {
            const int2 ev = ev_get(k);
            const int type = ev.y & 7, p = (ev.y >> 4) & 0xff;
            if (type == EVB_RESETCAND) {
                // a call on its old bucket by a definite caller with a lower index bumps it: cannot re-join (99.7 % of them)
                if (fcallA[(ev.y >> 12) & 0xff] < ev.x) ev_kill(k);
                else { const int s = atomicAdd(&scal[B_NRC], 1); if (s < RCCAP) BI(bl::RCL)[s] = k; }
            } else if (type == EVB_RJOIN) {
                atomicAdd(&scal[B_NRJ], 1);
            } else if (type == EVB_LEAVER) {
                if (ev.x < fcallA[p]) atomicAdd(&BI(bl::NLV)[p], 1);
            } else if (type == EVB_CALLER) {
                if (ev.x == fcallA[p]) BI(bl::FIE)[p] = 1;
            }
        }
        BSTAMP(7); // classify
        if (N > 0) __syncthreads(); // S4 (N is uniform)
        BSTAMP(8);

        // ---- resolve ----
        const int nrc = scal[B_NRC];
        if (nrc > 0) { // rare: reset cycles that may re-join — decided strictly in index order by one wavefront, then recount
            if (nrc > RCCAP) { status = PRACH_ERR_INTERNAL; why = 2; time_exit = t; break; }
            if (tid < 64) { // (first-caller table in registers, lane = bucket: prach_cluster.hip resolve_reset_candidates)
                const int n = __builtin_amdgcn_readfirstlane(nrc);
                int *const rcl = BI(bl::RCL), *const sidx = BI(bl::SIDX);
                int f0 = lane < nP ? fcallA[lane] : INT_MAX;
                for (int c = lane; c < n; c += 64) { // rank-sort the candidates by UE index into SIDX (free at this point)
                    const int myidx = ev_get(rcl[c]).x;
                    int rank = 0;
                    for (int j = 0; j < n; j++) rank += ev_get(rcl[j]).x < myidx ? 1 : 0;
                    sidx[rank] = rcl[c];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int base = 0; base < n; base += 64) {
                    const int mm = min(64, n - base);
                    int es = 0, cidx = 0, cinfo = 0;
                    if (lane < mm) { es = sidx[base + lane]; const int2 e = ev_get(es); cidx = e.x; cinfo = e.y; }
                    int cancelled = 0;
                    for (int s_ = 0; s_ < mm; s_++) {
                        const int idx = __builtin_amdgcn_readlane(cidx, s_), info = __builtin_amdgcn_readlane(cinfo, s_);
                        const int p = (info >> 4) & 0xff, q = (info >> 12) & 0xff;
                        if (__builtin_amdgcn_readlane(f0, q & 63) < idx) { if (lane == s_) cancelled = 1; } // bumped before its turn
                        else if (idx < __builtin_amdgcn_readlane(f0, p & 63)) { if (lane == (p & 63)) f0 = idx; } // its call becomes the first on p
                    }
                    if (lane < mm && cancelled) ev_kill(es);
                }
                if (lane < nP) fcallA[lane] = f0;
            } else if (tid < 64 + NPB) { BI(bl::NLV)[tid - 64] = 0; BI(bl::FIE)[tid - 64] = 0; }
            __syncthreads();
            for (int k = tid; k < N; k += TB) {
                const int2 e = ev_get(k);
                const int type = e.y & 7, p = (e.y >> 4) & 0xff;
                if (type == EVB_LEAVER) { if (e.x < fcallA[p]) atomicAdd(&BI(bl::NLV)[p], 1); }
                else if ((type == EVB_CALLER || type == EVB_RESETCAND) && e.x == fcallA[p]) BI(bl::FIE)[p] = 1;
            }
            __syncthreads();
        }
        // every call: scan count `check` (Beta.c:321-330), counters (Beta.c:334,349-351 / WithNOMA:650-652)
        {
            const int nrj = scal[B_NRJ];
            int my_coll = 0, my_txop = 0;
            for (int k = tid; k < N + nP; k += TB) {
                int idx = 0, p = 0, ispre = 0;
                bool caller = false;
                if (k < N) {
                    const int2 e = ev_get(k);
                    const int type = e.y & 7;
                    if (type == EVB_CALLER || type == EVB_RESETCAND) { caller = true; idx = e.x; p = (e.y >> 4) & 0xff; ispre = (e.y >> 3) & 1; }
                } else {
                    p = k - N;
                    if (fcallA[p] != INT_MAX && !BI(bl::FIE)[p]) { caller = true; idx = fcallA[p]; ispre = 1; } // a matched UE calls first
                }
                if (!caller) continue;
                const bool first = idx == fcallA[p];
                int rj = 0;
                if (nrj > 0) { // Msg3-timeout re-entries that stayed matched since the previous call on this bucket (rare)
                    int prev = (!first) ? fcallA[p] : -1;
                    for (int j = 0; j < N; j++) {
                        const int2 ej = ev_get(j);
                        const int tj = ej.y & 7;
                        if ((tj == EVB_CALLER || tj == EVB_RESETCAND) && ((ej.y >> 4) & 0xff) == p && ej.x < idx && ej.x > prev) prev = ej.x;
                    }
                    for (int j = 0; j < N; j++) {
                        const int2 ej = ev_get(j);
                        if ((ej.y & 7) == EVB_RJOIN && ((ej.y >> 4) & 0xff) == p && ej.x < idx && ej.x > prev) rj++;
                    }
                }
                const int check = 1 + (first ? BI(bl::TOTAL)[p] - ispre - BI(bl::NLV)[p] : 0) + rj;
                if (lcallA[p] < idx) atomicMax(&lcallA[p], idx);
                if (check == 1) {
                    const int s = atomicAdd(&scal[B_NS], 1);
                    if (s < BSC) BI(bl::SIDX)[s] = idx | (p << 20);
                    my_txop += 1;
                } else if (withnoma) { // WithNOMA:650-652
                    my_coll += check; my_txop += check;
                } else { // Beta.c:349-351
                    my_coll += 1; my_txop += 1;
                }
            }
            if (__any((my_coll | my_txop) != 0)) {
                my_coll = wave_sum(my_coll); my_txop = wave_sum(my_txop);
                if (lane == 0) { if (my_coll) atomicAdd(&scal[B_COLL], my_coll); if (my_txop) atomicAdd(&scal[B_TXOP], my_txop); }
            }
        }
        BSTAMP(9); // calls
        __syncthreads(); // S5: calls done; singles listed
        BSTAMP(10);
        if (tid == 0) { scal[B_NEV + parity] = 0; scal[B_NCAND + parity] = 0; } // (every thread has read them; the next subframe appends to the other pair)
        const int ns = scal[B_NS];
        if (ns > BSC) { status = PRACH_ERR_INTERNAL; why = 3; time_exit = t; break; }
        const int Gr = max(0, nGrantUL - 1 - grantCheck); // Beta.c:336-347
        // An UL grant (Beta.c:338-343) for the singleton caller `my` of bucket bp: into this subframe's granted set and the next subframe's event list; a
        // mid-window UE (the bucket's lowest matched index as the ring held it, with subframes of its window still ahead) leaves the ring's later
        // subframes — it was the bucket's only member, so its contributions there are exactly one count and the lowest index
        auto grant = [&](const int my, const int bp) {
            atomicOr(&(BU(bl::GBM) + parity * (GBITS / 32))[(my & (GBITS - 1)) >> 5], 1u << (my & 31));
            const int gp = atomicAdd(&scal[B_NGL + parity], 1);
            if (gp < GLCAP) (BI(bl::GLIST) + parity * GLCAP)[gp] = my; else scal[B_OVF] = 1;
            const int es_ = (int)((unsigned)(t + 1) & calmask), ep = atomicAdd(&ecnt[CR + es_], 1);
            if (ep < calcap) ecal[(size_t)es_ * (size_t)calcap + (size_t)(calcap - 1 - ep)] = my | CAL_GRANT; else scal[B_OVF] = 1;
            const int fm = BI(bl::FMINP)[bp];
            if ((fm >> 6) == my && fm != INT_MAX) {
                const int rem = fm & 63;
                for (int k = 1; k <= rem; k++) {
                    const int rs = ((t + k) & (HRING - 1)) * NPB + bp;
                    atomicSub(&hr[rs], 1);
                    atomicCAS(&mr[rs], (my << 6) | (rem - k), INT_MAX);
                }
            }
        };
        // a caller's sector: Philox — a function of the UE's own first activation draw, recomputed; the reference's stream — kept in the UE's record
        auto sector_of = [&](const int my) -> int {
            if (GLIBC) return rec32[2 * (size_t)my + 1].x & 0xFF;
            return sector_of_draw(philox_draw31(seed_lo, seed_hi, (unsigned)my, 0u, (unsigned)nUE, (unsigned)variant));
        };
        if (sectors) {
            // the dormant per-sector grant test (WithNOMA:626-637 with the call of :312): every 60-degree sector has its own budget of the 5 ms
            // window, grantCheck[sector] counts that sector's singleton callers.  A caller's sector is a function of its first activation draw
            // (sector_of above).  Up to one wavefront of singleton callers: ranked through v_readlane.
            if (ns > 0 && ns <= 64) {
                if (tid < 64) {
                    const int nsu = __builtin_amdgcn_readfirstlane(ns);
                    const bool have = tid < nsu;
                    const int sp = have ? BI(bl::SIDX)[tid] : 0;
                    const int my = have ? (sp & 0xFFFFF) : INT_MAX;
                    const int sec = have ? sector_of(my) : 7;
                    int rank = 0;
                    for (int s_ = 0; s_ < nsu; s_++) rank += (__builtin_amdgcn_readlane(my, s_) < my && __builtin_amdgcn_readlane(sec, s_) == sec) ? 1 : 0;
                    if (have && rank < nGrantUL - 1 - scal[B_SGC + sec]) grant(my, (sp >> 20) & 63);
                    int mine = 0;
#pragma unroll
                    for (int s_ = 0; s_ < 6; s_++) { const int c = __popcll(__ballot(sec == s_)); mine = tid == s_ ? c : mine; }
                    if (tid < 6) scal[B_SGC + tid] += mine; // (read above by this wavefront only, in program order)
                }
            } else if (ns > 0) {
                // more than a wavefront of singleton callers (several per preamble: callers whose bucket's other members have left, Beta.c:321-330): the
                // bin selection below with the key (sector, index): sector s owns the bins [s SB, (s + 1) SB), eight times coarser in the index
                int *const bins = BI(bl::BINS), *const sidx = BI(bl::SIDX), *const rcl = BI(bl::RCL), *const wtot = BI(bl::WTOT);
                constexpr int PER = BGB / TB, SB = BGB / 8;
                const int shift = binshift + 3;
#pragma unroll
                for (int u_ = 0; u_ < PER; u_++) bins[tid * PER + u_] = 0;
                if (tid == 0) scal[B_NCROSS] = 0;
                __syncthreads();
                for (int j = tid; j < ns; j += TB) {
                    const int sp = sidx[j], my = sp & 0xFFFFF;
                    const int sec = sector_of(my);
                    sidx[j] = sp | (sec << 26); // (UE indices have 20 bits, the bucket 6)
                    atomicAdd(&bins[sec * SB + (my >> shift)], 1);
                }
                __syncthreads();
                {
                    int c[PER], sum = 0;
#pragma unroll
                    for (int u_ = 0; u_ < PER; u_++) { c[u_] = bins[tid * PER + u_]; sum += c[u_]; }
                    const int x = wave_scan_incl(sum);
                    if (lane == 63) wtot[w] = x;
                    __syncthreads();
                    int run = x - sum;
                    for (int k = 0; k < w; k++) run += wtot[k];
#pragma unroll
                    for (int u_ = 0; u_ < PER; u_++) { bins[tid * PER + u_] = run; run += c[u_]; } // exclusive prefix
                }
                __syncthreads();
                for (int j = tid; j < ns; j += TB) {
                    const int e_ = sidx[j], my = e_ & 0xFFFFF, sec = e_ >> 26;
                    const int bin = sec * SB + (my >> shift);
                    const int Gs = nGrantUL - 1 - scal[B_SGC + sec];
                    const int before = bins[bin] - bins[sec * SB]; // singleton callers of this sector in lower bins
                    if (before >= Gs) continue;
                    const int cnt = (bin + 1 < BGB ? bins[bin + 1] : ns) - bins[bin];
                    if (before + cnt <= Gs) grant(my, (e_ >> 20) & 63);
                    else { const int s_ = atomicAdd(&scal[B_NCROSS], 1); if (s_ < RCCAP) rcl[s_] = e_; }
                }
                __syncthreads();
                const int ncross = scal[B_NCROSS];
                if (ncross > RCCAP) { status = PRACH_ERR_INTERNAL; why = 4; time_exit = t; break; }
                if (tid < ncross) {
                    const int e_ = rcl[tid], my = e_ & 0xFFFFF, sec = e_ >> 26;
                    const int bin = sec * SB + (my >> shift);
                    int rank = bins[bin] - bins[sec * SB];
                    for (int m = 0; m < ncross; m++) { const int o = rcl[m]; rank += ((o >> 26) == sec && ((o & 0xFFFFF) >> shift) == (my >> shift) && (o & 0xFFFFF) < my) ? 1 : 0; }
                    if (rank < nGrantUL - 1 - scal[B_SGC + sec]) grant(my, (e_ >> 20) & 63);
                }
                __syncthreads(); // (the budgets were read above)
                if (tid < 6) scal[B_SGC + tid] += bins[(tid + 1) * SB] - bins[tid * SB];
            }
        } else if (Gr > 0 && ns > 0 && ns <= 64) {
            if (tid < 64) { // up to one wavefront of singleton callers: every lane ranks its own index against the others through v_readlane
                const int nsu = __builtin_amdgcn_readfirstlane(ns);
                const int sp = tid < nsu ? BI(bl::SIDX)[tid] : 0;
                const int my = tid < nsu ? (sp & 0xFFFFF) : INT_MAX;
                int rank = 0;
                for (int s_ = 0; s_ < nsu; s_++) rank += __builtin_amdgcn_readlane(my, s_) < my ? 1 : 0;
                if (tid < nsu && rank < Gr) grant(my, (sp >> 20) & 63);
            }
        } else if (Gr > 0 && ns > 0) {
            // the Gr lowest-index singleton callers, in O(ns): counts per index bin, block-wide exclusive prefix, whole bins below the
            // crossing bin are granted, the crossing bin is ranked exactly
            int *const bins = BI(bl::BINS), *const sidx = BI(bl::SIDX), *const rcl = BI(bl::RCL), *const wtot = BI(bl::WTOT);
            constexpr int PER = BGB / TB; // consecutive bins per thread
#pragma unroll
            for (int u_ = 0; u_ < PER; u_++) bins[tid * PER + u_] = 0;
            if (tid == 0) scal[B_NCROSS] = 0;
            __syncthreads();
            for (int j = tid; j < ns; j += TB) atomicAdd(&bins[(sidx[j] & 0xFFFFF) >> binshift], 1);
            __syncthreads();
            {
                int c[PER], sum = 0;
#pragma unroll
                for (int u_ = 0; u_ < PER; u_++) { c[u_] = bins[tid * PER + u_]; sum += c[u_]; }
                const int x = wave_scan_incl(sum);
                if (lane == 63) wtot[w] = x;
                __syncthreads();
                int run = x - sum;
                for (int k = 0; k < w; k++) run += wtot[k];
#pragma unroll
                for (int u_ = 0; u_ < PER; u_++) { bins[tid * PER + u_] = run; run += c[u_]; } // exclusive prefix
            }
            __syncthreads();
            for (int j = tid; j < ns; j += TB) {
                const int sp = sidx[j], my = sp & 0xFFFFF;
                const int bin = my >> binshift;
                const int before = bins[bin];
                if (before >= Gr) continue;
                const int cnt = (bin + 1 < BGB ? bins[bin + 1] : ns) - before;
                if (before + cnt <= Gr) grant(my, (sp >> 20) & 63);
                else { const int s_ = atomicAdd(&scal[B_NCROSS], 1); if (s_ < RCCAP) rcl[s_] = sp; }
            }
            __syncthreads();
            const int ncross = scal[B_NCROSS];
            if (ncross > RCCAP) { status = PRACH_ERR_INTERNAL; why = 4; time_exit = t; break; }
            if (tid < ncross) {
                const int sp = rcl[tid], my = sp & 0xFFFFF;
                int rank = bins[my >> binshift];
                for (int m = 0; m < ncross; m++) rank += (rcl[m] & 0xFFFFF) < my ? 1 : 0;
                if (rank < Gr) grant(my, (sp >> 20) & 63);
            }
        }
        grantCheck += ns;
        BSTAMP(11); // grants
        if ((Gr > 0 || sectors) && ns > 0) __syncthreads(); // S6: the grants are in the next subframe's event list and in the granted set before that subframe begins
        BSTAMP(12);
        if (nsucc_tot == nUE) { time_exit = t; break; } // Beta.c:180
    }
    __syncthreads();
    if (status == PRACH_OK && scal[B_OVF]) { status = PRACH_ERR_INTERNAL; why = 5; time_exit = tlast; } // (raised by the last subframe's grants)

    // ---- the state after the last subframe (deferred outcome + the subframes a UE was matched in since its record was written),
    // end-of-trial sums (Beta.c:185-197) and the logged fields (Beta.c:501-508)
    const int tend = tlast + 1;
    {
        const CallTables tab{BI(bl::FCALL) + (tlast & 1) * NPB, BI(bl::LCALL) + (tlast & 1) * NPB};
        PRACH_G int *const timers = (PRACH_G int *)PD->timers;
        PRACH_G v4i_t *const logs = (PRACH_G v4i_t *)PD->logs;
        long long sumT = 0;
        int ptcS = 0, fcS = 0;
        unsigned long long ndS = 0;
        for (int i = tid; i < nUE; i += TB) {
            const BRec R = brec_load(rec32 + 2 * (size_t)i);
            UeState u = unpack(R.a);
            const ColdRegs cold = cold_unpack(R.b);
            if (status == PRACH_OK && tlast >= 0 && u.act != ACT_IDLE)
                pw_catch_up(u, (unsigned)R.b.w, granted_in(tlast & 1, i), i, tend, K.fmA, tab);
            const int timer = u.act == ACT_IDLE ? -1 : (u.act == ACT_DONE ? u.tb : tend - u.tb);
            if (u.act == ACT_DONE) { sumT += timer; ptcS += cold.ptc; fcS += cold.fcnt; }
            ndS += (unsigned)R.b.x & 0xFFFFFFu;
            timers[i] = u.act == ACT_DONE ? timer : INT_MIN;
            if (logs) {
                prach_ue_log o;
                o.idx = i; o.timer = timer; o.active = u.act - 1; o.txTime = u.tx; o.firstTxTime = cold.ftt;
                o.secondTxTime = cold.stt; o.nowBackoff = now_backoff(u.bo, tend); o.preamble = u.pre - 1;
                o.preambleChange = u.pre != 0; o.rarWindow = u.rar; o.maxRarCounter = u.mrc; o.preambleTxCounter = cold.ptc;
                o.msg2Flag = (u.act == ACT_M3 || u.act == ACT_DONE); o.connectionRequest = u.conn == 2 ? 48 : u.conn;
                o.msg4Flag = u.act == ACT_DONE; o.failCount = cold.fcnt;
                store_log(logs, i, o);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            sumT += __shfl_down(sumT, d); ptcS += __shfl_down(ptcS, d); fcS += __shfl_down(fcS, d); ndS += __shfl_down(ndS, d);
        }
        if (lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&scal[B_SUMT]), (unsigned long long)sumT);
            atomicAdd(reinterpret_cast<unsigned long long *>(&scal[B_ND]), ndS);
            atomicAdd(&scal[B_PTC], ptcS);
            atomicAdd(&scal[B_FC], fcS);
        }
    }
    __syncthreads();
    if (tid == 0) { // DevResult was zeroed by the engine before the launch
        PRACH_G DevResult *o = (PRACH_G DevResult *)PD->out;
        o->sumTimer = *reinterpret_cast<long long *>(&scal[B_SUMT]);
        o->draws = GLIBC ? base : *reinterpret_cast<unsigned long long *>(&scal[B_ND]);
        o->ptcSum = scal[B_PTC]; o->fcSum = scal[B_FC];
        o->nSuccess = scal[B_NSUCC]; o->finalSuccess = scal[B_NSUCC]; o->continueFailed = scal[B_CONTF];
        o->status = status;
        o->hard_error = why; // (2 reset-cycle candidates, 3 singleton callers, 4 crossing bin, 5 a calendar list or the granted list: reported by the engine)
        o->time_exit = time_exit;
        o->collisionPreambles = scal[B_COLL]; o->totalPreambleTxop = scal[B_TXOP];
        o->activeCheck = activeCheck;
        o->steps = steps;
        o->visits = (unsigned long long)(unsigned)scal[B_JOINS]; // (join-calendar entries: one per contention window)
        o->events = (unsigned long long)(unsigned)scal[B_EVENTS];
#ifdef PRACH_STAMPS
        for (int k = 0; k < 24; k++) o->fstamps[k] = fstamps[k];
#endif
    }
}

size_t batch_kernel_lds_bytes(int waves, bool glibc) { return glibc ? (size_t)BLG::END : waves == 8 ? (size_t)BL<8>::END : (size_t)BL<16>::END; }
int batch_max_preambles() { return NPB; }
int batch_max_rar_window() { return 64; }
int batch_max_subframes() { return 65000; }
int batch_max_groups(bool glibc) { return glibc ? BGG : (1 << 14); }
int batch_max_rar_window_two_per_cu() { return BCap<8>::HRING; }
int batch_calendar_slots(int backoff, int accessTime, int maxRarWindow) { // power of two >= the furthest a UE is ever scheduled ahead (+ its window), at most CR
    const int need = backoff + (accessTime > 5 ? accessTime : 5) + maxRarWindow + 70;
    int r = 64;
    while (r < need) r *= 2;
    return r;
}
int batch_max_calendar_slots() { return CR; }

// waves: wavefronts per workgroup — 8: 512 threads, two workgroups (two independent trials) per CU; 16: 1024 threads, one per CU
// glibc: the trials draw from the reference's own rand() stream (1024 threads only)
hipError_t launch_batch_kernel(const TrialDev *params, int ntrials, int waves, bool glibc, hipStream_t stream) {
    if (glibc) waves = 16;
    const size_t lds = batch_kernel_lds_bytes(waves, glibc);
    const void *fn = glibc ? reinterpret_cast<const void *>(&batch_kernel<16, true>)
                           : waves == 8 ? reinterpret_cast<const void *>(&batch_kernel<8>) : reinterpret_cast<const void *>(&batch_kernel<16>);
    hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return rc;
    if (glibc) hipLaunchKernelGGL((batch_kernel<16, true>), dim3(ntrials), dim3(1024), lds, stream, params);
    else if (waves == 8) hipLaunchKernelGGL(batch_kernel<8>, dim3(ntrials), dim3(512), lds, stream, params);
    else hipLaunchKernelGGL(batch_kernel<16>, dim3(ntrials), dim3(1024), lds, stream, params);
    return hipGetLastError();
}

} // namespace prach
