// prach_batch.hip — the BATCHED regime: one workgroup per trial, thousands of trials in flight (BASELINE configs[2] / [4]: the
// --times x nUE sweep).  The state of the in-flight trials lives in HBM, so what a UE costs is what is READ and WRITTEN of it and HOW.
//
// Round 3 walked one 32-bit pass word per live UE per subframe; the first round-4 form replaced the walk by calendars of UE indices and kept one
// 32-byte record per UE at its own address — and ran, like round 3, at the rate at which an MI355X reads and writes random 32-byte records
// (profiles/tools/gather_probe.hip: 2.0-2.4e10 records/s from HBM, the kernel: 1.9e10 events/s; extra Philox draws or LDS atomics per event cost
// nothing, one more scattered store per event 4 %: profiles/r04_grid.md).  Hence this form: THE STATE TRAVELS WITH THE EVENT.
//
// A UE in steady contention is bumped every subframe (Beta.c:346,358) and counts one RAR-window subframe each time (Beta.c:245), so when the
// event body lets go of a UE its whole trajectory until its next event is known: matched by preambleCollision scans in [tj, tj + dur), an
// event at tj + dur (prach_ue_body.h: pw_schedule).  The trial keeps, per future subframe (slot = subframe & calmask):
//   * the EVENT LIST of the subframe: the 32-byte records of the UEs whose next event falls into it, in CHUNKS of 64 records (2 KB, two planes of
//     16 bytes per UE).  The event body of a subframe reads its chunks — every wavefront-instruction a contiguous kilobyte —, and each wavefront
//     appends the records it has finished to ITS OWN open chunk of the subframe they are scheduled into (its open chunks and a stack of free chunk
//     ids sit in three of its vector registers: no shared cursor, nothing to wait for); a full chunk is entered into the subframe's chunk table.
//     A UE's record is in exactly one place at any time — no random read, no random write, no stale copy;
//   * the JOIN LIST: the indices of the UEs whose contention window opens in the subframe — at tj the UE adds itself to the per-bucket histogram
//     and lowest-index tables of the subframes tj .. tj + dur - 1, a ring of 16 subframes in LDS (dur x 2 LDS atomics ONCE per window).
// What a subframe costs is therefore its EVENTS, streamed, not its live UEs and not random records.
//   * An UL grant (the resolver, Beta.c:336-347) takes a UE out of contention early.  Its record is somewhere in a later subframe's list, so the grant
//     is only NOTED — per bucket and subframe, in a ring of the last 16 subframes (LDS) — and applied when the record comes up, as if it had been
//     looked at one subframe after the grant (nothing happens to a granted UE before Msg3, ten subframes later: hence maxRarWindow <= 11 here); a
//     granted mid-window UE — necessarily the only member of its bucket — is taken out of the histogram ring's later subframes by the granting
//     thread (its remaining window rides in the low bits of the lowest-index word), a UE granted in the subframe it was scheduled in skips its join.
//   * A UE that has finished for good (or whose txTime never comes) is written to its record's home (PD->rec32) — once per UE.
//   * The event body is prach_ue_body.h (shared with every other kernel); the resolver is prach_cluster.hip's for one workgroup.
//   * No per-subframe capacity on the resolver's event list or the leaver candidates (they continue in global memory); the chunk pool and the join
//     lists are sized by the engine — a trial that exhausts them leaves with PRACH_ERR_INTERNAL and is rerun with full-size ones, reported.
//   * batch_kernel<16, true>: the same in the reference's own rand() stream (what `prach_sim -t 100` issues per sweep point) — the event body
//     runs as a count pass + a block-wide prefix over the 64-UE groups + a select pass, see the kernel's head.
//   * PRACH_FLAG_SECTOR_GRANTS (WithNOMA:626-637): six grant budgets in the grant phase.
// Limits (the engine falls back to prach::cluster_kernel): nPreamble <= 64, maxRarWindow <= 11, < 65 000 subframes, backoff + accessTime +
// maxRarWindow + 70 <= 256 (the calendar's horizon); the reference-stream form: 131 072 UEs.
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

#ifndef PRACH_B_W8_WAVES
#define PRACH_B_W8_WAVES 4 // wavefronts per SIMD the 512-thread shape is compiled for: 4 = two workgroups per CU (6 = three: measured level, with spills)
#endif
#ifndef PRACH_B_XVALU
#define PRACH_B_XVALU 0 // (sensitivity experiment) extra Philox draws per event batch, results unused
#endif
#ifndef PRACH_B_XSALU
#define PRACH_B_XSALU 0 // (sensitivity experiment) extra dependent scalar instructions per event batch
#endif
#ifndef PRACH_B_XBR
#define PRACH_B_XBR 0   // (sensitivity experiment) extra exec-masked short blocks (s_and_saveexec / s_cbranch_execz / s_or) per event batch
#endif
#ifndef PRACH_B_XLDS
#define PRACH_B_XLDS 0  // (sensitivity experiment) extra returning LDS atomics per event batch (on dummy words)
#endif
constexpr int NPB = 64;       // stride of the per-bucket tables (nPreamble <= 64)
constexpr int BSC = 2048;     // singleton callers per subframe
constexpr int BGB = 1024;     // grant selection bins
constexpr int CR = 256;       // calendar slots: the subframes t .. t + 255 (counts in LDS); a trial uses the first calmask + 1 of them
constexpr int HRING = 16;     // subframes ahead the histogram / lowest-index ring and the grant notes hold: a window lasts maxRarWindow - 1 <= 10 subframes
constexpr int CHUNK = 128;    // int4 per chunk: plane A [64] (the hot record of prach_device.h), plane B [64] (see BRec)
constexpr int ROVCAP = 64;    // grants of one subframe beyond the first of their bucket (several singleton callers of one preamble: Beta.c:321-330)
// Workgroup shapes.  NWB wavefronts per workgroup: 16 (1024 threads, one workgroup = one trial per CU) or 8 (512 threads and an LDS
// footprint under 80 KB, so that TWO workgroups = two independent trials share a CU: while one waits at a barrier, the other one issues).
// What is held in LDS per subframe (event list and candidate list continue in global memory; the singleton list BSC and the
// reset-cycle / crossing-bin lists RCCAP are capacities: beyond them the engine reruns the trial on trial_kernel):
template <int NWB> struct BCap {
    static constexpr int EV = NWB == 16 ? 4096 : 1536;    // events of a subframe held in LDS (more: global memory)
#ifdef PRACH_QCAP
    static constexpr int CAND = PRACH_QCAP;                 // (test build: nearly every subframe's candidate list continues in global memory)
#else
    static constexpr int CAND = NWB == 16 ? 4096 : 1024;  // early-leaver candidates of a subframe held in LDS (more: global memory)
#endif
};

constexpr int EVB_CALLER = UEV_CALLER, EVB_RESETCAND = UEV_RESETCAND, EVB_RJOIN = UEV_RJOIN, EVB_LEAVER = 4;

// (the event / candidate counts and the free-chunk pool exist twice, by subframe parity: a wavefront that is already in the next subframe's body uses the other one)
enum { B_NSUCC = 0, B_COLL, B_TXOP, B_CONTF, B_NS, B_NRC, B_NRJ, B_OVF /* the chunk pool, a chunk table or a join list is full: the trial leaves */, B_NEV = 8 /* [2] */, B_NCAND = 10 /* [2] */,
       B_POOLH = 12 /* [2] free chunk ids in the shared pool's two halves */, B_PTC = 14, B_FC, B_SUMT = 16, B_ND = 18, B_JOINS = 20, B_EVENTS = 21, B_NCROSS = 22, B_BUMP = 23 /* chunks never used yet */, B_TICK = 30 /* [2] event batches handed out beyond the wavefronts' own first three (second: the reference stream's select pass) */,
       B_SGC = 24 /* [24, 30): sectorGrants[6], WithNOMA:260 (PRACH_FLAG_SECTOR_GRANTS) */, B_NROV = 32 /* [16] grants beyond the first of their bucket, per subframe of the ring */ };

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
    static constexpr int HR = FMINP + 4 * NPB;             // int [HRING][NPB] histogram ring: matched UEs per bucket in the subframes t .. t + HRING - 1 (row = subframe mod HRING)
    static constexpr int MR = HR + 4 * HRING * NPB;        // int [HRING][NPB] lowest matched index << 6 | remaining window, same ring
    static constexpr int RG = MR + 4 * HRING * NPB;        // int [HRING][NPB] grant notes: the UE granted as its bucket's (first) singleton caller in the subframes t - 15 .. t, -1 none
    static constexpr int BM = RG + 4 * HRING * NPB;        // unsigned [NPB] per bucket, bit (subframe mod 16): a grant note exists
    static constexpr int ROV = BM + 4 * NPB;               // int [HRING][ROVCAP] further grants of a subframe: index | bucket << 20
    static constexpr int NCHK = ROV + 4 * HRING * ROVCAP;  // int [CR] chunks entered into the event list of the subframes t .. (slot = subframe & calmask)
    static constexpr int JCNT = NCHK + 4 * CR;             // int [CR] entries in the join list of ...
    static constexpr int CANDL = JCNT + 4 * CR;            // int [CAND] early-leaver candidates: index | old bucket << 20
    static constexpr int DUMMY = CANDL + 4 * C::CAND;      // int [64] per-lane dummy words: what a lane adds to / takes the minimum of when it has nothing to contribute
    static constexpr int DUMMY2 = DUMMY + 4 * 64;          // int2 [64] the same for 8-byte entries
    static constexpr int DR = DUMMY2 + 8 * 64;             // int [HRING][NPB] difference ring of the histogram: + 1 in the subframe a UE's contention window opens, - 1 in the one after its last
    static constexpr int ER = DR + 4 * HRING * NPB;        // int [HRING][NPB] per subframe a contention window ENDS in (the first one it no longer covers): the lowest index among the UEs inside such a window
    static constexpr int RUN = ER + 4 * HRING * NPB;       // int [NPB] the differences summed up to this subframe: UEs inside their window, per bucket
    static constexpr int OPENT = RUN + 4 * NPB;            // int [NWB][256] per wavefront: the open chunk of every subframe ahead, (id + 1) << 8 | records in it
    static constexpr int END = OPENT + 4 * 256 * NWB;
    static_assert(SIDX % 16 == 0 && HR % 16 == 0, "alignment");
};
static_assert(BL<8>::END <= 80 * 1024, "two 512-thread workgroups per CU");
// The reference's own rand() stream (GLIBC instantiation, 1024 threads): per 64-UE group of the trial, the lanes that make at least one / two rand()
// calls in this subframe (two 64-bit masks) and the group's exclusive prefix of calls in index order.  BGG groups = 131 072 UEs at most.
constexpr int BGG = 2048;
struct BLG { static constexpr int GM = (BL<16>::END + 15) / 16 * 16, GPRE = GM + 16 * BGG, END = GPRE + 4 * BGG; };
static_assert(BLG::END <= 160 * 1024 && BLG::GM % 16 == 0, "LDS");

#define BI(off) (reinterpret_cast<int *>(smem + (off)))
#define BU(off) (reinterpret_cast<unsigned *>(smem + (off)))
#define BI2(off) (reinterpret_cast<int2 *>(smem + (off)))

// the 32-byte record of a UE: A = the hot record of prach_device.h {txTime, timer base, nowBackoff, packed}; B = {Philox draw index (24 bits) | the
// length of the window it was scheduled with << 24, preambleTxCounter | failCount << 16, secondTxTime | firstTxTime << 16, the UE's index} — in a chunk:
// A in plane A, B in plane B, at the same position; at home (PD->rec32, finished and idle UEs): word 7 is PW_DONE / PW_IDLE instead of the index
struct BRec { int4 a, b; };
__device__ __forceinline__ int4 ld_i4(const PRACH_G v4i_t *p) { const v4i_t v = *p; return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st_i4(PRACH_G v4i_t *p, const int4 a) { v4i_t v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w; *p = v; }
__device__ __forceinline__ ColdRegs cold_unpack(const int4 b) {
    ColdRegs c;
    c.ptc = b.y & 0xffff; c.fcnt = (int)((unsigned)b.y >> 16); c.stt = b.z & 0xffff; c.ftt = (int)((unsigned)b.z >> 16);
    return c;
}
__device__ __forceinline__ int4 cold_pack(const unsigned nd, const unsigned sdur, const ColdRegs &c, const int w7) {
    return make_int4((int)((nd & 0xFFFFFFu) | (sdur << 24)), (c.ptc & 0xffff) | (c.fcnt << 16), (c.stt & 0xffff) | (c.ftt << 16), w7);
}
// Join list entries: UE index [19:0] | bucket [25:20] | window length [31:26].  Chunk table entries: chunk id [23:0] | records in it [30:24].

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
    constexpr int TB = NWB * 64, BEV = BCap<NWB>::EV, CCAP = BCap<NWB>::CAND;
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
    PRACH_G v4i_t *const rec32 = (PRACH_G v4i_t *)PD->rec32;     // [nUE][2] a UE's record at home: before it arrives, and once it has finished
    PRACH_G v4i_t *const chunks = (PRACH_G v4i_t *)PD->chunks;   // [nchunks][CHUNK] the pool of 2 KB chunks
    PRACH_G int *const ctab = (PRACH_G int *)PD->ctab;           // [calmask + 1][tcap] chunk table: the chunks of every future subframe's event list
    const int tcap = PD->tcap;                                   // (PD->cpool [2][nchunks]: the shared pool of free chunk ids, what the wavefronts' own stacks cannot take)
    PRACH_G int *const jcal = (PRACH_G int *)PD->jcal;           // [calmask + 1][calcap] join lists
    const int calcap = PD->calcap;
    const unsigned calmask = (unsigned)PD->calmask;
    PRACH_G int *const candg = (PRACH_G int *)PD->qov;           // [nUE] early-leaver candidates of a subframe beyond their LDS part
    PRACH_G int *const sect_arr = (PRACH_G int *)PD->sector;     // [nUE] (PRACH_FLAG_SECTOR_GRANTS in the reference's stream only) the UE's sector
    const PRACH_G int *const sched = (const PRACH_G int *)PD->sched;
    int *const scal = BI(bl::SCAL);
    int *const lds = BI(0); // (the whole LDS as words: tables addressed by a selected word offset, bl::X / 4 + index)
    int2 *const gev = BI2(bl::GEV);
    // the event list of a subframe: BEV entries in LDS, the rest (event storms of extreme parameter sets) in global memory — no capacity
    PRACH_G v2i_t *const evov = (PRACH_G v2i_t *)PD->evov;
    auto ev_get = [&](const int k) -> int2 { if (k < BEV) return gev[k]; const v2i_t v = evov[k - BEV]; return make_int2(v.x, v.y); };
    auto ev_set = [&](const int k, const int a, const int b) { if (k < BEV) gev[k] = make_int2(a, b); else store_i2(&evov[k - BEV], a, b); };
    auto ev_kill = [&](const int k) { if (k < BEV) gev[k].y = 0; else evov[k - BEV].y = 0; };
    int *const nchk = BI(bl::NCHK), *const jcnt = BI(bl::JCNT);
    int *const hr = BI(bl::HR), *const mr = BI(bl::MR), *const rg = BI(bl::RG), *const rov = BI(bl::ROV);
    unsigned *const bmk = BU(bl::BM);
    int *const candl = BI(bl::CANDL);
    unsigned *const gm = BU(BLG::GM);  // (GLIBC only) [BGG][4]: lanes with >= 1 call (two words), lanes with 2 calls (two words)
    int *const gpre = BI(BLG::GPRE);   // (GLIBC only) [BGG]
    const PRACH_G int *const stream = (const PRACH_G int *)PD->stream;
    const unsigned long long stream_len = PD->stream_len;
    unsigned long long base = 0; // GLIBC: rand() calls consumed so far (relative to the stream window)

    const int totgroups = (nUE + 63) >> 6;
    // The event body's constants (divisor magics, list capacities, the pointers the body stores through) are wave-uniform: as such they would sit in scalar
    // registers across the whole step loop, where the budget of 102 is spent twice over (~100 spilled scalars: a v_readlane / v_writelane pair around every
    // use) — and the kernel is bound by instruction issue (profiles/r04_grid.md).  Held in vector registers they are a plain VALU operand.  The Philox key does NOT
    // live there: see the draws in the event body.
#define B_TO_VGPR(x) asm volatile("" : "+v"(x))
    int vcalcap = calcap;
    B_TO_VGPR(vcalcap);
    B_TO_VGPR(K.fmP.d); B_TO_VGPR(K.fmP.M); B_TO_VGPR(K.fmB.d); B_TO_VGPR(K.fmB.M); B_TO_VGPR(K.fmA.d); B_TO_VGPR(K.fmA.M); // (fm5: literals)
    B_TO_VGPR(K.maxMsg2);
    PRACH_G int *vjcal = jcal;
    B_TO_VGPR(vjcal);
    int vtrash = (PD->nchunks - 1) * CHUNK, vjtrash = ((int)calmask + 1) * calcap; // where lanes with nothing to store aim: the pool's last chunk (never handed out), 64 words behind the join lists
    B_TO_VGPR(vtrash); B_TO_VGPR(vjtrash);
#undef B_TO_VGPR
    // ... and what only the rare paths need (the shared chunk pool, the global parts of the lists, a finished UE's home) is read from the parameter block where it
    // is used — an opaque copy of the block's address keeps the compiler from hoisting those loads back out of the step loop into scalar registers
    auto rare = [&]() __attribute__((always_inline)) -> const TrialDev * { const TrialDev *q = PD; asm volatile("" : "+s"(q)); return q; };
    // calloc + initialUE (Beta.c:78-83)
    for (int i = tid; i < nUE; i += TB) { st_i4(rec32 + 2 * (size_t)i, make_int4(-1, 0, 0, 0)); st_i4(rec32 + 2 * (size_t)i + 1, make_int4(0, 0, 0, (int)PW_IDLE)); }
    for (int k = tid; k < HRING * NPB; k += TB) { hr[k] = 0; mr[k] = INT_MAX; rg[k] = -1; BI(bl::DR)[k] = 0; BI(bl::ER)[k] = INT_MAX; }
    for (int k = tid; k < CR; k += TB) { nchk[k] = 0; jcnt[k] = 0; }
    for (int k = tid; k < 256 * NWB; k += TB) BI(bl::OPENT)[k] = 0;
    if (tid < NPB) {
        BI(bl::TOTAL)[tid] = 0; BI(bl::RUN)[tid] = 0; BI(bl::NLV)[tid] = 0; BI(bl::FIE)[tid] = 0; BI(bl::FMINP)[tid] = INT_MAX; bmk[tid] = 0u;
        BI(bl::FCALL)[tid] = INT_MAX; BI(bl::FCALL)[NPB + tid] = INT_MAX; BI(bl::LCALL)[tid] = -1; BI(bl::LCALL)[NPB + tid] = -1;
    }
    if (tid < 64) scal[tid] = 0;
    if (GLIBC) for (int k = tid; k < 4 * BGG; k += TB) gm[k] = 0u;
    __syncthreads();

    int activeCheck = 0, grantCheck = 0, tlast = -1, time_exit = stop;
    int acNext = sched[0]; // the arrival table's entry of the NEXT access slot: loaded a slot ahead, so that no subframe waits for it (Beta.c:121-134)
    int why = 0;   // which capacity ended the trial (reported)
    unsigned long long steps = 0;
    int status = (nP > NPB || K.maxRar > 11 || stop > 65000 || (GLIBC && totgroups > BGG) || nUE >= (1 << 20) || calmask >= (unsigned)CR ||
                  PD->backoff + max(aT, 5) + K.maxRar + 70 > (int)calmask + 1) ? PRACH_ERR_UNSUPPORTED : PRACH_OK;
#ifdef PRACH_STAMPS
    unsigned long long fstamps[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fprev = __builtin_readcyclecounter();
#endif
    // ---- this wavefront's chunk bookkeeping ----
    // otab[te & 255] = (chunk id + 1) << 8 | records in it: the chunk this wavefront is filling for the subframe te, 0: none.  A UE is scheduled less than calmask + 1
    // <= 256 subframes ahead and the position of subframe t + 1 is emptied in subframe t (chunk_flush): no two live subframes share a position.  A lane takes its
    // place in the chunk with ONE atomic add on the word (finish()); only a chunk that is opened or filled up takes wave-uniform work.
    int *const otab = BI(bl::OPENT) + w * 256;
    int stash = 0;   // lanes [0, sp): free chunk ids (a chunk is free as soon as its records are in registers; the wavefront that read it fills it again)
    int sp = 0;
    int pool_parity = 0; // (this subframe's parity, for the shared pool's halves)
    auto chunk_free = [&](const int id) __attribute__((always_inline)) { // wave-uniform
        if (sp < 64) { stash = lane == sp ? id : stash; sp++; }
        else if (lane == 0) { const int h = atomicAdd(&scal[B_POOLH + pool_parity], 1); { const TrialDev *q = rare(); if (h >= 0 && h < q->nchunks) ((PRACH_G int *)q->cpool)[(size_t)pool_parity * (size_t)q->nchunks + (size_t)h] = id; } }
    };
    auto chunk_alloc = [&]() __attribute__((always_inline)) -> int { // wave-uniform
        if (sp > 0) { sp--; return __builtin_amdgcn_readlane(stash, sp); }
        int id = 0;
        if (lane == 0) { // the shared pool's half that was filled in the previous subframe, then chunks never used before
            const int side = pool_parity ^ 1;
            const int h = atomicSub(&scal[B_POOLH + side], 1);
            const TrialDev *q = rare();
            const int nch_ = q->nchunks;
            if (h > 0 && h <= nch_) id = __hip_atomic_load((PRACH_G int *)q->cpool + (size_t)side * (size_t)nch_ + (size_t)(h - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else {
                id = atomicAdd(&scal[B_BUMP], 1);
                if (id >= nch_ - 1) { scal[B_OVF] = 5; id = nch_ - 1; } // (the pool's last chunk takes what no longer matters: the trial leaves behind S2)
            }
        }
        return __builtin_amdgcn_readfirstlane(id);
    };
    // the chunk joins subframe te's event list (called by the lanes that have one to enter: lane 0 for the wavefront's, every lane its own in the epilogue)
    auto chunk_enter = [&](const int id, const int te, const int count) __attribute__((always_inline)) {
        const int sl = (int)((unsigned)te & calmask);
        const int seq = atomicAdd(&nchk[sl], 1);
        const TrialDev *q = rare(); // (once per chunk: the table's address and capacity are read where they are used)
        const int tc_ = q->tcap;
        if (seq < tc_) ((PRACH_G int *)q->ctab)[(size_t)sl * (size_t)tc_ + (size_t)seq] = id | (count << 24);
        else scal[B_OVF] = 4;
    };
    auto chunk_close = [&](const int id, const int te, const int count) __attribute__((always_inline)) { if (lane == 0) chunk_enter(id, te, count); }; // wave-uniform
    // this wavefront's open chunk of subframe te, if it has one, joins that subframe's list (behind S1 of te - 1: nothing is appended to it any more)
    auto chunk_flush = [&](const int te) __attribute__((always_inline)) {
        const int k = te & 255;
        const int old = __builtin_amdgcn_readfirstlane(otab[k]);
        if (old >> 8) {
            chunk_close((old >> 8) - 1, te, old & 255);
            if (lane == 0) otab[k] = 0;
        }
    };
    // Grant notes.  The ring's bits that stand for the subframes [max(s0, 0), s1] (s1 - s0 < 16, s1 at most 15 subframes old) — n of them, from bit (lo & 15) on,
    // wrapping — and of those the ones bucket p has a note in: one expression, no loop (a grant inside a UE's own window is the rare case)
    auto grant_bits = [&](const int p, const int s0, const int s1) __attribute__((always_inline)) -> unsigned {
        const int lo = max(s0, 0), n = s1 - lo + 1;
        const unsigned ones = n > 0 ? (n >= 16 ? 0xFFFFu : (1u << n) - 1u) : 0u, sh = (unsigned)lo & 15u;
        return bmk[p] & (((ones << sh) | ((ones << sh) >> 16)) & 0xFFFFu);
    };
    // ... and which of them (cand) is UE i's: the subframe of its UL grant, or -1
    auto grant_search = [&](unsigned cand, const int i, const int p, const int s0) __attribute__((always_inline)) -> int {
        const int lo = max(s0, 0);
        int found = -1;
        while (cand) {
            const int b_ = 31 - __clz((int)cand), s = lo + ((b_ - lo) & 15);
            if (rg[b_ * NPB + p] == i) found = s;
            else {
                const int nr = min(scal[B_NROV + b_], ROVCAP);
                for (int k = 0; k < nr; k++) if (rov[b_ * ROVCAP + k] == (i | (p << 20))) found = s;
            }
            cand &= ~(1u << b_);
        }
        return found;
    };
    auto granted_at = [&](const int i, const int p, const int s0, const int s1) -> int { return grant_search(grant_bits(p, s0, s1), i, p, s0); };

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
        pool_parity = parity;
        int *const fcallA = BI(bl::FCALL) + parity * NPB, *const lcallA = BI(bl::LCALL) + parity * NPB;
        int *const fcallB = BI(bl::FCALL) + (parity ^ 1) * NPB, *const lcallB = BI(bl::LCALL) + (parity ^ 1) * NPB;
        const int slot = (int)((unsigned)t & calmask);                       // this subframe's lists
        int *const histx = hr + (t & (HRING - 1)) * NPB, *const mlocx = mr + (t & (HRING - 1)) * NPB; // ... and its histogram / lowest matched index << 6 | window left
        const int nch = nchk[slot], njall = jcnt[slot];                      // complete: every chunk was entered, every join listed in an earlier subframe
        if (njall > calcap) scal[B_OVF] = 1;                                 // (this subframe's join list was too short: the trial leaves behind S2)
        const int nj = min(njall, calcap);
        const int narr = activeCheck - prevAC;                               // this access slot's arrivals (Beta.c:136-146) are events too: batches behind the chunks
        const int nb = nch + ((narr + 63) >> 6);
        const PRACH_G int *const clist = ctab + (size_t)slot * (size_t)tcap;
        BSTAMP(0); // loop head

        // The UEs that are inside a contention window opened EARLIER: the lowest index of every bucket among them, from the windows' end rows — a window that
        // ends in subframe t + k still has k - 1 subframes ahead.  One wavefront (lane = bucket) seeds this subframe's row with it; joins and events lower it further.
        for (int k = 1 + w; k <= 10; k += NWB) { // (a window ends at most maxRarWindow - 1 <= 10 subframes ahead; rows nothing ends in hold INT_MAX; one row per wavefront)
            const int v_ = BI(bl::ER)[((t + k) & (HRING - 1)) * NPB + lane];
            atomicMin(&lds[lsel(lm(v_ != INT_MAX), bl::MR / 4 + (t & (HRING - 1)) * NPB + lane, bl::DUMMY / 4 + lane)], (int)(((unsigned)v_ << 6) | (unsigned)(k - 1))); // (unsigned: an empty row holds INT_MAX, whose lane aims at the dummy word)
        }
        // ================= joins: UEs whose contention window opens in this subframe enter the histogram and the lowest-index tables =================
        { // (from the last wavefront down: the first ones have the most event batches; the next batch's entries are in flight while this one's atomics are issued)
            const PRACH_G int *const jl = jcal + (size_t)slot * (size_t)calcap;
            const int qs = (NWB - 1 - w) * 64;
            const int dmyj = bl::DUMMY / 4 + lane;
            int je_n = qs + lane < nj ? jl[qs + lane] : 0;
            for (int q0 = qs; q0 < nj; q0 += NWB * 64) {
                const int q = q0 + lane;
                const int je = je_n;
                je_n = q + NWB * 64 < nj ? jl[q + NWB * 64] : 0;
                const int i = je & 0xFFFFF, p = (je >> 20) & 63;
                int dur = (int)((unsigned)je >> 26);
                dur &= ~(lm(q >= nj) | lm(grant_search(grant_bits(p, t - 1, t - 1), i, p, t - 1) >= 0)); // (granted in the subframe it was scheduled in: out of contention, Beta.c:338-343)
                // the histogram of the matched UEs is kept as DIFFERENCES: + 1 where the window opens, - 1 where it has closed (summed up subframe by subframe
                // behind S1) — two atomics per window instead of one per subframe of it; a lane without a window adds to its own dummy word
                const lmask has = lm(dur > 0);
                atomicAdd(&lds[lsel(has, bl::DR / 4 + (t & (HRING - 1)) * NPB + p, dmyj)], 1);
                atomicSub(&lds[lsel(has, bl::DR / 4 + ((t + dur) & (HRING - 1)) * NPB + p, dmyj)], 1);
                // the lowest matched index: of THIS subframe (the ring's row t, as the event body's), and of the subframe the window ends in (ER: the later
                // subframes of the window are served from there, by the seeding at every subframe's head) — no loop over the window's subframes
                atomicMin(&lds[lsel(has, bl::MR / 4 + (t & (HRING - 1)) * NPB + p, dmyj)], (i << 6) | (dur - 1));
                atomicMin(&lds[lsel(has, bl::ER / 4 + ((t + dur) & (HRING - 1)) * NPB + p, dmyj)], i);
            }
        }
        BSTAMP(1); // joins
        BSTAMP(2);

        // ================= the event body: this subframe's event list, one chunk (64 UEs) at a time =================
        {
            int c_succ = 0, c_contf = 0;
            const int tmod = t % aT;
            const CallTables tab{fcallB, lcallB};
            // The loop over the list's batches is written for gfx950's ONE in-order memory counter (loads and stores complete in issue order as far as
            // s_waitcnt vmcnt can tell): at the top of a batch the records of the NEXT batch and the chunk-table entry of the one after are requested; they
            // are waited for once, right before this batch's stores are issued (pipe_sync), and the stores then have the next batch's compute phase to complete.
            int pd_n2 = 0;      // chunk-table entry two batches ahead (wave-uniform value in a vector register)
            BRec pR_n;          // this lane's record in the next batch
            pR_n.a = pR_n.b = make_int4(0, 0, 0, 0);
            auto pipe_sync = [&]() __attribute__((always_inline)) {
                asm volatile("" :: "v"(pR_n.a.x), "v"(pR_n.a.y), "v"(pR_n.a.z), "v"(pR_n.a.w), "v"(pR_n.b.x), "v"(pR_n.b.y), "v"(pR_n.b.z), "v"(pR_n.b.w), "v"(pd_n2));
            };
            auto desc_of = [&](const int b) -> int { return b < nch ? clist[b] : 0; }; // (arrival batches have no chunk)
            auto recs_of = [&](const int d, const int b) -> BRec { // (every lane loads: chunk 0 is always mapped)
                BRec R;
                const PRACH_G v4i_t *const c = chunks + (size_t)(b < nch ? (d & 0xFFFFFF) : 0) * CHUNK;
                R.a = ld_i4(c + lane); R.b = ld_i4(c + 64 + lane);
                return R;
            };
            // What follows selectPreamble / requestResourceAllocation for every event UE: bucket bookkeeping, special events for the resolver, the UE's next
            // schedule.  Straight-line code on lane masks (prach_ue_body.h, "without branches"): a lane that has nothing to add to a table or list aims at a dummy
            // word of its own, in LDS or in the pool's last chunk / behind the join lists in global memory.
            const int dmy = bl::DUMMY / 4 + lane, rowx = (t & (HRING - 1)) * NPB;
            auto finish = [&](const lmask vm, const int i, const UeState &u, const ColdRegs &cold, const unsigned nd, const FlatOut &o) __attribute__((always_inline)) {
                if (PRACH_B_XVALU) { // (sensitivity experiment: never changes a result)
                    int x_ = 0;
#pragma unroll
                    for (int k_ = 0; k_ < PRACH_B_XVALU; k_++) x_ ^= philox_draw31(seed_lo, seed_hi, (unsigned)i, nd + 77u + (unsigned)k_, (unsigned)nUE, (unsigned)variant);
                    if (x_ == 0x7fffffff && i == 0xffffe) scal[B_OVF] = 3; // (a condition the compiler cannot decide — `nd == 0xfffffff` it could: nd has 24 bits, and the draws were gone)
                }
                if (PRACH_B_XSALU) {
                    int z_ = t;
#pragma unroll
                    for (int k_ = 0; k_ < PRACH_B_XSALU; k_++) { asm volatile("s_mul_i32 %0, %0, 3\n\ts_add_i32 %0, %0, 1" : "+s"(z_)); }
                    if (z_ == 0x7fffffff) scal[B_OVF] = 3;
                }
                if (PRACH_B_XBR) {
                    int q_ = i;
#pragma unroll
                    for (int k_ = 0; k_ < PRACH_B_XBR; k_++) { if ((q_ >> (k_ & 15)) & 1) { asm volatile("v_add_u32 %0, %0, 7\n\tv_xor_b32 %0, %0, 5\n\tv_add_u32 %0, %0, 7\n\tv_xor_b32 %0, %0, 5\n\tv_add_u32 %0, %0, 7\n\tv_xor_b32 %0, %0, 5\n\tv_add_u32 %0, %0, 7\n\tv_xor_b32 %0, %0, 5\n\tv_add_u32 %0, %0, 7\n\tv_xor_b32 %0, %0, 5\n\tv_add_u32 %0, %0, 7\n\tv_xor_b32 %0, %0, 5\n\tv_add_u32 %0, %0, 7\n\tv_xor_b32 %0, %0, 5" : "+v"(q_)); } }
                    if (q_ == 0x7ffffff1) scal[B_OVF] = 3;
                }
                if (PRACH_B_XLDS) {
                    int y_ = 0;
#pragma unroll
                    for (int k_ = 0; k_ < PRACH_B_XLDS; k_++) y_ += atomicAdd(&scal[50 + (k_ & 7)], 1);
                    if (y_ == 0x7fffffff) scal[B_OVF] = 3;
                }
                // ---- bucket bookkeeping (this subframe's ring row; an event UE has nothing in the later ones: window left = 0) ----
                const int oldq = o.oldp & 63;
                atomicAdd(&lds[lsel(o.member_pre, bl::HR / 4 + rowx + oldq, dmy)], 1);
                atomicMin(&lds[lsel(lm(u.pend == PEND_STAY), bl::MR / 4 + rowx + oldq, dmy)], i << 6);
                atomicMin(&lds[lsel(lm(o.evtype == UEV_CALLER), bl::MR / 4 + rowx + (o.evp & 63), dmy)], i << 6);
                {
                    // (an early leaver matters only below its bucket's lowest caller, Beta.c:321-330: the lowest matched index only falls during the subframe, so a UE
                    //  at or above what it is NOW — this wavefront's joins are all in, the others' nearly — can be dropped here; the rest is looked at again behind S1)
                    const lmask isev = lm(o.evtype != UEV_NONE), cand = o.eclass & lm(i < (lds[bl::MR / 4 + rowx + oldq] >> 6));
                    if (__any((isev | cand) != 0)) { // (wave-uniform) every lane takes its own slot of the resolver's event list / of the candidate list
                        const int es = atomicAdd(&lds[lsel(isev, bl::SCAL / 4 + B_NEV + parity, dmy)], 1);
                        const int cs = atomicAdd(&lds[lsel(cand, bl::SCAL / 4 + B_NCAND + parity, dmy)], 1);
                        const lmask evl = isev & lm(es < BEV), cdl = cand & lm(cs < CCAP);
                        const int info = flat_event_info(o), cv = i | (o.oldp << 20);
                        BI2(0)[lsel(evl, bl::GEV / 8 + es, bl::DUMMY2 / 8 + lane)] = make_int2(i, info);
                        lds[lsel(cdl, bl::CANDL / 4 + cs, dmy)] = cv;
                        const lmask evg = isev & ~evl, cdg = cand & ~cdl;
                        if (__any((evg | cdg) != 0)) { // the lists' parts in global memory (event storms of extreme parameter sets)
                            if (evg) store_i2(&evov[es - BEV], i, info);
                            if (cdg) ((PRACH_G int *)rare()->qov)[cs - CCAP] = cv;
                        }
                    }
                }
                // the UE's schedule from here (prach_ue_body.h): matched in [tj, tj + dur), its next event at tj + dur
                const unsigned word = o.word;
                const unsigned tjn = word & 0xFFFFu, durn = (word >> 16) & 0x3Fu;
                const lmask sched = vm & lm(tjn != 0xFFFFu); // (0xFFFF: finished for good, or a txTime that never comes — Beta.c:167: the record goes home)
                const int te = (int)(tjn + durn);             // the subframe of its next event
                const lmask joins = sched & lm(durn > 0u);
                const int js = (int)(tjn & calmask);
                const int jp = atomicAdd(&lds[lsel(joins, bl::JCNT / 4 + js, dmy)], 1);
                // where the record goes: its place in this wavefront's open chunk of subframe te, by one atomic add on the chunk's word
                const int old = atomicAdd(&lds[lsel(sched, bl::OPENT / 4 + w * 256 + (te & 255), dmy)], 1);
                int cpos = old & 255, cid = (old >> 8) - 1;
                {
                    // ... and the wave-uniform rest: a subframe without an open chunk (its lanes counted from 0) gets one, a chunk that this batch filled up joins its
                    // subframe's list and the lanes beyond it start the next one — about one chunk per batch
                    unsigned long long bad = __ballot((sched & (lm(cid < 0) | lm(cpos >= 64))) != 0);
                    while (bad) {
                        const int te0 = __builtin_amdgcn_readlane(te, __ffsll((long long)bad) - 1), k = te0 & 255;
                        const int now = __builtin_amdgcn_readfirstlane(otab[k]);
                        const int tot = now & 255, id0p = now >> 8;
                        const lmask mine = sched & lm(te == te0);
                        int idn, left;
                        if (id0p == 0) { idn = chunk_alloc(); left = tot; cid = lsel(mine, idn, cid); }
                        else {
                            chunk_close(id0p - 1, te0, 64);
                            idn = chunk_alloc(); left = tot - 64;
                            const lmask over = mine & lm(cpos >= 64);
                            cid = lsel(over, idn, cid); cpos -= 64 & over;
                        }
                        if (lane == 0) otab[k] = ((idn + 1) << 8) | left;
                        bad &= ~__ballot(mine != 0);
                    }
                }
                pipe_sync(); // the next batch's records and the entry behind them have arrived: from here on only stores are issued
                {
                    // (a lane without a record to pass on writes into the pool's last chunk, which is never handed out: chunk_alloc)
                    PRACH_G v4i_t *const c = chunks + (size_t)(unsigned)lsel(sched, cid * CHUNK + cpos, vtrash + lane);
                    const int4 B = cold_pack(nd, durn, cold, lsel(sched, i, (int)word));
                    st_i4(c, pack(u)); st_i4(c + 64, B);
                    // (an entry beyond the list's capacity lands on its last one; the list's count says so when its subframe comes: the loop's head)
                    vjcal[(size_t)(unsigned)lsel(joins, js * vcalcap + min(jp, vcalcap - 1), vjtrash + lane)] = (int)((unsigned)i | (((word >> 24) & 0x3Fu) << 20) | (durn << 26));
                    const lmask home = vm & ~sched; // finished for good (or never to be looked at again): once per UE
                    if (__any(home != 0)) {
                        if (home) {
                            PRACH_G v4i_t *const h = (PRACH_G v4i_t *)rare()->rec32 + 2 * (size_t)i;
                            st_i4(h, pack(u)); st_i4(h + 1, B);
                        }
                    }
                }
            };
            const lmask wn = withnoma ? -1 : 0, rc_slot = (K.aT > 1 && tmod == 1) ? -1 : 0;
            // MODE 0: Philox, everything in one pass, in the branch-free form.  MODE 1 (GLIBC): the count pass — nothing is stored but the calling lanes.  MODE 2
            // (GLIBC): the full body (branched form) with the draws at their stream positions (sbase: position of the subframe's first call of the UE loop).
            auto body_pass = [&](auto MODE_, const unsigned long long sbase) __attribute__((always_inline)) {
                constexpr int MODE = decltype(MODE_)::value;
                // A wavefront's first three batches are its own (w, w + NWB, w + 2 NWB: the pipeline's depth); every further one is drawn from a counter as the
                // wavefront gets there, three batches ahead — batches differ (rare paths, partial chunks, the joins a wavefront did before), and the barrier behind
                // the body waited a batch's length for the slowest wavefront when they were dealt round-robin.  (Drawing the first three from the counter as well
                // was measured 9 % SLOWER on the grid: an exposed LDS round trip at the head of every subframe's body.)
                int b_c = w, b_n = w + NWB, b_n2 = w + 2 * NWB;
                int d_c = desc_of(b_c), d_n = desc_of(b_n);
                BRec R_c = recs_of(d_c, b_c);
                asm volatile("" :: "v"(R_c.a.x), "v"(R_c.a.y), "v"(R_c.a.z), "v"(R_c.a.w), "v"(R_c.b.x), "v"(R_c.b.y), "v"(R_c.b.z), "v"(R_c.b.w), "v"(d_n)); // (the first batch is waited for HERE, not at every batch's top)
                while (b_c < nb) {
                    const int b = b_c;
                    int tkv = 0; // (lane 0's add returns the ticket; it is read at the END of the batch: the LDS round trip is behind the batch's work)
                    if (b_n2 < nb) tkv = atomicAdd(&lds[lane == 0 ? bl::SCAL / 4 + B_TICK + (MODE == 2 ? 1 : 0) : bl::DUMMY / 4 + lane], 1);
                    if (b_n < nb) { pR_n = recs_of(d_n, b_n); pd_n2 = desc_of(b_n2); } // in flight while this batch is worked on
                    const bool arrival = b >= nch;
                    const int dcu = __builtin_amdgcn_readfirstlane(d_c);
                    BRec R = R_c;
                    if (!arrival && MODE != 1) chunk_free(dcu & 0xFFFFFF); // its records are in registers: the chunk can be filled again (by this wavefront, below)
                    // (the next batch moves into place at the END of this one, behind pipe_sync: a register move of a value still in flight would wait for it here)
                    auto rotate = [&]() __attribute__((always_inline)) {
                        d_c = d_n; d_n = pd_n2; R_c = pR_n;
                        const int b_n3 = b_n2 < nb ? 3 * NWB + __builtin_amdgcn_readfirstlane(tkv) : nb;
                        b_c = b_n; b_n = b_n2; b_n2 = b_n3;
                    };
                    const int ia = prevAC + (b - nch) * 64 + lane; // (arrival batches: Beta.c:136-146 in index order)
                    if constexpr (MODE == 0) {
                        const lmask vm = arrival ? lm(ia < activeCheck) : lm(lane < (dcu >> 24));
                        R.a.w &= vm; // (a lane without a record: inactive, nothing pending — it takes no branch of the state machine and aims at the dummies)
                        unsigned nd = arrival ? (withnoma ? 2u : 0u) : ((unsigned)R.b.x & 0xFFFFFFu);
                        const int i = arrival ? ia : (R.b.w & 0xFFFFF);
                        UeState u = unpack(R.a);
                        ColdRegs cold = cold_unpack(R.b);
                        if (arrival) { // ue_activate (Beta.c:136-146; activateUEs WithNOMA:383-394 also draws twice)
                            u.tx = lsel(vm, t + 1, -1); u.tb = t; u.bo = 0; u.act = ACT_M1 & vm; u.conn = 0; u.pre = 0; u.rar = 0; u.mrc = 0; u.pend = PEND_NONE;
                            cold.ptc = 0; cold.ftt = t + 1; cold.stt = 0; cold.fcnt = 0;
                        } else {
                            // an UL grant noted since the UE was scheduled (Beta.c:338-343) is applied as if the UE had been looked at one subframe after it: nothing
                            // else happens to a granted UE before its Msg3, ten subframes later
                            const int sdur = (int)((unsigned)R.b.x >> 24), p_ = (u.pre - 1) & 63;
                            const unsigned gb = grant_bits(p_, t - sdur - 1, t - 1) & (unsigned)(lm(u.act == ACT_M1) & ~lm(u.pre == 0));
                            const int gs = grant_search(gb, i, p_, t - sdur - 1);
                            const lmask granted = lm(gs >= 0);
                            flat_catch_up(u, sdur, granted, i, t, lsel(granted, gs + 1, t), K.fmA, tab);
                        }
                        const FlatPlan pl = flat_plan(u, t, K.maxRar, K.maxMsg2);
                        int d1 = 0, d2 = 0;
                        if (__any(pl.need != 0)) { // (draws only where a lane needs one, and behind the plan: the shape every digest and fuzzer covers — LABNOTES, round 4)
                            // (the key and the trial's two constant counter words enter Philox through an opaque copy made HERE: left to the compiler, the twenty round
                            //  keys and the first round's products are hoisted out of the step loop and held in 24 vector registers the whole kernel long — registers the
                            //  body does not have: it spilled to scratch, and a scratch load makes the in-order memory counter wait for the next batch's records)
                            unsigned k0_ = seed_lo, k1_ = seed_hi, c2_ = (unsigned)nUE, c3_ = (unsigned)variant;
                            asm volatile("" : "+s"(k0_), "+s"(k1_), "+s"(c2_), "+s"(c3_));
                            if (__any(pl.need > 1)) philox_draw31_x2(k0_, k1_, (unsigned)i, nd, c2_, c3_, d1, d2);
                            else d1 = philox_draw31(k0_, k1_, (unsigned)i, nd, c2_, c3_);
                            nd += (unsigned)pl.need;
                        }
                        FlatOut o = flat_select(u, cold, pl, d1, d2, t, rc_slot, K, wn, c_succ, c_contf);
                        o.word = (unsigned)lsel(vm, (int)flat_schedule(u, t, K.maxRar), (int)PW_IDLE);
                        finish(vm, i, u, cold, nd, o);
                        rotate();
                    } else {
                        bool v = arrival ? ia < activeCheck : lane < (dcu >> 24);
                        if (!v || arrival) { R.a = make_int4(-1, 0, 0, 0); R.b = make_int4(0, 0, 0, 0); }
                        const int i = arrival ? ia : (R.b.w & 0xFFFFF);
                        UeState u = unpack(R.a);
                        ColdRegs cold = cold_unpack(R.b);
                        unsigned nd = (unsigned)R.b.x & 0xFFFFFFu;
                        const int sdur = (int)((unsigned)R.b.x >> 24);
                        int tcu = t;
                        bool granted = false;
                        if (v && !arrival && u.act == ACT_M1 && u.pre != 0) {
                            const int gs = granted_at(i, u.pre - 1, t - sdur - 1, t - 1);
                            if (gs >= 0) { granted = true; tcu = gs + 1; }
                        }
                        if (v && !arrival) pw_catch_up(u, pw_make(t - sdur, sdur, 0), granted, i, tcu, K.fmA, tab);
                        if (v && arrival) {
                            ue_activate(u, i, t, cold);
                            if (withnoma) nd = 2;
                            // the reference's stream: the UE's sector is fixed by the first of its two activation calls (WithNOMA:393-410), which sit at the head of this
                            // subframe's calls in index order (read in the select pass: the window has been checked) — kept per UE for the grant phase (PD->sector)
                            if (sectors && MODE == 2) sect_arr[i] = sector_of_draw(stream[base + 2ull * (unsigned long long)(i - prevAC)]);
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
                        if (v && pl.need > 0) { // the UE's position inside its group from the two lane masks
                            const unsigned lo_ = ln < 32 ? (1u << ln) - 1u : 0xffffffffu, hi_ = ln < 32 ? 0u : (1u << (ln - 32)) - 1u;
                            const int before = __popc(gm[4 * g_] & lo_) + __popc(gm[4 * g_ + 1] & hi_) + __popc(gm[4 * g_ + 2] & lo_) + __popc(gm[4 * g_ + 3] & hi_);
                            const unsigned long long o_ = sbase + (unsigned long long)gpre[g_] + (unsigned long long)before;
                            d1 = stream[o_];
                            if (pl.need > 1) d2 = stream[o_ + 1];
                        }
                        const UeOut uo = ue_select(u, pl, d1, d2, i, t, tmod, K, cold, c_succ, c_contf);
                        FlatOut o;
                        o.evtype = uo.evtype; o.evp = uo.evp; o.evq = uo.evq; o.oldp = uo.oldp; o.member_pre = lm(uo.member_pre); o.eclass = lm(uo.eclass);
                        o.word = v ? pw_schedule(u, t, K.maxRar) : PW_IDLE;
                        finish(lm(v), i, u, cold, nd, o);
                        rotate();
                    }
                }
            };
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
        __syncthreads(); // S1: histogram / lowest callers / candidate list are complete; the caller tables of t - 1 are free; nothing is appended to subframe t + 1's list any more
        BSTAMP(4);

        // early leavers below the bucket's lowest caller are the only ones a rank can need
        {
            chunk_flush(t + 1); // this wavefront's open chunk of the next subframe joins that subframe's list
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
            if (GLIBC) // every draw of this subframe has been read: the marks go (all of them: 2 x 16 bytes per thread)
                for (int k = tid; k < BGG; k += TB) *reinterpret_cast<uint4 *>(&gm[4 * k]) = make_uint4(0u, 0u, 0u, 0u);
            if (tid < NPB) { // this workgroup's histogram / lowest callers ARE the totals (only read here: the filter above reads them too)
                lcallB[tid] = -1; BI(bl::NLV)[tid] = 0; BI(bl::FIE)[tid] = 0;
                const int m_ = mlocx[tid];
                const int run_ = BI(bl::RUN)[tid] + BI(bl::DR)[(t & (HRING - 1)) * NPB + tid]; // the UEs inside their window in this subframe: the differences summed up
                BI(bl::RUN)[tid] = run_; BI(bl::DR)[(t & (HRING - 1)) * NPB + tid] = 0;        // (this row is the subframe t + HRING from here on)
                BI(bl::TOTAL)[tid] = histx[tid] + run_; fcallA[tid] = m_ == INT_MAX ? INT_MAX : (m_ >> 6); BI(bl::FMINP)[tid] = m_;
                // the grant notes of subframe t - 16 make room for this subframe's (no record scheduled that long ago is still on its way)
                rg[(t & 15) * NPB + tid] = -1; bmk[tid] &= ~(1u << (t & 15));
            }
            if (tid == 0) {
                scal[B_EVENTS] += nch * 64 + narr; scal[B_JOINS] += nj; // (reported, never read by the simulation: records read, whole chunks)
                scal[B_NS] = 0; scal[B_NRC] = 0; scal[B_NRJ] = 0; scal[B_NROV + (t & 15)] = 0; scal[B_TICK] = 0; scal[B_TICK + 1] = 0;
                nchk[slot] = 0; jcnt[slot] = 0; // (this slot is the subframe t + calmask + 1 from here on: nothing is scheduled that far ahead)
                if (scal[B_POOLH + (parity ^ 1)] < 0) scal[B_POOLH + (parity ^ 1)] = 0; // (the half that was drawn from in this subframe takes the next subframe's free chunks)
            }
        }
        BSTAMP(5); // leaver filter
        __syncthreads(); // S2
        BSTAMP(6);
        if (scal[B_OVF]) { status = PRACH_ERR_INTERNAL; why = 4 + scal[B_OVF]; time_exit = t; break; } // 5 a join list, 6 the grant notes, 8 a chunk table, 9 the chunk pool: full
        const int N = scal[B_NEV + parity];
        if (tid < NPB) { histx[tid] = 0; mlocx[tid] = INT_MAX; BI(bl::ER)[(t & (HRING - 1)) * NPB + tid] = INT_MAX; } // (these ring rows are the subframe t + HRING from here on; ER's row t held the windows that ended before t)
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
        // An UL grant (Beta.c:338-343) for the singleton caller `my` of bucket bp: NOTED (its record is in some later subframe's list and takes the grant when it
        // comes up); a mid-window UE (the bucket's lowest matched index as the ring held it, with subframes of its window still ahead) leaves the ring's later
        // subframes — it was the bucket's only member, so its contributions there are exactly one count and the lowest index
        auto grant = [&](const int my, const int bp) {
            const int row = t & 15;
            if (atomicCAS(&rg[row * NPB + bp], -1, my) != -1) { // (a further singleton caller of the same preamble in this subframe)
                const int gp = atomicAdd(&scal[B_NROV + row], 1);
                if (gp < ROVCAP) rov[row * ROVCAP + gp] = my | (bp << 20); else scal[B_OVF] = 2;
            }
            atomicOr(&bmk[bp], 1u << row);
            const int fm = BI(bl::FMINP)[bp];
            if ((fm >> 6) == my && fm != INT_MAX) {
                const int rem = fm & 63;
                if (rem > 0) { atomicSub(&BI(bl::DR)[((t + 1) & (HRING - 1)) * NPB + bp], 1); atomicAdd(&BI(bl::DR)[((t + rem + 1) & (HRING - 1)) * NPB + bp], 1); } // out of the histogram from t + 1 on
                if (rem > 0) atomicCAS(&BI(bl::ER)[((t + rem + 1) & (HRING - 1)) * NPB + bp], my, INT_MAX); // ... and out of the lowest-index table of its window (it was alone in it)
            }
        };
        // a caller's sector: Philox — a function of the UE's own first activation draw, recomputed; the reference's stream — kept in the UE's record
        auto sector_of = [&](const int my) -> int {
            if (GLIBC) return sect_arr[my];
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
        if ((Gr > 0 || sectors) && ns > 0) __syncthreads(); // S6: the grant notes are complete before the next subframe's joins and events look for them
        BSTAMP(12);
        if (nsucc_tot == nUE) { time_exit = t; break; } // Beta.c:180
    }
    __syncthreads();
    if (status == PRACH_OK && scal[B_OVF]) { status = PRACH_ERR_INTERNAL; why = 4 + scal[B_OVF]; time_exit = tlast; } // (raised by the last subframe's grants)

    // ---- the state after the last subframe.  Records still on their way (in some later subframe's list, or in a wavefront's open chunk) go home first: the
    // deferred outcome of the last subframe (or the grant noted for them) and the subframes they were matched in since they were scheduled are applied, as
    // the event body would have (pw_catch_up).  Then end-of-trial sums (Beta.c:185-197) and the logged fields (Beta.c:501-508) of every UE from its home record.
    const int tend = tlast + 1;
    if (status == PRACH_OK && tlast >= 0) {
        for (int k = lane; k < 256; k += 64) { // every open chunk of this wavefront joins its subframe's list (position k stands for the subframe in (tlast + 1, tlast + 256] that is k mod 256)
            const int old = otab[k];
            if (old >> 8) chunk_enter((old >> 8) - 1, tend + (int)(((unsigned)k - (unsigned)tend) & 255u), old & 255);
        }
        __syncthreads();
        const CallTables tab{BI(bl::FCALL) + (tlast & 1) * NPB, BI(bl::LCALL) + (tlast & 1) * NPB};
        for (int sl = 0; sl <= (int)calmask; sl++) {
            const int n = min(nchk[sl], tcap);
            const int te = tend + (int)(((unsigned)sl - (unsigned)tend) & calmask); // the subframe this slot stands for
            for (int b = w; b < n; b += NWB) {
                const int d = ctab[(size_t)sl * (size_t)tcap + (size_t)b];
                const PRACH_G v4i_t *const c = chunks + (size_t)(d & 0xFFFFFF) * CHUNK;
                if (lane < (d >> 24)) {
                    const int4 ra = ld_i4(c + lane), rb = ld_i4(c + 64 + lane);
                    UeState u = unpack(ra);
                    const int i = rb.w & 0xFFFFF, sdur = (int)((unsigned)rb.x >> 24);
                    int tcu = tend;
                    bool granted = false;
                    if (u.act == ACT_M1 && u.pre != 0) {
                        const int gs = granted_at(i, u.pre - 1, max(te - sdur - 1, tlast - 14), tlast);
                        if (gs >= 0) { granted = true; tcu = gs + 1; }
                    }
                    pw_catch_up(u, pw_make(te - sdur, sdur, 0), granted, i, tcu, K.fmA, tab);
                    st_i4(rec32 + 2 * (size_t)i, pack(u)); st_i4(rec32 + 2 * (size_t)i + 1, make_int4(rb.x & 0xFFFFFF, rb.y, rb.z, (int)PW_IDLE));
                }
            }
        }
    }
    __syncthreads();
    {
        PRACH_G int *const timers = (PRACH_G int *)PD->timers;
        PRACH_G v4i_t *const logs = (PRACH_G v4i_t *)PD->logs;
        long long sumT = 0;
        int ptcS = 0, fcS = 0;
        unsigned long long ndS = 0;
        for (int i = tid; i < nUE; i += TB) {
            const int4 ra = ld_i4(rec32 + 2 * (size_t)i), rb = ld_i4(rec32 + 2 * (size_t)i + 1);
            const UeState u = unpack(ra);
            const ColdRegs cold = cold_unpack(rb);
            const int timer = u.act == ACT_IDLE ? -1 : (u.act == ACT_DONE ? u.tb : tend - u.tb);
            if (u.act == ACT_DONE) { sumT += timer; ptcS += cold.ptc; fcS += cold.fcnt; }
            ndS += (unsigned)rb.x & 0xFFFFFFu;
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
        o->hard_error = why; // (2 reset-cycle candidates, 3 singleton callers, 4 crossing bin, 5 a join list, 6 the grant notes, 8 a chunk table, 9 the chunk pool: reported by the engine)
        o->time_exit = time_exit;
        o->collisionPreambles = scal[B_COLL]; o->totalPreambleTxop = scal[B_TXOP];
        o->activeCheck = activeCheck;
        o->steps = steps;
        o->visits = (unsigned long long)(unsigned)scal[B_JOINS]; // (join-list entries: one per contention window)
        o->events = (unsigned long long)(unsigned)scal[B_EVENTS]; // (records streamed through the event body, whole chunks)
#ifdef PRACH_STAMPS
        for (int k = 0; k < 24; k++) o->fstamps[k] = fstamps[k];
#endif
    }
}

size_t batch_kernel_lds_bytes(int waves, bool glibc) { return glibc ? (size_t)BLG::END : waves == 8 ? (size_t)BL<8>::END : (size_t)BL<16>::END; }
int batch_max_preambles() { return NPB; }
int batch_max_rar_window() { return 11; }
int batch_max_subframes() { return 65000; }
int batch_max_groups(bool glibc) { return glibc ? BGG : (1 << 14); }
int batch_max_rar_window_two_per_cu() { return 11; }
int batch_chunk_bytes() { return CHUNK * 16; }
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
