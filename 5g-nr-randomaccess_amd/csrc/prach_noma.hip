// prach_noma.hip — NOMA.c's simulation loop (NOMA.c:644-714) on gfx950: the sector power-level grouping.
//
// A CLUSTER of G workgroups per (seed, nUE) trial (G = 1: one workgroup), Philox draws per (UE, draw#).
// Ownership of 64-UE groups is static and interleaved (group g -> workgroup g % G, wave (g / G) % 16), so a
// UE's record never leaves its CU.  Work is organised per 5 ms ACCESS SLOT, not per subframe:
//   (pass A of slot s+1 runs inside pass B of slot s: one sweep over the records per slot)
//   pass A   activation of newly arrived UEs from the activation table (activeUE, NOMA.c:131-192: first preamble,
//            sector, Rayleigh channel gain — built by noma_activation_kernel below, see there) and the
//            transmitter gather of preambleSectorCollisionDetection (NOMA.c:206-212) as a 6 x nPreamble
//            LDS histogram (count + lowest index per bin, LDS atomics);
//   exchange (G > 1) every workgroup publishes its 6 x nPreamble bins as self-validating 8-byte granules
//            (write-through sc1 stores, tag = slot+1) and reads everybody's (relaxed sc1 loads, re-read until
//            the tag matches; bounded) — ONE exchange per slot, none on the other four subframes;
//   resolve  (every workgroup, redundantly) one wavefront per sector: singletons in preamble order (ballot
//            compaction), stable rank-sort by channel gain (the bubble sort of NOMA.c:90-103), greedy pairing
//            of UEs whose gains differ by more than 15 in 10*ln (NOMA.c:268-298: ballot + find-first over the
//            sorted lanes), 2 grants per sector, the pair's decode draws (stateless Philox counter per
//            (slot, sector, grant)), leftovers (NOMA.c:299-307); a grant is one atomicOr into the UE's record,
//            issued by the UE's owner;
//   pass B   msg2Results (NOMA.c:449-498) for every transmitter of the slot, then resourceRequestAllocation
//            (NOMA.c:499-546) for the slot's subframe AND the following accessTime-1 subframes in registers:
//            a UE's record is loaded and stored once per slot.  Timers are stored as bases (NOMA.c:702-706
//            costs no traffic).
// In the simulation kernel doubles are only compared / multiplied (the library is built with -ffp-contract=off: HIP's _rn intrinsics are plain
// operators, which the compiler would otherwise fuse into fma — one rounding where the reference has two).  The activation table's
// cos / sin / log come from the device's math library, which is not the reference's libm to the last bit: every place where such a
// value is rounded to float or compared is checked against an error band (noma_activation_kernel: flagged UEs are recomputed on the
// host; the resolver: NOMA_AMBIGUOUS, the trial is rerun with the host-built table), so the RESULTS are the reference's bit for bit.
#include "prach_device.h"
#include "prach_device_fn.h"
#include "prach_noma_act.h"
#include <limits.h>

#ifdef PRACH_STAMPS
#define NSTAMP(k)                                                                                      \
    do {                                                                                               \
        if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); fstamps[k] += now_ - fprev; fprev = now_; } \
    } while (0)
#else
#define NSTAMP(k) do { } while (0)
#endif

#pragma clang fp contract(off) // (a * b + c stays two roundings, as in the reference built for baseline x86-64)

namespace prach {

namespace {

// packed word of the NOMA record
constexpr unsigned N_RA_BIT = 1u << 2, N_FAIL_BIT = 1u << 3, N_MSG2_BIT = 1u << 4, N_M3W_BIT = 1u << 5, N_PRE_SHIFT = 6,
                   N_RETX_SHIFT = 14, N_RAR5_BIT = 1u << 22;
constexpr int NOMA_VARIANT = 2;
constexpr unsigned NGR_NONE = 0xFFFFFu;
constexpr unsigned NSPIN_LIMIT = 1u << 22;

struct NLds {
    int *cnt;    // [6*nP] this workgroup's transmitters per (sector, preamble)
    int *who;    // [6*nP] lowest transmitter index per bin
    int *tcnt;   // [6*nP] totals over the cluster
    int *twho;   // [6*nP]
    int *sidx;   // [6*64] per sector: singleton UE indices (preamble order, then sorted order)
    double *sg;  // [6*64] their channel gains
    double *slg; // [6*64] ln(gain)
    int *scal;   // [32]
};
enum { N_NSUCC = 0, N_STATUS, N_PTC, N_FC, N_MAXT, N_NSUCCTOT, N_MAXTTOT, N_PAIRD, N_SUMT = 8, N_ND = 10, N_SX = 12, N_AMBIG = 13 };

__device__ __forceinline__ NLds ncarve(char *smem, int nP) {
    NLds L;
    L.sg = reinterpret_cast<double *>(smem);
    L.slg = L.sg + 6 * 64;
    int *ip = reinterpret_cast<int *>(L.slg + 6 * 64);
    L.sidx = ip; ip += 6 * 64;
    L.scal = ip; ip += 32;
    L.cnt = ip; ip += 6 * nP;
    L.who = ip; ip += 6 * nP;
    L.tcnt = ip; ip += 6 * nP;
    L.twho = ip; ip += 6 * nP;
    return L;
}

__device__ __forceinline__ long long nld(const PRACH_G long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void nst(PRACH_G long long *p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... or, for a cluster that has verified that it runs on one XCD (prologue handshake, as in prach_lcluster.hip), resident in that XCD's L2
__device__ __forceinline__ void nstx(const bool same_xcd, PRACH_G long long *p, long long v) {
    if (same_xcd) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long nmk(unsigned lo20, unsigned hi20, unsigned tag) {
    const unsigned w0 = (lo20 & 0xFFFFFu) | ((tag & 0xFFFu) << 20), w1 = (hi20 & 0xFFFFFu) | (((tag >> 12) & 0xFu) << 20);
    return (long long)(((unsigned long long)w1 << 32) | w0);
}
__device__ __forceinline__ bool nok(long long g, unsigned tag) {
    const unsigned w0 = (unsigned)g, w1 = (unsigned)((unsigned long long)g >> 32);
    return (w0 >> 20) == (tag & 0xFFFu) && ((w1 >> 20) & 0xFu) == ((tag >> 12) & 0xFu);
}
__device__ __forceinline__ long long nwait(const PRACH_G long long *p, unsigned tag, int *status_word) {
    long long g = nld(p);
    unsigned spins = 0;
    while (!nok(g, tag)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > NSPIN_LIMIT) { *status_word = PRACH_ERR_TIMEOUT; break; } // peer not resident? the engine reruns with one workgroup per trial
        g = nld(p);
    }
    return g;
}

// ---- activeUE on the device (NOMA.c:131-192): one thread per UE of every trial of the launch; the UE's own Philox counter gives its draws, so UEs are
// independent.  The arithmetic and its error bands: prach_noma_act.h.  Flagged UEs are recomputed on the host with the reference's libm before the
// simulation starts; tests/test_noma.py compares this table with prach_noma_activation_table (host libm) UE by UE.
__global__ __launch_bounds__(256) void noma_activation_kernel(const TrialDev *__restrict__ params, const int ntrials, unsigned *__restrict__ flags) {
    const int T = blockIdx.y;
    if (T >= ntrials) return;
    const TrialDev &P = params[T];
    const int i = blockIdx.x * 256 + threadIdx.x, nUE = P.nUE;
    if (i >= nUE) return;
    unsigned k = 0;
    const ActUe o = noma_active_ue([&]() -> int { return philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k++, (unsigned)nUE, NOMA_VARIANT); }, P.nP, P.cell_radius);
    const_cast<int *>(P.n_pre0)[i] = o.pre;
    const_cast<int *>(P.n_sector)[i] = o.sec;
    const_cast<double *>(P.n_gain)[i] = o.gain;
    const_cast<double *>(P.n_lgain)[i] = o.lgain;
    const_cast<unsigned *>(P.n_nd0)[i] = k;
    if (o.flag) {
        const unsigned q = atomicAdd(&flags[0], 1u);
        if (q < (unsigned)NOMA_ACT_FLAG_CAP) { flags[2 + 2 * q] = (unsigned)T; flags[3 + 2 * q] = (unsigned)i; }
    }
}

} // namespace

size_t noma_kernel_lds_bytes(int nP) { return sizeof(double) * 2 * 6 * 64 + sizeof(int) * (6 * 64 + 32 + 4 * 6 * nP); }

__global__ __launch_bounds__(WG_THREADS) void noma_kernel(const TrialDev *__restrict__ params, const int G, const int nT, const int xpack) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int T = blockIdx.x / G, b = blockIdx.x % G; // a cluster = consecutive blocks (in-order dispatch completes whole clusters)
    if (xpack) { // XCD-packed launch (prach_lcluster.hip): a cluster = the blocks of equal blockIdx % 8 of a chunk of 8 G blocks
        const int chunk = blockIdx.x / (8 * G), within = blockIdx.x % (8 * G);
        T = chunk * 8 + (within & 7); b = within >> 3;
        if (T >= nT) return;
    }
    const TrialG P(params[T]);
    const NLds L = ncarve(smem, P.nP);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nUE = P.nUE, nP = P.nP, aT = P.aT, nGrantUL = P.nGrantUL, nb = 6 * P.nP;
    const PRACH_G int *pre0 = P.n_pre0, *sector = P.n_sector;
    const PRACH_G double *gain = P.n_gain, *lgain = P.n_lgain;
    const FastMod fmP = make_fastmod(nP), fmB = make_fastmod(P.backoff), fmA = make_fastmod(aT);
    const int totgroups = (nUE + 63) >> 6;
    const int lgroups = (totgroups + G - 1) / G;
    const int mbstride = 1 + nb; // granules per mailbox: header + bins
    PRACH_G long long *const mbox = reinterpret_cast<PRACH_G long long *>(P.mbox);

    for (int x = tid; x < lgroups * 64; x += WG_THREADS) { // calloc + initUserInfo (NOMA.c:651-655), own groups
        const int g = b + G * (x >> 6), i = g * 64 + (x & 63);
        if (g < totgroups && i < nUE) {
            store_rec(&P.rec[i], make_int4(0, 0, 0, 0));
            P.ptc[i] = 0; P.ftt[i] = 0; P.stt[i] = 0; P.fcnt[i] = 0; P.nd[i] = 0;
        }
    }
    if (tid < 32) L.scal[tid] = 0;
    __syncthreads();
    if (tid == 0) L.scal[N_MAXT] = -1;
    __syncthreads();
    bool sx = false;
    if (xpack && G > 1 && G <= 64) { // same-XCD handshake: XCC ids through write-through granules (tag 0xFFFF, header of the parity-1 mailbox)
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xfu;
        PRACH_G long long *const hs = mbox + (size_t)G * mbstride;
        if (tid == 0) nst(hs + (size_t)b * mbstride, nmk(xcc, 0u, 0xFFFFu));
        if (tid < 64) {
            bool same = true;
            if (tid < G) same = ((unsigned)nwait(hs + (size_t)tid * mbstride, 0xFFFFu, &L.scal[N_STATUS]) & 0xFFFFFu) == xcc;
            const bool all = __ballot(!same) == 0ull;
            if (tid == 0) L.scal[N_SX] = all ? 1 : 0;
        }
        __syncthreads();
        sx = L.scal[N_SX] != 0 && L.scal[N_STATUS] == PRACH_OK;
    }

    int activeCheck = 0, time_exit = P.stop, status = PRACH_OK;
    bool all_done = false;
    // msg2Results SETS rarWindow to the literal 5 and then compares it with maxRarWindow (NOMA.c:453-455): with
    // maxRarWindow > 5 a failed transmitter is never rescheduled (txTime += 3 and rarWindow = 5 is all that happens)
    const bool rar_expires = 5 >= P.maxRarWindow;
    // PRACH_FLAG_NOMA_NONSECTOR: the cell-wide grouping preambleCollisionDetection (NOMA.c:325-447; its call at NOMA.c:688 is commented
    // out in the reference): ONE group with ONE budget of nGrantUL per access slot instead of six; a pair whose decode draw says
    // "one of the two" always decodes the weaker UE, without the second draw (NOMA.c:413-415 vs :287-290)
    const bool nonsector = (P.flags & PRACH_FLAG_NOMA_NONSECTOR) != 0;
    const int nsect = nonsector ? 1 : 6;
    const bool devact = P.n_devact != 0;

    // pass A for one UE of the slot whose subframe is tA: activation of the newly arrived (activeUE, NOMA.c:131-140: everything
    // else comes from the activation table) and the transmitter gather (NOMA.c:207: RA==0, txTime==time+1, msg2==0,
    // nowBackoff<=0, RaFailed==0) into this workgroup's (sector, preamble) bins
    // (pre0v / nd0v / secv: the UE's activation-table entries, loaded by the caller TOGETHER with the record: they do not depend on it, and a busy group's pass is a
    //  chain of global round trips otherwise — record, draw index, sector, one after the other)
    auto pass_a_lane = [&](const int i, int4 &r, bool &dirty, const int tA, const int prevA, const int acA, const int pre0v, const unsigned nd0v, const int secv) {
        if (i >= prevA && i < acA) {
            r.x = tA + 1; r.y = tA; r.z = 0;
            r.w = 1 | (pre0v << N_PRE_SHIFT);
            P.ptc[i] = 1; P.ftt[i] = tA + 1; P.nd[i] = nd0v;
            dirty = true;
        }
        const unsigned pk = (unsigned)r.w;
        if (i < acA && (pk & 3) == 1 && !(pk & (N_RA_BIT | N_FAIL_BIT | N_MSG2_BIT)) && r.x == tA + 1 && now_backoff(r.z, tA) <= 0) {
            const int bin = (nonsector ? 0 : secv) * nP + (int)((pk >> N_PRE_SHIFT) & 0xff);
            atomicAdd(&L.cnt[bin], 1);
            atomicMin(&L.who[bin], i);
        }
    };
    for (int k = tid; k < nb; k += WG_THREADS) { L.cnt[k] = 0; L.who[k] = INT_MAX; L.tcnt[k] = 0; L.twho[k] = INT_MAX; }
    __syncthreads();
    if (P.stop > 0) { // slot 0: its pass A on its own; every later slot's pass A rides on the previous slot's pass B (one sweep)
        const int ac0 = P.sched[0];
        for (int j = w;; j += NW) {
            const int g = b + G * j;
            if (g >= (ac0 + 63) >> 6) break;
            const int i = g * 64 + lane;
            int4 r = make_int4(0, 0, 0, 0);
            bool dirty = false;
            const bool in0 = i < ac0;
            pass_a_lane(i, r, dirty, 0, 0, ac0, in0 ? pre0[i] : 0, in0 ? P.n_nd0[i] : 0u, in0 && !nonsector ? sector[i] : 0); // (nothing is active before slot 0: no record to load)
            if (dirty) store_rec(&P.rec[i], r);
        }
    }
    __syncthreads();

#ifdef PRACH_STAMPS // (make DIAG=1; PRACH_PRINT_STAMPS=1 prints them per SUBFRAME: x accessTime = per slot; workgroup 0, wavefront 0)
    unsigned long long fstamps[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, fprev = __builtin_readcyclecounter();
#endif
    unsigned long long dead0 = 0ull, dead1 = 0ull; // this wavefront's groups (in the order it visits them) that pass B no longer has to look at
    for (int s = 0, t0 = 0; t0 < P.stop && status == PRACH_OK; s++, t0 += aT) {
        const int t = t0; // the slot's subframe (time % accessTime == 0)
        NSTAMP(0); // loop head
        activeCheck = P.sched[s]; // NOMA.c:675-681 (clamped running sum == the arrival table)
        const int acNext = t0 + aT < P.stop ? P.sched[s + 1] : activeCheck; // the next slot's arrivals (none after the last slot)
        const unsigned tag = (unsigned)(s + 1);
        const int ngroups = (max(activeCheck, acNext) + 63) >> 6;
        // ---- exchange: totals over the cluster ----
        int nsucc_tot, maxt_tot;
        if (G == 1) {
            for (int k = tid; k < nb; k += WG_THREADS) { L.tcnt[k] = L.cnt[k]; L.twho[k] = L.who[k]; L.cnt[k] = 0; L.who[k] = INT_MAX; }
            nsucc_tot = L.scal[N_NSUCC]; maxt_tot = L.scal[N_MAXT];
            __syncthreads();
        } else {
            PRACH_G long long *const mygr = mbox + ((size_t)(s & 1) * G + b) * mbstride;
            for (int k = tid; k < nb; k += WG_THREADS) {
                const int wv = L.who[k];
                nstx(sx, mygr + 1 + k, nmk((unsigned)L.cnt[k], wv == INT_MAX ? NGR_NONE : (unsigned)wv, tag));
                L.cnt[k] = 0; L.who[k] = INT_MAX; // the next slot's gather starts in this slot's pass B
            }
            if (tid == 0) nstx(sx, mygr, nmk((unsigned)L.scal[N_NSUCC], (unsigned)(L.scal[N_MAXT] + 1), tag));
            for (int k0 = tid; k0 < G * nb; k0 += 4 * WG_THREADS) { // four granule loads in flight per thread
                long long gv[4];
                int kk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    kk[u] = k0 + u * WG_THREADS;
                    if (kk[u] < G * nb) { const int wg = kk[u] / nb, bin = kk[u] - wg * nb; gv[u] = nld(mbox + ((size_t)(s & 1) * G + wg) * mbstride + 1 + bin); }
                    else gv[u] = 0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (kk[u] >= G * nb) continue;
                    const int wg = kk[u] / nb, bin = kk[u] - wg * nb;
                    if (!nok(gv[u], tag)) gv[u] = nwait(mbox + ((size_t)(s & 1) * G + wg) * mbstride + 1 + bin, tag, &L.scal[N_STATUS]);
                    const unsigned c = (unsigned)gv[u] & 0xFFFFFu, wv = (unsigned)((unsigned long long)gv[u] >> 32) & 0xFFFFFu;
                    if (c) atomicAdd(&L.tcnt[bin], (int)c);
                    if (wv != NGR_NONE) atomicMin(&L.twho[bin], (int)wv);
                }
            }
            NSTAMP(1); // publish + bins gathered
            if (tid >= WG_THREADS - 64) { // headers: successes so far, latest success subframe
                const int l = tid - (WG_THREADS - 64);
                int ns = 0, mt = -1;
                if (l < G) {
                    const long long g_ = nwait(mbox + ((size_t)(s & 1) * G + l) * mbstride, tag, &L.scal[N_STATUS]);
                    ns = (int)((unsigned)g_ & 0xFFFFFu);
                    mt = (int)((unsigned)((unsigned long long)g_ >> 32) & 0xFFFFFu) - 1;
                }
                ns = wave_sum(ns); mt = wave_max(mt); // (DPP: prach_device_fn.h; the whole wavefront is here)
                if (l == 0) { L.scal[N_NSUCCTOT] = ns; L.scal[N_MAXTTOT] = mt; }
            }
            __syncthreads();
            NSTAMP(2); // barrier behind the gather
            if (L.scal[N_STATUS] != PRACH_OK) { status = L.scal[N_STATUS]; break; }
            nsucc_tot = L.scal[N_NSUCCTOT]; maxt_tot = L.scal[N_MAXTTOT];
        }
        if (nsucc_tot == nUE) { time_exit = maxt_tot; all_done = true; break; } // NOMA.c:707-710: `time` of the last success

        // ---- resolve: one wavefront per sector (NOMA.c:214-309), identical on every workgroup ----
        if (w < nsect) {
            const int sct = w;
            const bool single = lane < nP && L.tcnt[sct * nP + lane] == 1;
            const int myidx = single ? L.twho[sct * nP + lane] : -1;
            const unsigned long long sm = __ballot(single);
            const int count = __popcll(sm);
            if (count > 0) {
                const int pos = __popcll(sm & lanemask_lt(lane));
                if (count <= nGrantUL) { // NOMA.c:252-260
                    if (single && ((myidx >> 6) % G) == b) grant_rec(&P.rec[myidx]);
                } else {
                    if (single) { L.sidx[sct * 64 + pos] = myidx; L.sg[sct * 64 + pos] = gain[myidx]; L.slg[sct * 64 + pos] = lgain[myidx]; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    NSTAMP(8); // resolve: gains loaded
                    // stable ascending rank by gain == the bubble sort with strict < (NOMA.c:90-103)
                    int uidx = -1;
                    double ug = 0, ulg = 0;
                    int rank = 0;
                    bool ambiguous = false;
                    if (lane < count) {
                        uidx = L.sidx[sct * 64 + lane]; ug = L.sg[sct * 64 + lane]; ulg = L.slg[sct * 64 + lane];
#pragma unroll 8
                        for (int j = 0; j < count; j++) { // (unrolled: eight broadcast reads in flight instead of one LDS round trip per comparison)
                            const double gj = L.sg[sct * 64 + j];
                            rank += (gj < ug || (gj == ug && j < lane)) ? 1 : 0;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane < count) { L.sidx[sct * 64 + rank] = uidx; L.slg[sct * 64 + rank] = ulg; L.sg[sct * 64 + rank] = ug; } // (every lane has read the unsorted gains: barrier above)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    int cidx = -1;
                    double clg = 0;
                    if (lane < count) { cidx = L.sidx[sct * 64 + lane]; clg = L.slg[sct * 64 + lane]; }
                    // device-built table: two gains closer than the error band of the device's libm (ACT_GAIN_ORDER_BAND, prach_noma_act.h: twice the asserted per-gain error, with margin) could be ordered the other way by the reference's —
                    // in sorted order it is enough to look at neighbours (n_devact == 2: test hook, every sort counts as ambiguous)
                    if (devact && lane + 1 < count) { const double ga = L.sg[sct * 64 + lane], gb = L.sg[sct * 64 + lane + 1]; if (__dsub_rn(gb, ga) <= ACT_GAIN_ORDER_BAND * gb || P.n_devact == 2) ambiguous = true; }
                    NSTAMP(9); // resolve: ranked and sorted
                    unsigned long long valid = count >= 64 ? ~0ull : ((1ull << count) - 1ull);
                    const double clg10 = __dmul_rn(10.0, clg); // (NOMA.c:272: 10 * log(gain), the same product on either side of the difference)
                    int grants = 0, npd = 0;
                    bool grantme = false;
                    const unsigned long long lows = count >= 2 ? ((1ull << (count - 1)) - 1ull) : 0ull; // i < count - 1
                    unsigned long long above = ~0ull;                                                    // bits behind the last i looked at
                    for (;;) { // NOMA.c:268-298, over the still unpaired i in ascending order
                        const unsigned long long rest = valid & lows & above;
                        if (!rest) break;
                        const int i = __ffsll((long long)rest) - 1;
                        above = ~((2ull << i) - 1ull);
                        // (lane i's 10 ln g through two v_readlane — i is wave-uniform — instead of a ds_bpermute round trip per comparison: this loop is a latency chain)
                        const double lgi10 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(clg10), i), __builtin_amdgcn_readlane(__double2loint(clg10), i));
                        const double diff = __dsub_rn(clg10, lgi10); // 10*log(high) - 10*log(low)
                        if (devact && lane < count && fabs(__dsub_rn(diff, 15.0)) < 1e-9) ambiguous = true; // (|error| of 10 ln g1 - 10 ln g2 < 1e-12)
                        // (the lanes still unpaired, not lane 0, not i itself: scalar mask algebra on the comparison's ballot instead of a 64-bit shift per lane)
                        const unsigned long long mj = __ballot(diff > 15.0) & valid & ~(1ull << i) & ~1ull;
                        if (!mj) continue;
                        const int j = __ffsll((long long)mj) - 1;
                        valid &= ~((1ull << i) | (1ull << j));
                        if (grants < nGrantUL) {
                            const unsigned kd = (unsigned)(((s * 6 + sct) * nGrantUL + grants) * 2);
                            grants++;
                            const int d1 = philox_draw31(P.seed_lo, P.seed_hi, 0xFFFFFFFFu, kd, (unsigned)nUE, NOMA_VARIANT);
                            int decoded = 2; // both
                            npd++;
                            if (d1 <= 644245094) { // (double)rand()/RAND_MAX < 0.3 (NOMA.c:284-285)
                                if (nonsector) decoded = 0; // rx[0], the weaker UE: NOMA.c:413-415
                                else {
                                    const int d2 = philox_draw31(P.seed_lo, P.seed_hi, 0xFFFFFFFFu, kd + 1, (unsigned)nUE, NOMA_VARIANT);
                                    decoded = d2 % 2; // index into rx[] = {low, high}
                                    npd++;
                                }
                            }
                            if ((lane == i && (decoded == 2 || decoded == 0)) || (lane == j && (decoded == 2 || decoded == 1))) grantme = true;
                        }
                    }
                    { // leftovers in sorted order while grants remain (NOMA.c:299-307)
                        const bool left = lane < count && ((valid >> lane) & 1ull);
                        const unsigned long long lm = __ballot(left);
                        if (left && __popcll(lm & lanemask_lt(lane)) < nGrantUL - grants) grantme = true;
                    }
                    NSTAMP(10); // resolve: paired
                    if (grantme && ((cidx >> 6) % G) == b) grant_rec(&P.rec[cidx]);
                    if (b == 0 && lane == 0 && npd) atomicAdd(&L.scal[N_PAIRD], npd);
                    if (__any(ambiguous) && lane == 0) L.scal[N_AMBIG] = 1;
                }
            }
        }
        NSTAMP(3); // resolve (sector 0), rest
        __syncthreads();
        NSTAMP(4); // barrier behind the resolve
        // ---- pass B: msg2Results for the slot's transmitters, then resourceRequestAllocation for the slot's
        //      subframe and the accessTime-1 following ones, all on registers (one load / store per slot) ----
        {
            const int tend_slot = min(t0 + aT, P.stop);
            const bool has_next = t0 + aT < P.stop;
            int c_succ = 0, c_maxt = -1;
            for (int k = tid; k < nb; k += WG_THREADS) { L.tcnt[k] = 0; L.twho[k] = INT_MAX; } // (the resolver is done with them)
            for (int j = w, jl = 0;; j += NW, jl++) {
                const int g = b + G * j;
                if (g >= ngroups) break;
                if (jl < 128 && ((jl < 64 ? dead0 >> jl : dead1 >> (jl - 64)) & 1ull)) continue; // nothing can happen to any UE of this group any more (below): not even loaded
                const int i = g * 64 + lane;
                int4 r = make_int4(0, 0, 0, 0);
                unsigned k = 0, nd0v = 0;
                int pre0v = 0, secv = 0;
                if (i < activeCheck) { r = load_rec(&P.rec[i]); k = P.nd[i]; } // (the draw index rides along: only this wavefront ever writes it)
                if (i < acNext && !nonsector) secv = sector[i];
                if (i >= activeCheck && i < acNext) { pre0v = pre0[i]; nd0v = P.n_nd0[i]; }
                unsigned pk = (unsigned)r.w;
                bool alive = i < activeCheck && !(pk & (N_RA_BIT | N_FAIL_BIT));
                if (!__any(alive || (i >= activeCheck && i < acNext))) { if (__all(i < activeCheck || i >= nUE)) { if (jl < 64) dead0 |= 1ull << jl; else if (jl < 128) dead1 |= 1ull << (jl - 64); } continue; }
                bool dirty = false, nd_loaded = false; // (nd_loaded: the draw index moved and has to be stored)
                // -- msg2Results (NOMA.c:692-696 -> :449-498) --
                const bool tx = alive && (pk & 3) == 1 && r.x == t + 1 && now_backoff(r.z, t) <= 0;
                if (__any(tx)) {
                    const bool granted = tx && (pk & PK_GRANT_BIT);
                    const bool txfail = tx && !granted && !(pk & N_MSG2_BIT);
                    int retx = (int)((pk >> N_RETX_SHIFT) & 0xff);
                    const bool perm = txfail && rar_expires && retx + 1 >= P.maxMsg2; // msg1ReTx reaches maxMsg1ReTx: dropped for good
                    int d1 = 0, d2 = 0;
                    if (rar_expires && __any(txfail)) {
                        if (txfail) nd_loaded = true;
                        d1 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k, (unsigned)nUE, NOMA_VARIANT);
                        if (__any(perm)) d2 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k + 1, (unsigned)nUE, NOMA_VARIANT);
                        if (txfail) k += perm ? 2u : 1u;
                    }
                    if (granted) { // NOMA.c:491-497
                        pk = (pk & ~(3u | N_M3W_BIT | PK_GRANT_BIT)) | 2u | N_MSG2_BIT;
                        r.x += 10; P.stt[i] = r.x;
                        dirty = true;
                    } else if (txfail && !rar_expires) { // NOMA.c:452-455 with the window test false
                        r.x += 3;
                        pk |= N_RAR5_BIT;
                        dirty = true;
                    } else if (txfail) { // NOMA.c:452-489
                        r.x += 3;
                        retx++;
                        const int tmp = fastmod(d1, fmB);
                        r.x = slot_align_fm(r.x + tmp, fmA);
                        const int X = r.x - (t + 1) - 1;
                        P.stt[i] = r.x;
                        if (perm) {
                            const int np = fastmod(d2, fmP);
                            pk = (pk & ~((0xffu << N_PRE_SHIFT) | (0xffu << N_RETX_SHIFT))) | ((unsigned)np << N_PRE_SHIFT) | N_FAIL_BIT;
                            P.ptc[i] = 0;
                            r.z = X; // frozen: a dropped UE's counters no longer tick (NOMA.c:703)
                            r.y = 0;
                            alive = false;
                        } else {
                            pk = (pk & ~(0xffu << N_RETX_SHIFT)) | ((unsigned)retx << N_RETX_SHIFT);
                            gadd(&P.ptc[i], 1);
                            r.z = enc_backoff(X, t);
                        }
                        dirty = true;
                    }
                }
                // -- resourceRequestAllocation, subframes t0 .. tend_slot-1 (NOMA.c:699 -> :499-546) --
                // (every lane takes its OWN subframe tt = txTime of the slot in one step — UEs are independent — instead of a loop over the slot's subframes with wave-wide
                //  draws in each: a UE has one such subframe per slot unless accessTime exceeds the 49 subframes a failed Msg3 waits, hence the loop around it)
                for (;;) {
                    const int tt = r.x;
                    const bool m3 = alive && (pk & 3) == 2 && (pk & N_MSG2_BIT) && tt >= t0 && tt < tend_slot;
                    if (!__any(m3)) break;
                    {
                    const bool m3first = m3 && !(pk & N_M3W_BIT), m3to = m3 && (pk & N_M3W_BIT);
                    if (m3) nd_loaded = true;
                    const int d1 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k, (unsigned)nUE, NOMA_VARIANT);
                    int d2 = 0;
                    if (__any(m3to)) d2 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k + 1, (unsigned)nUE, NOMA_VARIANT);
                    if (m3first) { // NOMA.c:502-512
                        k += 1;
                        const float pf = (float)d1 / (float)2147483647;
                        if ((double)pf > 0.1) {
                            const int nbo = now_backoff(r.z, tt);
                            pk = (pk & ~3u) | N_RA_BIT;
                            r.y = (tt - r.y) + 6; r.z = nbo;
                            alive = false;
                            c_succ++; c_maxt = max(c_maxt, tt);
                        } else { r.x += 49; pk |= N_M3W_BIT; }
                        dirty = true;
                    } else if (m3to) { // NOMA.c:514-543
                        k += 2;
                        gadd(&P.fcnt[i], 1); // msg3Faile
                        const int np = fastmod(d1, fmP);
                        const int tmp = fastmod(d2, fmB);
                        r.x = slot_align_fm(r.x + tmp, fmA);
                        P.stt[i] = r.x;
                        r.z = enc_backoff(r.x - tt - 1, tt);
                        pk = (pk & ~(3u | N_MSG2_BIT | N_RAR5_BIT | (0xffu << N_PRE_SHIFT) | (0xffu << N_RETX_SHIFT))) | 1u | ((unsigned)np << N_PRE_SHIFT);
                        P.ptc[i] = 0;
                        r.y = tt;
                        dirty = true;
                    }
                }
                }
                if (nd_loaded) P.nd[i] = k;
                r.w = (int)pk;
                // -- the NEXT slot's pass A on the record as it stands now: one load and one store per UE and slot --
                if (has_next) pass_a_lane(i, r, dirty, t0 + aT, activeCheck, acNext, pre0v, nd0v, secv);
                if (dirty) store_rec(&P.rec[i], r);
                {
                    // A UE acts only in a subframe its txTime names (msg2Results: txTime == slot + 1, NOMA.c:449-455; resourceRequestAllocation: txTime == subframe,
                    // NOMA.c:499-502), and only acting changes txTime: one whose txTime lies before the next slot's (it was in backoff at its last slot, ~1 % of the UEs)
                    // only ticks its timer from here on — and timers are stored as bases.  A group of which every UE has arrived and is finished, dropped or such a
                    // sleeper is never loaded again (bit-exact: its records never change).
                    const unsigned st = pk & 3u;
                    const bool over = i >= nUE || (i < activeCheck && ((pk & (N_RA_BIT | N_FAIL_BIT)) != 0u || (st == 1u && r.x < t0 + aT + 1) || (st == 2u && r.x < t0 + aT)));
                    if (has_next && __all(over)) { if (jl < 64) dead0 |= 1ull << jl; else if (jl < 128) dead1 |= 1ull << (jl - 64); }
                }
            }
            c_succ = wave_sum(c_succ); c_maxt = wave_max(c_maxt);
            if (lane == 0 && c_succ) { atomicAdd(&L.scal[N_NSUCC], c_succ); atomicMax(&L.scal[N_MAXT], c_maxt); }
        }
        NSTAMP(5); // pass B + next pass A
        __syncthreads();
        NSTAMP(6); // barrier behind the pass
    }
    __syncthreads();

    // saveResult (NOMA.c:618-625) + per-UE dump (owned UEs)
    const int tend = all_done ? time_exit + 1 : P.stop; // executed subframes
    long long sumT = 0;
    int ptcS = 0, fcS = 0;
    unsigned long long ndS = 0;
    for (int x = tid; x < lgroups * 64; x += WG_THREADS) {
        const int g = b + G * (x >> 6), i = g * 64 + (x & 63);
        if (g >= totgroups || i >= nUE) continue;
        const int4 r = load_rec_plain(&P.rec[i]);
        const unsigned pk = (unsigned)r.w;
        const int act = pk & 3;
        const bool ra = pk & N_RA_BIT, fail = pk & N_FAIL_BIT;
        const int timer = ra ? r.y : (fail ? 0 : (act > 0 ? tend - r.y : 0));
        const int ntx = P.ptc[i];
        if (ra) { sumT += timer; ptcS += ntx; }
        if (fail) fcS++;
        ndS += P.nd[i];
        P.timers[i] = ra ? timer : INT_MIN;
        if (P.logs) {
            prach_ue_log o;
            o.idx = i; o.timer = timer; o.active = act; o.txTime = r.x; o.firstTxTime = P.ftt[i]; o.secondTxTime = P.stt[i];
            o.nowBackoff = (ra || fail) ? r.z : now_backoff(r.z, tend);
            o.preamble = (int)((pk >> N_PRE_SHIFT) & 0xff);
            o.preambleChange = i < activeCheck ? sector[i] : -1;         // NOMA: sector
            o.rarWindow = (pk & N_RAR5_BIT) ? 5 : 0;                      // NOMA.c:453 / :457,485,539
            o.maxRarCounter = (int)((pk >> N_RETX_SHIFT) & 0xff);        // NOMA: msg1ReTx
            o.preambleTxCounter = ntx;                                    // NOMA: nTxPreamble
            o.msg2Flag = (pk & N_MSG2_BIT) ? 1 : 0;                       // NOMA: msg2
            o.connectionRequest = (pk & N_M3W_BIT) ? 49 : 0;              // NOMA: msg3Wait
            o.msg4Flag = ra ? 1 : 0;                                      // NOMA: RA
            o.failCount = (fail ? 1 : 0) | (P.fcnt[i] << 16);             // NOMA: RaFailed | msg3Faile << 16
            store_log(P.logs, i, o);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sumT += __shfl_down(sumT, d); ptcS += __shfl_down(ptcS, d); fcS += __shfl_down(fcS, d); ndS += __shfl_down(ndS, d); }
    if (lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&L.scal[N_SUMT]), (unsigned long long)sumT);
        atomicAdd(reinterpret_cast<unsigned long long *>(&L.scal[N_ND]), ndS);
        atomicAdd(&L.scal[N_PTC], ptcS);
        atomicAdd(&L.scal[N_FC], fcS);
    }
    __syncthreads();
    if (tid == 0) { // DevResult was zeroed by the engine before the launch
        PRACH_G DevResult *o = P.out;
        gadd(&o->sumTimer, *reinterpret_cast<long long *>(&L.scal[N_SUMT]));
        gadd(&o->draws, *reinterpret_cast<unsigned long long *>(&L.scal[N_ND]) + (b == 0 ? (unsigned long long)L.scal[N_PAIRD] : 0ull));
        gadd(&o->ptcSum, L.scal[N_PTC]);
        gadd(&o->fcSum, L.scal[N_FC]);
        gadd(&o->nSuccess, L.scal[N_NSUCC]);
        gadd(&o->finalSuccess, L.scal[N_NSUCC]);
        if (status != PRACH_OK) gmin(&o->status, status);
        if (L.scal[N_AMBIG]) o->hard_error = NOMA_AMBIGUOUS; // (every workgroup resolves identically: they all store the same value)
        gmax(&o->dbg[0], (unsigned long long)(L.scal[N_MAXT] + 1)); // latest success subframe + 1 (host: exit time when all succeeded)
        if (b == 0) {
            o->time_exit = time_exit;
            o->activeCheck = activeCheck;
            o->steps = (unsigned long long)tend;
#ifdef PRACH_STAMPS
            for (int k = 0; k < 24; k++) o->fstamps[k] = fstamps[k];
#endif
        }
    }
}

hipError_t launch_noma_kernel(const TrialDev *params, int ntrials, int G, int maxP, int xpack, hipStream_t stream) {
    const size_t lds = noma_kernel_lds_bytes(maxP);
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&noma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return rc;
    if (G <= 1) xpack = 0;
    const int grid = xpack ? ((ntrials + 7) / 8) * 8 * G : ntrials * G;
    hipLaunchKernelGGL(noma_kernel, dim3(grid), dim3(WG_THREADS), lds, stream, params, G, ntrials, xpack);
    return hipGetLastError();
}

int noma_kernel_blocks_per_cu(int maxP) {
    const size_t lds = noma_kernel_lds_bytes(maxP);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&noma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&noma_kernel), WG_THREADS, lds) != hipSuccess || nb < 1) return 1;
    return nb;
}

hipError_t launch_noma_activation(const TrialDev *params, int ntrials, int maxUE, unsigned *flags, hipStream_t stream) {
    if (ntrials <= 0 || maxUE <= 0) return hipSuccess;
    if (ntrials > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(noma_activation_kernel, dim3((unsigned)((maxUE + 255) / 256), (unsigned)ntrials), dim3(256), 0, stream, params, ntrials, flags);
    return hipGetLastError();
}

} // namespace prach
