// prach_noma.hip — NOMA.c's simulation loop (NOMA.c:644-714) on gfx950: the sector power-level grouping.
//
// One 1024-thread workgroup per (seed, nUE) trial; Philox draws per (UE, draw#).  Per 5 ms access slot:
//   pass A   activation of newly arrived UEs from the host-built activation table (activeUE, NOMA.c:131-192:
//            first preamble, sector, Rayleigh channel gain — prach_noma_activation_table) and the
//            transmitter gather of preambleSectorCollisionDetection (NOMA.c:206-212) as a 6 x nPreamble
//            LDS histogram (count + lowest index per bin, LDS atomics);
//   resolve  one wavefront per sector: singletons in preamble order (ballot compaction), stable rank-sort
//            by channel gain (the bubble sort of NOMA.c:90-103), greedy pairing of UEs whose gains differ by
//            more than 15 in 10*ln (NOMA.c:268-298: ballot + find-first over the sorted lanes), 2 grants
//            per sector, the pair's decode draws, leftovers (NOMA.c:299-307); a grant is one atomicOr
//            into the UE's record;
//   pass B   msg2Results (NOMA.c:449-498) for every transmitter.
// Every ms: resourceRequestAllocation (NOMA.c:499-546); timers are stored as bases (NOMA.c:702-706 costs
// no traffic).  Same 16-byte hot record idea as the other kernels; doubles are only compared/multiplied
// on the device (explicit _rn intrinsics, no contraction), never produced by transcendental functions,
// so results are bit-identical to the reference's libm-based run.
#include "prach_device.h"
#include "prach_device_fn.h"
#include <limits.h>

namespace prach {

namespace {

// packed word of the NOMA record
constexpr unsigned N_RA_BIT = 1u << 2, N_FAIL_BIT = 1u << 3, N_MSG2_BIT = 1u << 4, N_M3W_BIT = 1u << 5,
                   N_PRE_SHIFT = 6, N_RETX_SHIFT = 14;
constexpr int NOMA_VARIANT = 2;

struct NLds {
    int *cnt;    // [6*nP] transmitters per (sector, preamble)
    int *who;    // [6*nP] lowest transmitter index per bin
    int *sidx;   // [6*64] per sector: singleton UE indices (preamble order, then sorted order)
    double *sg;  // [6*64] their channel gains
    double *slg; // [6*64] ln(gain)
    int *scal;   // [32]
};
enum { N_NSUCC = 0, N_STATUS, N_PTC, N_FC, N_SUMT = 8, N_ND = 10 };

__device__ __forceinline__ NLds ncarve(char *smem, int nP) {
    NLds L;
    L.sg = reinterpret_cast<double *>(smem);
    L.slg = L.sg + 6 * 64;
    int *ip = reinterpret_cast<int *>(L.slg + 6 * 64);
    L.sidx = ip; ip += 6 * 64;
    L.scal = ip; ip += 32;
    L.cnt = ip; ip += 6 * nP;
    L.who = ip; ip += 6 * nP;
    return L;
}

} // namespace

size_t noma_kernel_lds_bytes(int nP) { return sizeof(double) * 2 * 6 * 64 + sizeof(int) * (6 * 64 + 32 + 2 * 6 * nP); }

__global__ __launch_bounds__(WG_THREADS) void noma_kernel(const TrialDev *__restrict__ params) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TrialDev P = params[blockIdx.x];
    const NLds L = ncarve(smem, P.nP);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nUE = P.nUE, nP = P.nP, aT = P.aT, nGrantUL = P.nGrantUL;
    const int *pre0 = P.n_pre0, *sector = P.n_sector;
    const double *gain = P.n_gain, *lgain = P.n_lgain;
    const FastMod fmP = make_fastmod(nP), fmB = make_fastmod(P.backoff), fmA = make_fastmod(aT);

    for (int i = tid; i < nUE; i += WG_THREADS) { // calloc + initUserInfo (NOMA.c:651-655)
        P.rec[i] = make_int4(0, 0, 0, 0);
        P.ptc[i] = 0; P.ftt[i] = 0; P.stt[i] = 0; P.fcnt[i] = 0; P.nd[i] = 0;
    }
    if (tid < 32) L.scal[tid] = 0;
    __syncthreads();

    int activeCheck = 0, tlast = -1, time_exit = P.stop;
    unsigned long long steps = 0;
    const int totgroups = (nUE + 63) >> 6;
    (void)totgroups;

    for (int t = 0; t < P.stop; t++) {
        steps++;
        tlast = t;
        const bool slot = (t % aT) == 0;
        const int prevAC = activeCheck;
        if (slot) { // NOMA.c:668-681
            activeCheck = P.sched[t / aT];
            for (int k = tid; k < 6 * nP; k += WG_THREADS) { L.cnt[k] = 0; L.who[k] = INT_MAX; }
            __syncthreads();
            // ---- pass A: activation + transmitter gather ----
            const int ngroups = (activeCheck + 63) >> 6;
            for (int g = w; g < ngroups; g += NW) {
                const int i = g * 64 + lane;
                if (i >= activeCheck) continue;
                int4 r = load_rec(&P.rec[i]);
                if (i >= prevAC) { // activeUE (NOMA.c:131-140): everything else comes from the activation table
                    r.x = t + 1; r.y = t; r.z = 0;
                    r.w = 1 | (pre0[i] << N_PRE_SHIFT);
                    P.ptc[i] = 1; P.ftt[i] = t + 1; P.nd[i] = P.n_nd0[i];
                    P.rec[i] = r;
                }
                const unsigned pk = (unsigned)r.w;
                const int act = pk & 3;
                // transmitter (NOMA.c:207): RA==0, txTime==time+1, msg2==0, nowBackoff<=0, RaFailed==0
                if (act == 1 && !(pk & (N_RA_BIT | N_FAIL_BIT | N_MSG2_BIT)) && r.x == t + 1 && now_backoff(r.z, t) <= 0) {
                    const int b = sector[i] * nP + (int)((pk >> N_PRE_SHIFT) & 0xff);
                    atomicAdd(&L.cnt[b], 1);
                    atomicMin(&L.who[b], i);
                }
            }
            __syncthreads();
            // ---- resolve: one wavefront per sector (NOMA.c:214-309) ----
            if (w < 6) {
                const int s = w;
                const bool single = lane < nP && L.cnt[s * nP + lane] == 1;
                const int myidx = single ? L.who[s * nP + lane] : -1;
                const unsigned long long sm = __ballot(single);
                const int count = __popcll(sm);
                if (count > 0) {
                    const int pos = __popcll(sm & lanemask_lt(lane));
                    if (count <= nGrantUL) { // NOMA.c:252-260
                        if (single) atomicOr(reinterpret_cast<unsigned *>(&P.rec[myidx]) + 3, PK_GRANT_BIT);
                    } else {
                        if (single) { L.sidx[s * 64 + pos] = myidx; L.sg[s * 64 + pos] = gain[myidx]; L.slg[s * 64 + pos] = lgain[myidx]; }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        // stable ascending rank by gain == the bubble sort with strict < (NOMA.c:90-103)
                        int uidx = -1;
                        double ug = 0, ulg = 0;
                        int rank = 0;
                        if (lane < count) {
                            uidx = L.sidx[s * 64 + lane]; ug = L.sg[s * 64 + lane]; ulg = L.slg[s * 64 + lane];
                            for (int j = 0; j < count; j++) { const double gj = L.sg[s * 64 + j]; rank += (gj < ug || (gj == ug && j < lane)) ? 1 : 0; }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (lane < count) { L.sidx[s * 64 + rank] = uidx; L.slg[s * 64 + rank] = ulg; }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        int cidx = -1;
                        double clg = 0;
                        if (lane < count) { cidx = L.sidx[s * 64 + lane]; clg = L.slg[s * 64 + lane]; }
                        unsigned long long valid = count >= 64 ? ~0ull : ((1ull << count) - 1ull);
                        int grants = 0;
                        bool grantme = false;
                        for (int i = 0; i < count - 1; i++) { // NOMA.c:268-298
                            if (!((valid >> i) & 1ull)) continue;
                            const double lgi = __shfl(clg, i);
                            const double diff = __dsub_rn(__dmul_rn(10.0, clg), __dmul_rn(10.0, lgi)); // 10*log(high) - 10*log(low)
                            const bool cond = lane > 0 && lane != i && lane < count && ((valid >> lane) & 1ull) && diff > 15.0;
                            const unsigned long long mj = __ballot(cond);
                            if (!mj) continue;
                            const int j = __ffsll((long long)mj) - 1;
                            valid &= ~((1ull << i) | (1ull << j));
                            if (grants < nGrantUL) {
                                grants++;
                                const int lowidx = __shfl(cidx, i);
                                unsigned k = P.nd[lowidx];
                                const int d1 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)lowidx, k, (unsigned)nUE, NOMA_VARIANT);
                                int decoded = 2; // both
                                if (d1 <= 644245094) { // (double)rand()/RAND_MAX < 0.3 (NOMA.c:284-285)
                                    const int d2 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)lowidx, k + 1, (unsigned)nUE, NOMA_VARIANT);
                                    decoded = d2 % 2; // index into rx[] = {low, high}
                                    k += 2;
                                } else k += 1;
                                if (lane == 0) P.nd[lowidx] = k;
                                if ((lane == i && (decoded == 2 || decoded == 0)) || (lane == j && (decoded == 2 || decoded == 1))) grantme = true;
                            }
                        }
                        { // leftovers in sorted order while grants remain (NOMA.c:299-307)
                            const bool left = lane < count && ((valid >> lane) & 1ull);
                            const unsigned long long lm = __ballot(left);
                            if (left && __popcll(lm & lanemask_lt(lane)) < nGrantUL - grants) grantme = true;
                        }
                        if (grantme) atomicOr(reinterpret_cast<unsigned *>(&P.rec[cidx]) + 3, PK_GRANT_BIT);
                    }
                }
            }
            __syncthreads();
        }
        // ---- pass B: msg2Results for transmitters (slot steps), resourceRequestAllocation (every ms) ----
        {
            const int ngroups = (activeCheck + 63) >> 6;
            int c_succ = 0;
            for (int g = w; g < ngroups; g += NW) {
                const int i = g * 64 + lane;
                int4 r = make_int4(0, 0, 0, 0);
                if (i < activeCheck) r = load_rec(&P.rec[i]);
                unsigned pk = (unsigned)r.w;
                const int act = pk & 3;
                const bool alive = i < activeCheck && !(pk & (N_RA_BIT | N_FAIL_BIT));
                const bool tx = slot && alive && act == 1 && r.x == t + 1 && now_backoff(r.z, t) <= 0; // NOMA.c:692
                const bool m3 = alive && act == 2 && (pk & N_MSG2_BIT) && r.x == t;                    // NOMA.c:501
                if (!__any(tx || m3)) continue;
                const bool granted = tx && (pk & PK_GRANT_BIT);
                const bool txfail = tx && !granted && !(pk & N_MSG2_BIT);
                const bool m3first = m3 && !(pk & N_M3W_BIT), m3to = m3 && (pk & N_M3W_BIT);
                int retx = (int)((pk >> N_RETX_SHIFT) & 0xff);
                const bool perm = txfail && retx + 1 >= P.maxMsg2; // msg1ReTx reaches maxMsg1ReTx: dropped for good
                const int need = m3to ? 2 : ((txfail && perm) ? 2 : ((txfail || m3first) ? 1 : 0));
                int d1 = 0, d2 = 0;
                unsigned k = 0;
                if (__any(need > 0)) {
                    if (need > 0) k = P.nd[i];
                    d1 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k, (unsigned)nUE, NOMA_VARIANT);
                    if (__any(need > 1)) d2 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k + 1, (unsigned)nUE, NOMA_VARIANT);
                    if (need > 0) P.nd[i] = k + (unsigned)need;
                }
                bool dirty = false;
                if (granted) { // NOMA.c:491-497
                    pk = (pk & ~(3u | N_M3W_BIT | PK_GRANT_BIT)) | 2u | N_MSG2_BIT;
                    r.x += 10; P.stt[i] = r.x;
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
                    } else {
                        pk = (pk & ~(0xffu << N_RETX_SHIFT)) | ((unsigned)retx << N_RETX_SHIFT);
                        P.ptc[i] = P.ptc[i] + 1;
                        r.z = enc_backoff(X, t);
                    }
                    dirty = true;
                } else if (m3first) { // NOMA.c:502-512
                    const float pf = (float)d1 / (float)2147483647;
                    if ((double)pf > 0.1) {
                        const int nbo = now_backoff(r.z, t);
                        pk = (pk & ~3u) | N_RA_BIT;
                        r.y = (t - r.y) + 6; r.z = nbo;
                        c_succ++;
                    } else { r.x += 49; pk |= N_M3W_BIT; }
                    dirty = true;
                } else if (m3to) { // NOMA.c:514-543
                    P.fcnt[i] = P.fcnt[i] + 1; // msg3Faile
                    const int np = fastmod(d1, fmP);
                    const int tmp = fastmod(d2, fmB);
                    r.x = slot_align_fm(r.x + tmp, fmA);
                    P.stt[i] = r.x;
                    r.z = enc_backoff(r.x - t - 1, t);
                    pk = (pk & ~(3u | N_MSG2_BIT | (0xffu << N_PRE_SHIFT) | (0xffu << N_RETX_SHIFT))) | 1u | ((unsigned)np << N_PRE_SHIFT);
                    P.ptc[i] = 0;
                    r.y = t;
                    dirty = true;
                }
                if (dirty) { r.w = (int)pk; P.rec[i] = r; }
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) c_succ += __shfl_down(c_succ, d);
            if (lane == 0 && c_succ) atomicAdd(&L.scal[N_NSUCC], c_succ);
        }
        __syncthreads();
        if (L.scal[N_NSUCC] == nUE) { time_exit = t; break; } // NOMA.c:707-710
    }
    __syncthreads();

    // saveResult (NOMA.c:618-625) + per-UE dump
    const int tend = tlast + 1;
    long long sumT = 0;
    int ptcS = 0, fcS = 0;
    unsigned long long ndS = 0;
    for (int i = tid; i < nUE; i += WG_THREADS) {
        const int4 r = P.rec[i];
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
            o.rarWindow = 0;
            o.maxRarCounter = (int)((pk >> N_RETX_SHIFT) & 0xff);        // NOMA: msg1ReTx
            o.preambleTxCounter = ntx;                                    // NOMA: nTxPreamble
            o.msg2Flag = (pk & N_MSG2_BIT) ? 1 : 0;                       // NOMA: msg2
            o.connectionRequest = (act == 2 || ra) ? ((pk & N_M3W_BIT) ? 49 : 0) : ((pk & N_M3W_BIT) ? 49 : 0); // NOMA: msg3Wait
            o.msg4Flag = ra ? 1 : 0;                                      // NOMA: RA
            o.failCount = (fail ? 1 : 0) | (P.fcnt[i] << 16);             // NOMA: RaFailed | msg3Faile << 16
            P.logs[i] = o;
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
    if (tid == 0) {
        DevResult o;
        o.status = L.scal[N_STATUS]; o.time_exit = time_exit; o.nSuccess = L.scal[N_NSUCC]; o.collisionPreambles = 0; o.totalPreambleTxop = 0;
        o.activeCheck = activeCheck; o.continueFailed = 0; o.finalSuccess = L.scal[N_NSUCC]; o.ptcSum = L.scal[N_PTC]; o.fcSum = L.scal[N_FC];
        o.draws = *reinterpret_cast<unsigned long long *>(&L.scal[N_ND]); o.steps = steps;
        o.sumTimer = *reinterpret_cast<long long *>(&L.scal[N_SUMT]);
        for (int k = 0; k < 4; k++) o.dbg[k] = 0;
        for (int k = 0; k < 8; k++) o.stamps6[k] = 0;
        *P.out = o;
    }
}

hipError_t launch_noma_kernel(const TrialDev *params, int ntrials, int maxP, hipStream_t stream) {
    const size_t lds = noma_kernel_lds_bytes(maxP);
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&noma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return rc;
    hipLaunchKernelGGL(noma_kernel, dim3(ntrials), dim3(WG_THREADS), lds, stream, params);
    return hipGetLastError();
}

} // namespace prach
