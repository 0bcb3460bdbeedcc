// prach_kernels.hip — hand-written HIP for gfx950 (MI355X / CDNA4): the per-subframe PRACH
// random-access loop of RandomAccessSimulatorBeta.c:111-197 / RandomAccessWithNOMA.c:267-351.
//
// One 1024-thread workgroup (16 wave64) simulates one (seed, nUE) trial; a batch of trials is a grid
// of workgroups.  UE state is struct-of-arrays with ONE 16-byte hot record per UE (prach_device.h),
// read with one coalesced global_load_dwordx4 per lane per subframe; timer and backoff counters are
// stored as bases so waiting UEs are never rewritten.
//
// The reference runs the UE loop in index order with immediate side effects (preambleCollision
// bumps every matched UE, Beta.c:353-358; grants go to callers in index order, Beta.c:336-347).
// That order is reproduced EXACTLY, in parallel, by the decomposition validated on the CPU in
// oracle/phase_model.c:
//   pass      each wave owns a contiguous index range; per UE: deferred apply of the previous
//             subframe's outcome, activation (Beta.c:136-146), selectPreamble (Beta.c:229-312),
//             requestResourceAllocation (Beta.c:371-411) on its own state; per-wave LDS histogram of
//             "pre-members" per preamble bucket (the UEs a preambleCollision scan would match), the
//             lowest-index would-be caller per bucket, and an index-ordered list of special events
//             (late joiners etc.), ranked with wavefront ballots;
//   resolve   prefix over waves, first caller / check (= scan count) per call, singleton callers
//             ranked by index for the RAR grants of the 5 ms window, collision counters;
//   apply     folded into the next subframe's pass (txTime bumps, Msg2 success).
// RNG: Philox4x32-10 per (UE, draw#) lane-locally, or the reference's own glibc rand() stream
// addressed through an index-ordered prefix sum of per-UE draw counts (bit-exact vs the reference).
#include "prach_device.h"
#include "prach_device_fn.h"
#include "prach_ue_body.h"
#include <limits.h>

namespace prach {

// LDS carve-up (dynamic shared memory, 16-byte aligned base; prach_device.h sizes)
constexpr int DEADW_T = 1024; // 32768 groups = 2M UEs; larger trials simply do not skip
struct Lds {
    Event *evl;    // [EVCAP]
    int *sidx;     // [SCAP]
    int *rclist;   // [RCCAP]
    int *scal;     // [64] scalars, see S_*
    int *evcnt;    // [NW]
    int *evoff;    // [NW+1]
    unsigned *wdraws; // [NW]
    int *wavehist; // [NW*nP]  per-wave pre-member histogram (then: exclusive prefix over waves)
    int *smidx;    // [NW*nP]  per-wave lowest-index STAY pre-member per bucket
    int *smle;     // [NW*nP]  its inclusive rank among the wave's pre-members of the bucket
    int *total;    // [nP]
    int *gsm;      // [nP] global lowest-index STAY pre-member
    int *gsmle;    // [nP]
    int *fcall;    // [nP] index of the first caller on the bucket this subframe (INT_MAX: none)
    int *lcall;    // [nP] index of the last caller (-1: none)
    unsigned *dead; // [DEADW_T] bitmap of 64-UE groups in which every UE has finished (skipped by every pass)
};
enum { S_NSUCC = 0, S_COLL, S_TXOP, S_CONTF, S_FINS, S_NS, S_NRC, S_NPOST, S_STATUS, S_PTC, S_FC, S_ND_LO,
       S_SUMT_LO = 16 /* 64-bit at [16,17] */, S_ND64 = 18 /* 64-bit at [18,19] */,
       S_SGC = 24 /* [24,30): sectorGrants[6], WithNOMA:260 */, S_SGN = 32 /* [32,38): singleton callers of this subframe per sector */ };

__device__ __forceinline__ Lds carve(char *smem, int nP) {
    Lds L;
    L.evl = reinterpret_cast<Event *>(smem);
    int *ip = reinterpret_cast<int *>(smem + sizeof(Event) * EVCAP);
    L.sidx = ip; ip += SCAP;
    L.rclist = ip; ip += RCCAP;
    L.scal = ip; ip += 64;
    L.evcnt = ip; ip += NW;
    L.evoff = ip; ip += NW + 16;
    L.wdraws = reinterpret_cast<unsigned *>(ip); ip += NW;
    L.wavehist = ip; ip += NW * nP;
    L.smidx = ip; ip += NW * nP;
    L.smle = ip; ip += NW * nP;
    L.total = ip; ip += nP;
    L.gsm = ip; ip += nP;
    L.gsmle = ip; ip += nP;
    L.fcall = ip; ip += nP;
    L.lcall = ip; ip += nP;
    L.dead = reinterpret_cast<unsigned *>(ip); ip += DEADW_T;
    return L;
}
size_t trial_kernel_lds_bytes(int nP) {
    return sizeof(Event) * EVCAP + sizeof(int) * (SCAP + RCCAP + 64 + NW + NW + 16 + NW + 3 * NW * nP + 5 * nP + DEADW_T);
}

// ---------------------------------------------------------------------------------------------
// one pass over the UEs [0, activeCheck); MODE: 0 fused (apply + activate + select)  [philox]
//                                                1 apply + activate + count draws     [glibc]
//                                                2 select only                        [glibc]
//                                                3 apply only                         [final]
// ---------------------------------------------------------------------------------------------
template <int MODE, bool GLIBC>
__device__ __forceinline__ void ue_pass(const TrialG &P, const Lds &L, const int t, const int prevAC,
                                        const int activeCheck, const unsigned long long stepbase) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nP = P.nP, aT = P.aT;
    const FastMod fmP = make_fastmod(nP), fmB = make_fastmod(P.backoff), fmA = make_fastmod(aT), fm5 = make_fastmod(5);
    const bool withnoma = P.variant == PRACH_VARIANT_WITHNOMA_C;
    const UeK K{P.maxRarWindow, P.maxMsg2, aT, withnoma, fmP, fmB, fmA, fm5};
    ColdGlobal cold{P.ptc, P.ftt, P.stt, P.fcnt};
    const int groups = (activeCheck + 63) >> 6;
    const int gper = (groups + NW - 1) / NW;
    const int g0 = w * gper, g1 = min(groups, g0 + gper);
    int evn = 0;
    unsigned wdraw = 0;
    unsigned long long woff = 0;
    if (MODE == 2 && GLIBC) {
        woff = stepbase;
        for (int k = 0; k < w; k++) woff += L.wdraws[k];
    }
    int c_succ = 0, c_contf = 0;
    const int tp = t - 1;

    for (int g = g0; g < g1; g++) {
        if (g < DEADW_T * 32 && ((L.dead[g >> 5] >> (g & 31)) & 1u)) continue; // every UE of the group has finished
        const int i = g * 64 + lane;
        const bool valid = i < activeCheck;
        int4 r = make_int4(-1, 0, 0, 0);
        if (valid) r = load_rec(&P.rec[i]);
        UeState u = unpack(r);
        bool dirty = false;

        if (MODE != 2) {
            // ---- deferred outcome of subframe t-1 (prach_ue_body.h ue_apply) ----
            dirty = ue_apply(u, ((unsigned)r.w & PK_GRANT_BIT) != 0u, i, tp, fmA, CallTables{L.fcall, L.lcall});
            // ---- activation (Beta.c:136-146; activateUEs WithNOMA:383-394 also draws twice) ----
            if (MODE != 3 && valid && i >= prevAC) {
                ue_activate(u, i, t, cold);
                if (!GLIBC && withnoma) P.nd[i] = 2;
                dirty = true;
            }
        }

        // what this UE will do in subframe t, from its own state
        const UePlan pl = ue_plan(u, t, P.maxRarWindow, P.maxMsg2);
        const int need = pl.need;

        if (MODE == 1 || MODE == 3) {
            if (MODE == 1) wdraw += (unsigned)need;
            if (dirty) store_rec(&P.rec[i], pack(u));
            continue;
        }

        // nothing to do for the whole wavefront?  (waiting / finished / not yet arrived UEs)
        if (!__any(pl.busy || dirty)) {
            if (g < DEADW_T * 32 && __all(i >= P.nUE || u.act == ACT_DONE) && lane == 0) atomicOr(&L.dead[g >> 5], 1u << (g & 31));
            continue;
        }

        // ---- draws ----
        int d1 = 0, d2 = 0;
        if (__any(need > 0)) {
            if (GLIBC) {
                const int x = wave_scan_incl(need); // (DPP: prach_device_fn.h)
                const unsigned long long o = woff + (unsigned long long)(x - need);
                if (need > 0) d1 = P.stream[o];
                if (need > 1) d2 = P.stream[o + 1];
                woff += (unsigned long long)__builtin_amdgcn_readlane(x, 63);
            } else {
                unsigned k = 0;
                if (need > 0) k = P.nd[i];
                d1 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k, (unsigned)P.nUE, (unsigned)P.variant);
                if (__any(need > 1))
                    d2 = philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, k + 1, (unsigned)P.nUE, (unsigned)P.variant);
                if (need > 0) P.nd[i] = k + (unsigned)need;
            }
        }

        // ---- selectPreamble / requestResourceAllocation on own state (prach_ue_body.h ue_select) ----
        const UeOut o = ue_select(u, pl, d1, d2, i, t, t % aT, K, cold, c_succ, c_contf);
        dirty = dirty || o.dirty;
        const int oldp = o.oldp, evq = o.evq;
        const bool member_pre = o.member_pre;
        // (this kernel's event list format: EV_* of prach_device.h; a passive member is an ordered event here)
        const int evtype = o.passive ? EV_PASSIVE : (o.evtype == UEV_CALLER ? EV_CALLER : (o.evtype == UEV_RESETCAND ? EV_RESETCAND : (o.evtype == UEV_RJOIN ? EV_RJOIN : 0)));
        const int evp = o.passive ? oldp : o.evp;
        const int pend = u.pend;

        // ---- bucket bookkeeping for the resolver (wavefront-level) ----
        const bool stay = pend == PEND_STAY;
        bool winner = false;
        if (__any(stay)) {
            if (stay) {
                int *smp = &L.smidx[w * nP + oldp];
                if (__atomic_load_n(smp, __ATOMIC_RELAXED) == INT_MAX) {
                    atomicMin(smp, i);
                    winner = __atomic_load_n(smp, __ATOMIC_RELAXED) == i;
                }
            }
        }
        const bool rankq = winner || evtype == EV_CALLER || evtype == EV_RESETCAND;
        int le = 0;
        {
            unsigned long long qm = __ballot(rankq);
            const int qp = winner ? oldp : evp;
            while (qm) {
                const int l = __ffsll((long long)qm) - 1;
                qm &= qm - 1;
                const int p = __builtin_amdgcn_readlane(qp, __builtin_amdgcn_readfirstlane(l));
                const unsigned long long mm = __ballot(member_pre && oldp == p);
                if (lane == l) le = L.wavehist[w * nP + p] + __popcll(mm & lanemask_le(lane));
            }
        }
        if (member_pre) atomicAdd(&L.wavehist[w * nP + oldp], 1);
        if (winner) L.smle[w * nP + oldp] = le;
        {
            const unsigned long long em = __ballot(evtype != 0);
            if (em) {
                if (evtype != 0) {
                    Event e;
                    e.idx = i; e.info = evtype | (evp << 8) | (evq << 16); e.le = le; e.pad = 0;
                    ((Event *)P.evbuf)[g0 * 64 + evn + __popcll(em & lanemask_lt(lane))] = e;
                }
                evn += __popcll(em);
            }
        }
        if (dirty) store_rec(&P.rec[i], pack(u));
    }

    if (MODE == 1) {
        wdraw = (unsigned)wave_sum((int)wdraw);
        if (lane == 0) L.wdraws[w] = wdraw;
    }
    if (MODE == 0 || MODE == 2) {
        c_succ = wave_sum(c_succ); c_contf = wave_sum(c_contf);
        if (lane == 0) {
            L.evcnt[w] = evn;
            if (c_succ) { atomicAdd(&L.scal[S_NSUCC], c_succ); atomicAdd(&L.scal[S_FINS], c_succ); }
            if (c_contf) atomicAdd(&L.scal[S_CONTF], c_contf);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// resolve subframe t: who calls preambleCollision, with what scan count, who gets the RAR grants
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void resolve(const TrialG &P, const Lds &L, const int activeCheck, int &grantCheck) {
    const int tid = threadIdx.x;
    const int nP = P.nP;
    const bool withnoma = P.variant == PRACH_VARIANT_WITHNOMA_C;
    const int groups = (activeCheck + 63) >> 6;
    const int gper = (groups + NW - 1) / NW;

    __syncthreads(); // pass complete: wavehist / smidx / smle / evcnt / event segments are final
    for (int p = tid; p < nP; p += WG_THREADS) {
        int acc = 0, sm = INT_MAX, sle = 0;
        for (int w = 0; w < NW; w++) {
            const int s = L.smidx[w * nP + p];
            if (sm == INT_MAX && s != INT_MAX) { sm = s; sle = acc + L.smle[w * nP + p]; }
            const int h = L.wavehist[w * nP + p];
            L.wavehist[w * nP + p] = acc;
            acc += h;
        }
        L.total[p] = acc; L.gsm[p] = sm; L.gsmle[p] = sle; L.fcall[p] = sm; L.lcall[p] = -1;
    }
    if (tid <= NW) {
        int s = 0;
        for (int k = 0; k < tid; k++) s += L.evcnt[k];
        L.evoff[tid] = s;
    }
    if (tid == 0) { L.scal[S_NS] = 0; L.scal[S_NRC] = 0; L.scal[S_NPOST] = 0; }
    __syncthreads();

    const int N = L.evoff[NW];
    Event *const EVA = N <= EVCAP ? L.evl : (Event *)P.evbuf2; // LDS or global: a generic pointer
    for (int k = tid; k < N; k += WG_THREADS) {
        int w = 0;
        while (w + 1 < NW && L.evoff[w + 1] <= k) w++;
        Event e = ((const Event *)P.evbuf)[w * gper * 64 + (k - L.evoff[w])];
        const int type = e.info & 0xff, p = (e.info >> 8) & 0xff;
        e.le += L.wavehist[w * nP + p];
        if (type == EV_CALLER) atomicMin(&L.fcall[p], e.idx);
        else if (type == EV_RESETCAND) { const int s = atomicAdd(&L.scal[S_NRC], 1); if (s < RCCAP) L.rclist[s] = k; }
        else atomicAdd(&L.scal[S_NPOST], 1);
        EVA[k] = e;
    }
    __syncthreads();

    // reset cycles that could land on this subframe: strictly in index order (rare, few)
    const int nrc = L.scal[S_NRC];
    if (nrc > 0) {
        if (tid == 0 && nrc <= RCCAP) {
            const int n = nrc;
            for (int a = 1; a < n; a++) { // insertion sort: list position == index order
                const int v = L.rclist[a];
                int b = a - 1;
                while (b >= 0 && L.rclist[b] > v) { L.rclist[b + 1] = L.rclist[b]; b--; }
                L.rclist[b + 1] = v;
            }
            for (int a = 0; a < n; a++) {
                Event e = EVA[L.rclist[a]];
                const int p = (e.info >> 8) & 0xff, q = (e.info >> 16) & 0xff;
                if (L.fcall[q] < e.idx) { e.info = 0; EVA[L.rclist[a]] = e; } // bumped before its turn
                else if (e.idx < L.fcall[p]) L.fcall[p] = e.idx;
            }
        } else if (tid == 0) {
            // more candidates than the staged list holds (degenerate parameters: every expiry is a reset cycle):
            // the event list itself is in index order, walk it — slow, exact, any size
            for (int k = 0; k < N; k++) {
                Event e = EVA[k];
                if ((e.info & 0xff) != EV_RESETCAND) continue;
                const int p = (e.info >> 8) & 0xff, q = (e.info >> 16) & 0xff;
                if (L.fcall[q] < e.idx) { e.info = 0; EVA[k] = e; }
                else if (e.idx < L.fcall[p]) L.fcall[p] = e.idx;
            }
        }
        __syncthreads();
    }

    // every call: scan count `check` (Beta.c:321-330), counters (Beta.c:334,349-351 / WithNOMA:650-652)
    const int npost = L.scal[S_NPOST];
    int *const sing = (N + nP <= SCAP) ? L.sidx : (int *)P.sidx;
    for (int k = tid; k < N + nP; k += WG_THREADS) {
        int idx = 0, p = 0, le = 0;
        bool caller = false;
        if (k < N) {
            const Event e = EVA[k];
            const int type = e.info & 0xff;
            if (type == EV_CALLER || type == EV_RESETCAND) { caller = true; idx = e.idx; p = (e.info >> 8) & 0xff; le = e.le; }
        } else {
            p = k - N;
            const int sm = L.gsm[p];
            if (sm != INT_MAX && sm == L.fcall[p]) { caller = true; idx = sm; le = L.gsmle[p]; }
        }
        if (!caller) continue;
        const bool first = idx == L.fcall[p];
        int post = 0;
        if (npost > 0) { // members that stayed matched after their own turn (passive / Msg3 re-entry): rare
            int prev = -1;
            for (int j = 0; j < N; j++) {
                const Event ej = EVA[j];
                const int tj = ej.info & 0xff;
                if ((tj == EV_CALLER || tj == EV_RESETCAND) && ((ej.info >> 8) & 0xff) == p && ej.idx < idx && ej.idx > prev)
                    prev = ej.idx;
            }
            const int sm = L.gsm[p];
            if (!first && sm == L.fcall[p] && sm < idx && sm > prev) prev = sm;
            for (int j = 0; j < N; j++) {
                const Event ej = EVA[j];
                const int tj = ej.info & 0xff;
                if (((ej.info >> 8) & 0xff) != p || ej.idx >= idx) continue;
                if (tj == EV_RJOIN && ej.idx > prev) post++;
                if (tj == EV_PASSIVE && first) post++;
            }
        }
        const int check = 1 + (first ? L.total[p] - le : 0) + post;
        atomicMax(&L.lcall[p], idx);
        if (check == 1) {
            const int s = atomicAdd(&L.scal[S_NS], 1);
            sing[s] = idx;
            atomicAdd(&L.scal[S_TXOP], 1);
        } else if (withnoma) {
            atomicAdd(&L.scal[S_COLL], check);
            atomicAdd(&L.scal[S_TXOP], check);
        } else {
            atomicAdd(&L.scal[S_COLL], 1);
            atomicAdd(&L.scal[S_TXOP], 1);
        }
    }
    __syncthreads();

    // RAR grants to the first (nGrantUL-1-grantCheck) singleton callers in index order (Beta.c:336-347)
    const int ns = L.scal[S_NS];
    if (P.flags & PRACH_FLAG_SECTOR_GRANTS) {
        // the dormant per-sector grant test (WithNOMA:626-637 with the call of :312): every 60-degree sector has its own budget,
        // grantCheck[sector] counts that sector's singleton callers of the 5 ms window
        for (int j = tid; j < ns; j += WG_THREADS) {
            const int my = sing[j];
            const int sj = P.sector[my];
            int rank = 0;
            for (int m = 0; m < ns; m++) { const int o = sing[m]; rank += (o < my && P.sector[o] == sj) ? 1 : 0; }
            if (rank < P.nGrantUL - 1 - L.scal[S_SGC + sj]) grant_rec(&P.rec[my]);
            atomicAdd(&L.scal[S_SGN + sj], 1);
        }
        __syncthreads();
        if (tid < 6) { L.scal[S_SGC + tid] += L.scal[S_SGN + tid]; L.scal[S_SGN + tid] = 0; }
        for (int k = tid; k < NW * nP; k += WG_THREADS) { L.wavehist[k] = 0; L.smidx[k] = INT_MAX; }
        __syncthreads();
        return;
    }
    const int G = max(0, P.nGrantUL - 1 - grantCheck);
    for (int j = tid; j < ns; j += WG_THREADS) {
        const int my = sing[j];
        int rank = 0;
        for (int m = 0; m < ns; m++) rank += sing[m] < my ? 1 : 0;
        if (rank < G) {
            grant_rec(&P.rec[my]);
        }
    }
    grantCheck += ns;
    for (int k = tid; k < NW * nP; k += WG_THREADS) { L.wavehist[k] = 0; L.smidx[k] = INT_MAX; }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// the trial kernel
// ---------------------------------------------------------------------------------------------
template <bool GLIBC>
__global__ __launch_bounds__(WG_THREADS) void trial_kernel(const TrialDev *__restrict__ params) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TrialG P(params[blockIdx.x]);
    const Lds L = carve(smem, P.nP);
    const int tid = threadIdx.x;
    const int nUE = P.nUE, nP = P.nP, aT = P.aT;

    // calloc + initialUE (Beta.c:78-83)
    for (int i = tid; i < nUE; i += WG_THREADS) {
        store_rec(&P.rec[i], make_int4(-1, 0, 0, 0));
        P.ptc[i] = 0; P.ftt[i] = 0; P.stt[i] = 0; P.fcnt[i] = 0; P.nd[i] = 0;
    }
    for (int k = tid; k < NW * nP; k += WG_THREADS) { L.wavehist[k] = 0; L.smidx[k] = INT_MAX; L.smle[k] = 0; }
    for (int k = tid; k < nP; k += WG_THREADS) { L.fcall[k] = INT_MAX; L.lcall[k] = -1; L.total[k] = 0; L.gsm[k] = INT_MAX; L.gsmle[k] = 0; }
    if (tid < 64) L.scal[tid] = 0;
    for (int k = tid; k < DEADW_T; k += WG_THREADS) L.dead[k] = 0;
    if (tid < NW) { L.evcnt[tid] = 0; L.wdraws[tid] = 0; }
    __syncthreads();

    int activeCheck = 0, grantCheck = 0, tlast = -1, time_exit = P.stop;
    unsigned long long base = 0; // glibc: draws consumed so far (relative to the stream window)
    unsigned long long steps = 0;
    int status = PRACH_OK;

    for (int t = 0; t < P.stop; t++) {
        steps++;
        tlast = t;
        if (t % 5 == 0) { // Beta.c:112 (hard-coded 5); WithNOMA:268-274
            grantCheck = 0;
            if ((P.flags & PRACH_FLAG_SECTOR_GRANTS) && threadIdx.x < 6) L.scal[S_SGC + threadIdx.x] = 0; // (a barrier follows in the pass)
        }
        const int prevAC = activeCheck;
        if (t % aT == 0 && activeCheck != nUE) activeCheck = P.sched[t / aT]; // Beta.c:121-134
        if ((P.flags & PRACH_FLAG_SECTOR_GRANTS) && activeCheck > prevAC) {
            // activateUEs (WithNOMA:393-410): theta from the first of the two activation draws fixes the UE's sector (r is never read)
            for (int i = prevAC + (int)threadIdx.x; i < activeCheck; i += WG_THREADS) {
                // (glibc: a position beyond the stream window reads entry 0 instead — the window check below then ends the trial with
                //  PRACH_ERR_STREAM and the engine reruns it with a larger window, so the value is never used)
                const unsigned long long so = base + 2ull * (unsigned long long)(i - prevAC);
                const int d = GLIBC ? P.stream[so < P.stream_len ? so : 0ull]
                                    : philox_draw31(P.seed_lo, P.seed_hi, (unsigned)i, 0u, (unsigned)nUE, (unsigned)P.variant);
                P.sector[i] = sector_of_draw(d);
            }
        }
        if (GLIBC) {
            const unsigned long long actdraws =
                P.variant == PRACH_VARIANT_WITHNOMA_C ? 2ull * (unsigned long long)(activeCheck - prevAC) : 0ull;
            ue_pass<1, true>(P, L, t, prevAC, activeCheck, 0);
            __syncthreads();
            unsigned long long tot = actdraws;
            for (int k = 0; k < NW; k++) tot += L.wdraws[k];
            if (base + tot > P.stream_len) { status = PRACH_ERR_STREAM; time_exit = t; break; }
            ue_pass<2, true>(P, L, t, prevAC, activeCheck, base + actdraws);
            base += tot;
        } else {
            ue_pass<0, false>(P, L, t, prevAC, activeCheck, 0);
        }
        resolve(P, L, activeCheck, grantCheck);
        if (L.scal[S_NSUCC] == nUE) { time_exit = t; break; } // Beta.c:180 (loop variable not advanced)
    }
    __syncthreads();
    if (status == PRACH_OK && tlast >= 0) {
        ue_pass<3, GLIBC>(P, L, tlast + 1, activeCheck, activeCheck, 0);
    }
    __syncthreads();

    // end-of-trial sums (Beta.c:185-197) and the logged fields (Beta.c:501-508)
    const int tend = tlast + 1;
    long long sumT = 0;
    int ptcS = 0, fcS = 0;
    unsigned long long ndS = 0;
    for (int i = tid; i < nUE; i += WG_THREADS) {
        const int4 r = load_rec_plain(&P.rec[i]);
        const int act = (r.w >> PK_ACT_SHIFT) & 3, conn = (r.w >> PK_CONN_SHIFT) & 3, pre = (r.w >> PK_PRE_SHIFT) & 0xff,
                  rar = (r.w >> PK_RAR_SHIFT) & 0xff, mrc = (r.w >> PK_MRC_SHIFT) & 0xff;
        const int timer = act == ACT_IDLE ? -1 : (act == ACT_DONE ? r.y : tend - r.y);
        const int ptc = P.ptc[i], fc = P.fcnt[i];
        if (act == ACT_DONE) { sumT += timer; ptcS += ptc; fcS += fc; }
        ndS += P.nd[i];
        P.timers[i] = act == ACT_DONE ? timer : INT_MIN;
        if (P.logs) {
            prach_ue_log o;
            o.idx = i; o.timer = timer; o.active = act - 1; o.txTime = r.x; o.firstTxTime = P.ftt[i];
            o.secondTxTime = P.stt[i]; o.nowBackoff = now_backoff(r.z, tend); o.preamble = pre - 1;
            o.preambleChange = pre != 0; o.rarWindow = rar; o.maxRarCounter = mrc; o.preambleTxCounter = ptc;
            o.msg2Flag = (act == ACT_M3 || act == ACT_DONE); o.connectionRequest = conn == 2 ? 48 : conn;
            o.msg4Flag = act == ACT_DONE; o.failCount = fc;
            store_log(P.logs, i, o);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sumT += __shfl_down(sumT, d); ptcS += __shfl_down(ptcS, d); fcS += __shfl_down(fcS, d); ndS += __shfl_down(ndS, d);
    }
    if ((tid & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&L.scal[S_SUMT_LO]), (unsigned long long)sumT);
        atomicAdd(reinterpret_cast<unsigned long long *>(&L.scal[S_ND64]), ndS);
        atomicAdd(&L.scal[S_PTC], ptcS);
        atomicAdd(&L.scal[S_FC], fcS);
    }
    __syncthreads();
    if (tid == 0) {
        DevResult o;
        o.status = status != PRACH_OK ? status : L.scal[S_STATUS];
        o.time_exit = time_exit;
        o.nSuccess = L.scal[S_NSUCC];
        o.collisionPreambles = L.scal[S_COLL];
        o.totalPreambleTxop = L.scal[S_TXOP];
        o.activeCheck = activeCheck;
        o.continueFailed = L.scal[S_CONTF];
        o.finalSuccess = L.scal[S_FINS];
        o.ptcSum = L.scal[S_PTC];
        o.fcSum = L.scal[S_FC];
        o.draws = GLIBC ? base : *reinterpret_cast<unsigned long long *>(&L.scal[S_ND64]);
        o.steps = steps;
        o.sumTimer = *reinterpret_cast<long long *>(&L.scal[S_SUMT_LO]);
        o.dbg[0] = o.dbg[1] = o.dbg[2] = o.dbg[3] = 0;
        *(DevResult *)P.out = o;
    }
}

hipError_t launch_trial_kernel(const TrialDev *params, int ntrials, int rng_mode, int maxP, hipStream_t stream) {
    const size_t lds = trial_kernel_lds_bytes(maxP);
    hipError_t rc;
    if (rng_mode == PRACH_RNG_GLIBC) {
        rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&trial_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return rc;
        hipLaunchKernelGGL(trial_kernel<true>, dim3(ntrials), dim3(WG_THREADS), lds, stream, params);
    } else {
        rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&trial_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return rc;
        hipLaunchKernelGGL(trial_kernel<false>, dim3(ntrials), dim3(WG_THREADS), lds, stream, params);
    }
    return hipGetLastError();
}

} // namespace prach
