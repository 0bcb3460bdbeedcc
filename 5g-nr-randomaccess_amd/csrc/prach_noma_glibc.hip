// prach_noma_glibc.hip — NOMA.c (PRACH_VARIANT_NOMA_C) in the REFERENCE'S OWN rand() stream, on the GPU.
//
// Why this path is separate from prach_noma.hip: NOMA.c draws everything from one global stream, and activeUE (NOMA.c:131-192) consumes a
// DATA-DEPENDENT number of draws per UE (two rejection loops) through double-precision libm calls (cos, sin, sqrt, log, pow), so every stream
// position is data dependent.  One workgroup per trial; the device copy of the stream is consumed at exactly the positions the reference's
// index-ordered loops reach (block-wide exclusive prefix sums over the draw counts, chunk after chunk in index order, phase after phase in the
// reference's order).  Two forms:
//   * noma_glibc_trial_kernel (round 3, the default): the WHOLE trial in ONE launch.  activeUE runs on the device (prach_noma_act.h; the arrivals'
//     stream positions by iteration), the passes run over a list of the UEs that are still alive (a band of a few thousand at nUE = 100 000),
//     the grouping is one wavefront's rank sort + ballot pairing.  141 ms for nUE = 100 000 (round 2: 1.5 s); the trials of a call side by side.  The device's cos / sin / log are not
//     the host libm's to the last bit: a value inside the error band (a float rounding boundary, the rejection threshold, two gains too
//     close to order) ends the launch with NOMA_GLIBC_AMBIGUOUS (a few percent of the trials) and the trial is run again in the other form:
//   * noma_glibc_slot_kernel: one launch per access slot, the slot's arrivals activated on the HOST between two launches with the libm the
//     reference links (prach_noma_activation_stream) — bit-identical by construction, 2000 launches and host round trips per trial; with the same
//     list of live UEs (kept on the device from launch to launch) 148 ms at nUE = 100 000, one trial at a time.
// This is the bit-exact-vs-the-reference's-own-files mode of config 4, not the throughput mode (Philox, prach_noma.hip).  Reference:
// NOMA.c:131-192 (activeUE), :194-324 / :325-447 (grouping), :449-498 (msg2Results), :499-546 (resourceRequestAllocation), :665-711 (the time step).
#include "prach_device.h"
#include "prach_device_fn.h"
#include "prach_noma_act.h"

#pragma clang fp contract(off) // (a * b + c stays two roundings, as in the reference built for baseline x86-64)

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <limits.h>
#include <vector>

namespace prach {

namespace {

struct alignas(8) NUe { // UserInfo (NOMA.c:8-39), the fields the simulation reads
    int timer, active, txTime, firstTxTime, secondTxTime, nowBackoff, preamble, sector, rarWindow, msg1ReTx, nTxPreamble, msg2, msg3Wait,
        msg3Faile, RaFailed, RA;
    double gain, lgain; // channelGain and its natural log (host libm: the pairing test of NOMA.c:276 uses 10*log(high) - 10*log(low))
};
static_assert(sizeof(NUe) == 80, "NUe layout");

struct SlotCtl { // per-launch control block (device memory, mirrored on the host)
    unsigned long long pos;     // in: stream position behind this slot's activations; out: position behind the slot
    unsigned long long stream_len;
    int time, activeCheck, nSuccess, exit_time; // exit_time >= 0: every UE has succeeded at that subframe (NOMA.c:707-710)
    int status, pad;
};

struct NParams {
    int nUE, nP, backoff, nGrantUL, maxRarWindow, maxMsg1ReTx, aT, stop, nonsector;
};

#define NG __attribute__((address_space(1)))

__device__ __forceinline__ int ng_align(int sub, int aT) { // NOMA.c:464-475
    const int m = sub % aT;
    if (m == 0) return sub + 1;
    if (m == 1) return sub;
    return sub + (aT - m + 1);
}

// block-wide exclusive prefix of one int per thread (1024 threads); returns the block total through `tot`
__device__ __forceinline__ int block_excl_scan(const int v, int *wtot, int &tot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int x = wave_scan_incl(v); // (DPP: prach_device_fn.h)
    __syncthreads(); // (wtot of the previous scan has been read)
    if (lane == 63) wtot[w] = x;
    __syncthreads();
    int add = 0, t = 0;
    for (int k = 0; k < NW; k++) { const int c = wtot[k]; if (k < w) add += c; t += c; }
    tot = t;
    return x - v + add;
}

constexpr int NOMA_GLIBC_AMBIGUOUS = -77; // (internal) a value of the device-side activation / a gain comparison fell inside the device libm's error band

// One access slot (the slot's subframe `time` and the aT - 1 behind it).  live == nullptr: the passes run over every arrived UE [0, ac) as the reference's
// loops do.  Otherwise over live[0 .. nlive): the indices, ascending, of the arrived UEs that have neither succeeded nor been dropped for good — nothing
// ever touches the others again (every action below asks RA == 0 and RaFailed == 0), and an index-ordered prefix over the list is the prefix over all
// UEs; succ_removed: the successes among the UEs already taken off the list.  devact: the gains were computed by the device's math library (noma_glibc_trial_kernel) — every comparison of gains is then
// checked against the error band and reported (status NOMA_GLIBC_AMBIGUOUS: the host reruns the trial with its own libm).
__device__ __forceinline__ void noma_glibc_slot(NG NUe *const ue, const NG int *const stream, const NParams &K, int *cnt, int *who, int *wtot, int *sh,
                                                int *gs_idx, double *gs_g, double *gs_lg,
                                                const int time, const int ac, const NG int *const live, const int nlive, const int succ_removed, const unsigned long long slen, const bool devact,
                                                unsigned long long &pos_io, int &status_out, int &exit_time_out, int &nsucc_out) {
    const int tid = threadIdx.x;
    const int nP = K.nP, aT = K.aT;
    unsigned long long pos = pos_io;
    const int nsect = K.nonsector ? 1 : 6;
    const int nn = live ? nlive : ac; // UEs the passes look at: the live list (ascending indices) or every arrived UE
    auto idx_of = [&](const int q) -> int { return live ? live[q] : q; };

    // ---- transmitter gather (NOMA.c:207-213 / :331-339): per (sector, preamble) count and lowest transmitter ----
    for (int k = tid; k < 6 * 64; k += WG_THREADS) { cnt[k] = 0; who[k] = INT_MAX; }
    __syncthreads();
    for (int q = tid; q < nn; q += WG_THREADS) {
        const int i = idx_of(q);
        const NG NUe &u = ue[i];
        if (u.RA == 0 && u.txTime == time + 1 && u.msg2 == 0 && u.nowBackoff <= 0 && u.RaFailed == 0) {
            const int b = (K.nonsector ? 0 : u.sector) * nP + u.preamble;
            atomicAdd(&cnt[b], 1);
            atomicMin(&who[b], i);
        }
    }
    __syncthreads();
    // ---- grouping (NOMA.c:214-309 per sector / :341-437 cell-wide) by ONE wavefront, sectors in order because the pair draws come straight from the
    // stream: lane = preamble.  Singleton transmitters in preamble order (ballot compaction), stable ascending rank by gain == the reference's bubble
    // sort with strict < (NOMA.c:90-103), greedy pairing over the sorted lanes (ballot + find-first), leftovers while grants remain — prach_noma.hip's
    // resolver with the reference's stream instead of a counter.  (Before: one thread and the O(count^2) bubble sort on scratch arrays, ~1 ms per slot.)
    if (tid < 64) {
        const int lane = tid;
        int status = PRACH_OK;
        for (int s = 0; s < nsect && status == PRACH_OK; s++) {
            const bool single = lane < nP && cnt[s * nP + lane] == 1;
            const int myidx = single ? who[s * nP + lane] : -1;
            const unsigned long long sm = __ballot(single);
            const int count = __popcll(sm);
            if (count == 0) continue;
            if (count <= K.nGrantUL) { // NOMA.c:245-250 / :377-382: every one of them is granted
                if (single) ue[myidx].msg2 = 1;
                continue;
            }
            if (single) { const int q = __popcll(sm & lanemask_lt(lane)); gs_idx[q] = myidx; gs_g[q] = ue[myidx].gain; gs_lg[q] = ue[myidx].lgain; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int uidx = -1, rank = 0;
            double ug = 0, ulg = 0;
            bool ambiguous = false;
            if (lane < count) {
                uidx = gs_idx[lane]; ug = gs_g[lane]; ulg = gs_lg[lane];
                for (int j = 0; j < count; j++) {
                    const double gj = gs_g[j];
                    rank += (gj < ug || (gj == ug && j < lane)) ? 1 : 0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < count) { gs_idx[rank] = uidx; gs_lg[rank] = ulg; gs_g[rank] = ug; } // (every lane has read the unsorted gains: barrier above)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int cidx = -1;
            double clg = 0;
            if (lane < count) { cidx = gs_idx[lane]; clg = gs_lg[lane]; }
            // (the libm's gains could order two neighbours of the sorted order the other way: within the error band of the device's)
            if (devact && lane + 1 < count) { const double ga = gs_g[lane], gb = gs_g[lane + 1]; if (__dsub_rn(gb, ga) <= ACT_GAIN_ORDER_BAND * gb) ambiguous = true; }
            unsigned long long valid = count >= 64 ? ~0ull : ((1ull << count) - 1ull);
            int grants = 0;
            bool grantme = false;
            const double clg10 = __dmul_rn(10.0, clg); // (NOMA.c:272: 10 * log(gain), the same product on either side of the difference)
            const unsigned long long lows = count >= 2 ? ((1ull << (count - 1)) - 1ull) : 0ull; // i < count - 1
            unsigned long long above = ~0ull;                                                    // bits behind the last i looked at
            for (;;) { // NOMA.c:268-298 (enNoma stays 0: :266,269), over the still unpaired i in ascending order — as prach_noma.hip's resolver
                const unsigned long long rest = valid & lows & above;
                if (!rest) break;
                const int i = __ffsll((long long)rest) - 1;
                above = ~((2ull << i) - 1ull);
                const double lgi10 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(clg10), i), __builtin_amdgcn_readlane(__double2loint(clg10), i));
                const double diff = __dsub_rn(clg10, lgi10); // 10*log(high) - 10*log(low)
                if (devact && lane < count && fabs(__dsub_rn(diff, 15.0)) < 1e-9) ambiguous = true; // (any singleton this close to the threshold: at worst one more rerun with the host's table)
                const unsigned long long mj = __ballot(diff > 15.0) & valid & ~(1ull << i) & ~1ull;
                if (!mj) continue;
                const int j = __ffsll((long long)mj) - 1;
                valid &= ~((1ull << i) | (1ull << j));
                if (grants < K.nGrantUL) {
                    grants++;
                    if (pos >= slen) { status = PRACH_ERR_STREAM; break; }
                    const double pd = __ddiv_rn((double)stream[pos++], 2147483647.0);
                    int decoded = 2; // both
                    if (pd < 0.3) {
                        if (K.nonsector) decoded = 0; // rx[0], the weaker UE: NOMA.c:413-415
                        else {
                            if (pos >= slen) { status = PRACH_ERR_STREAM; break; }
                            decoded = stream[pos++] % 2; // NOMA.c:287-290: index into rx[] = {low, high}
                        }
                    }
                    if ((lane == i && (decoded == 2 || decoded == 0)) || (lane == j && (decoded == 2 || decoded == 1))) grantme = true;
                }
            }
            { // leftovers in sorted order while grants remain (NOMA.c:299-307)
                const bool left = lane < count && ((valid >> lane) & 1ull);
                const unsigned long long lm = __ballot(left);
                if (left && __popcll(lm & lanemask_lt(lane)) < K.nGrantUL - grants) grantme = true;
            }
            if (grantme && status == PRACH_OK) ue[cidx].msg2 = 1;
            if (__any(ambiguous) && status == PRACH_OK) status = NOMA_GLIBC_AMBIGUOUS;
        }
        if (tid == 0) { sh[0] = (int)(pos & 0xffffffffull); sh[1] = (int)(pos >> 32); sh[2] = status; }
    }
    __syncthreads();
    pos = ((unsigned long long)(unsigned)sh[1] << 32) | (unsigned)sh[0];
    int status = sh[2];
    __syncthreads();

    // ---- msg2Results in index order (NOMA.c:692-696 -> :449-498): draws at pos + index-ordered prefix ----
    const bool rar_expires = 5 >= K.maxRarWindow; // rarWindow is SET to the literal 5, then compared (NOMA.c:453-455)
    for (int c0 = 0; c0 < nn && status == PRACH_OK; c0 += WG_THREADS) {
        const bool in_ = c0 + tid < nn;
        const int i = in_ ? idx_of(c0 + tid) : 0;
        int need = 0;
        bool act = false;
        if (in_) {
            const NG NUe &u = ue[i];
            act = u.nowBackoff <= 0 && u.txTime == time + 1 && u.active == 1 && u.RA == 0 && u.RaFailed == 0;
            if (act && u.msg2 == 0 && rar_expires) need = (u.msg1ReTx + 1 >= K.maxMsg1ReTx) ? 2 : 1;
        }
        int tot;
        const int off = block_excl_scan(need, wtot, tot);
        if (pos + (unsigned long long)tot > slen) { status = PRACH_ERR_STREAM; break; }
        if (act) {
            NG NUe &u = ue[i];
            if (u.msg2 == 0) { // (active == 1 holds)
                u.rarWindow = 5;
                u.txTime += 3;
                if (rar_expires) {
                    u.nTxPreamble++;
                    u.rarWindow = 0;
                    u.msg1ReTx++;
                    const int tmp = stream[pos + (unsigned long long)off] % K.backoff;
                    u.txTime = ng_align(u.txTime + tmp, aT);
                    u.nowBackoff = u.txTime - (time + 1) - 1;
                    u.secondTxTime = u.txTime;
                    if (u.msg1ReTx >= K.maxMsg1ReTx) {
                        u.preamble = stream[pos + (unsigned long long)off + 1] % nP;
                        u.RaFailed++;
                        u.nTxPreamble = 0; u.rarWindow = 0; u.msg1ReTx = 0; u.timer = 0;
                    }
                }
            } else { // msg2 == 1 (NOMA.c:491-497)
                u.active = 2;
                u.txTime += 10;
                u.secondTxTime = u.txTime;
                u.msg3Wait = 0;
            }
        }
        pos += (unsigned long long)tot;
    }

    // ---- the slot's subframes: resourceRequestAllocation (NOMA.c:499-546), timerIncrease (:702-706), successUEs (:707-710) ----
    int exit_time = -1, nsucc = 0;
    for (int k = 0; k < aT && time + k < K.stop && status == PRACH_OK && exit_time < 0; k++) {
        const int tk = time + k;
        int succ_here = 0;
        for (int c0 = 0; c0 < nn && status == PRACH_OK; c0 += WG_THREADS) {
            const bool in_ = c0 + tid < nn;
            const int i = in_ ? idx_of(c0 + tid) : 0;
            int need = 0;
            bool due = false;
            if (in_) {
                const NG NUe &u = ue[i];
                due = u.txTime == tk && u.msg2 == 1 && u.active == 2 && u.RaFailed == 0;
                if (due) need = u.msg3Wait <= 48 ? 1 : 2;
            }
            int tot;
            const int off = block_excl_scan(need, wtot, tot);
            if (pos + (unsigned long long)tot > slen) { status = PRACH_ERR_STREAM; break; }
            if (in_) {
                NG NUe &u = ue[i];
                if (due) {
                    if (u.msg3Wait <= 48) {
                        const float p = (float)stream[pos + (unsigned long long)off] / (float)2147483647;
                        if (p > 0.1) { u.active = 0; u.RA = 1; u.timer = u.timer + 6; }
                        else { u.txTime += 49; u.msg3Wait = 49; }
                    } else {
                        u.RA = 0; u.msg3Faile++; u.active = 1; u.msg2 = 0;
                        u.preamble = stream[pos + (unsigned long long)off] % nP;
                        const int tmp = stream[pos + (unsigned long long)off + 1] % K.backoff;
                        u.txTime = ng_align(u.txTime + tmp, aT);
                        u.secondTxTime = u.txTime;
                        u.nowBackoff = u.txTime - tk - 1;
                        u.rarWindow = 0; u.nTxPreamble = 0; u.msg1ReTx = 0; u.timer = 0;
                    }
                }
                if (u.active > 0 && u.RA == 0 && u.RaFailed == 0) { u.timer++; if (u.nowBackoff > 0) u.nowBackoff--; }
                succ_here += u.RA == 1 ? 1 : 0;
            }
            pos += (unsigned long long)tot;
        }
        // successUEs counts over ALL nUE; a UE beyond activeCheck has not succeeded, so the count over the arrived ones decides
        int tot;
        (void)block_excl_scan(succ_here, wtot, tot);
        nsucc = tot + succ_removed;
        if (nsucc == K.nUE) exit_time = tk;
    }
    __syncthreads();
    pos_io = pos; status_out = status; exit_time_out = exit_time; nsucc_out = nsucc;
}

// The UEs that succeeded or were dropped in this slot leave the list of live UEs: stable compaction in place, 1024 entries at a time (an entry only ever
// moves down, and the barriers of the block-wide prefix separate a block's reads from the writes of the next).  Returns the new length.
__device__ __forceinline__ int noma_live_compact(NG NUe *const ue, NG int *const live, const int nlive, int *wtot, int &succ_removed) {
    const int tid = threadIdx.x;
    int out = 0;
    for (int c0 = 0; c0 < nlive; c0 += WG_THREADS) {
        const bool in_ = c0 + tid < nlive;
        const int i = in_ ? live[c0 + tid] : 0;
        int keep = 0, ra = 0;
        if (in_) { const NG NUe &u = ue[i]; ra = u.RA == 1 ? 1 : 0; keep = (ra || u.RaFailed != 0) ? 0 : 1; }
        int tot;
        const int excl = block_excl_scan(keep, wtot, tot);
        succ_removed += __syncthreads_count(in_ && ra);
        if (keep) live[out + excl] = i;
        out += tot;
    }
    return out;
}

// one access slot per launch: the arrivals of the slot have been activated by the host (prach_noma_activation_stream) before it
struct LiveState { int nlive, succ_removed, prevAC, pad; }; // (device memory, zeroed before the first slot, kept from launch to launch)

__global__ __launch_bounds__(WG_THREADS) void noma_glibc_slot_kernel(NUe *ue_, const int *stream_, SlotCtl *ctl_, int *live_, LiveState *ls_, const NParams K) {
    __shared__ int cnt[6 * 64], who[6 * 64], wtot[NW], sh[8], gs_idx[64];
    __shared__ double gs_g[64], gs_lg[64];
    NG SlotCtl *const ctl = (NG SlotCtl *)ctl_;
    NG NUe *const ue = (NG NUe *)ue_;
    NG int *const live = (NG int *)live_;
    NG LiveState *const ls = (NG LiveState *)ls_;
    const int tid = threadIdx.x, ac = ctl->activeCheck;
    int nlive = ls->nlive, succ_removed = ls->succ_removed;
    const int prevAC = ls->prevAC;
    unsigned long long pos = ctl->pos;
    __syncthreads(); // (everybody has read the state thread 0 rewrites below)
    for (int i = prevAC + tid; i < ac; i += WG_THREADS) live[nlive + (i - prevAC)] = i; // the slot's arrivals (activated by the host) join the list
    nlive += ac - prevAC;
    __syncthreads();
    int status = PRACH_OK, exit_time = -1, nsucc = 0;
    noma_glibc_slot(ue, (const NG int *)stream_, K, cnt, who, wtot, sh, gs_idx, gs_g, gs_lg, ctl->time, ac, live, nlive, succ_removed, ctl->stream_len, false, pos, status, exit_time, nsucc);
    if (status == PRACH_OK && exit_time < 0) nlive = noma_live_compact(ue, live, nlive, wtot, succ_removed);
    if (tid == 0) {
        ctl->pos = pos;
        ctl->nSuccess = nsucc;
        ctl->exit_time = exit_time;
        ctl->status = status;
        ls->nlive = nlive; ls->succ_removed = succ_removed; ls->prevAC = ac;
    }
}

// The WHOLE trial in one launch: activeUE (NOMA.c:131-192) runs on the device too.  The arrivals of a slot take their draws from the stream in index
// order and each takes a data-dependent number of them (two rejection loops), so their stream positions are found by iteration: every arrival
// assumes a start, runs activeUE from there and reports how many values it took; a block-wide prefix gives the starts these counts imply; until
// no start moves (the first arrival's start is right from the beginning, so every round fixes at least one more: two or three rounds in practice).
// Gains come from the device's math library: prach_noma_act.h flags every value inside its error band, the slot's comparisons do the same —
// either ends the trial with NOMA_GLIBC_AMBIGUOUS and the host runs it again slot by slot with its own libm.
struct TrialCtl { unsigned long long pos, steps; int nSuccess, time_exit, status, activeCheck; };
struct TrialArgs { NUe *ue; const int *stream; const int *sched; int *live; TrialCtl *ctl; unsigned long long slen; float cell_radius; int pad; NParams K; };

// (one workgroup per trial: the trials of a call — the seeds of one sweep point — run side by side)
__global__ __launch_bounds__(WG_THREADS) void noma_glibc_trial_kernel(const TrialArgs *__restrict__ args) {
    __shared__ int cnt[6 * 64], who[6 * 64], wtot[NW], sh[8], gs_idx[64];
    __shared__ double gs_g[64], gs_lg[64];
    const TrialArgs &A = args[blockIdx.x];
    const NParams K = A.K;
    const unsigned long long slen = A.slen;
    const float cell_radius = A.cell_radius;
    TrialCtl *const ctl_ = A.ctl;
    NG NUe *const ue = (NG NUe *)A.ue;
    const NG int *const stream = (const NG int *)A.stream;
    const NG int *const sched = (const NG int *)A.sched;
    NG int *const live = (NG int *)A.live; // the arrived UEs that have neither succeeded nor been dropped, ascending (see noma_glibc_slot)
    const int tid = threadIdx.x, aT = K.aT;
    for (int i = tid; i < K.nUE; i += WG_THREADS) { // calloc + initUserInfo (NOMA.c:651-655): everything 0, sector -1
        NG int *const p = (NG int *)&ue[i];
#pragma unroll
        for (int k = 0; k < (int)(sizeof(NUe) / 4); k++) p[k] = 0;
        ue[i].sector = -1;
    }
    __syncthreads();
    unsigned long long pos = 0, steps = 0;
    int activeCheck = 0, nlive = 0, succ_removed = 0, status = PRACH_OK, nsucc = 0, time_exit = K.stop;
    for (int s = 0, t = 0; t < K.stop && status == PRACH_OK; s++, t += aT) {
        const int prevAC = activeCheck;
        activeCheck = sched[s]; // NOMA.c:675-681
        for (int c0 = prevAC; c0 < activeCheck && status == PRACH_OK; c0 += WG_THREADS) { // NOMA.c:682-686, 1024 arrivals at a time
            const int i = c0 + tid;
            const bool v = i < activeCheck;
            unsigned long long start = pos + 4ull * (unsigned long long)tid; // (most UEs take four values)
            ActUe o{0, 0, 0.0, 0.0, false};
            int used = 0, total = 0;
            bool oob = false;
            for (int round = 0;; round++) {
                used = 0; oob = false;
                if (v) o = noma_active_ue([&]() -> int {
                        const unsigned long long q = start + (unsigned long long)used++;
                        if (q < slen) return stream[q];
                        oob = true; // (beyond the window: the trial ends with PRACH_ERR_STREAM below; any value that lets the loops end will do)
                        return 2147483647 / 3;
                    }, K.nP, cell_radius);
                int tot;
                const int excl = block_excl_scan(v ? used : 0, wtot, tot);
                const unsigned long long ns = pos + (unsigned long long)excl;
                const int moved = __syncthreads_or(v && ns != start);
                start = ns;
                total = tot;
                if (!moved) break;
                if (round > 1100) { status = PRACH_ERR_INTERNAL; break; } // (cannot happen: every round fixes at least one more start)
            }
            if (__syncthreads_or(v && oob)) status = PRACH_ERR_STREAM;
            else if (__syncthreads_or(v && o.flag)) status = NOMA_GLIBC_AMBIGUOUS;
            if (v && status == PRACH_OK) {
                NG NUe &u = ue[i];
                u.active = 1; u.preamble = o.pre; u.nTxPreamble = 1; u.txTime = t + 1; u.timer = 0; u.rarWindow = 0; u.msg1ReTx = 0; u.nowBackoff = 0;
                u.firstTxTime = t + 1; u.sector = o.sec; u.gain = o.gain; u.lgain = o.lgain;
                live[nlive + (i - prevAC)] = i; // (arrivals join the list in index order, behind everybody who arrived before)
            }
            pos += (unsigned long long)total;
        }
        nlive += activeCheck - prevAC;
        __syncthreads(); // the activated records are in memory
        if (status != PRACH_OK) break;
        int st = PRACH_OK, ex = -1;
        noma_glibc_slot(ue, stream, K, cnt, who, wtot, sh, gs_idx, gs_g, gs_lg, t, activeCheck, live, nlive, succ_removed, slen, true, pos, st, ex, nsucc);
        status = st;
        steps = (unsigned long long)min(t + aT, K.stop);
        if (status != PRACH_OK) break;
        if (ex >= 0) { time_exit = ex; steps = (unsigned long long)ex + 1ull; break; }
        nlive = noma_live_compact(ue, live, nlive, wtot, succ_removed);
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) {
        NG TrialCtl *const ctl = (NG TrialCtl *)ctl_;
        ctl->pos = pos; ctl->steps = steps; ctl->nSuccess = nsucc; ctl->time_exit = time_exit; ctl->status = status; ctl->activeCheck = activeCheck;
    }
}

} // namespace

// saveResult (NOMA.c:618-625) and the logged fields from the final UE records
static void noma_glibc_finish(const prach_cfg &c, const NUe *hue, unsigned long long pos, int nSuccess, int time_exit, unsigned long long steps, int activeCheck,
                              int32_t nAccess, prach_result *res, prach_ue_log *logs) {
    const int nUE = c.nUE;
    long long delay = 0;
    int nTxP = 0, failed = 0;
    for (int i = 0; i < nUE; i++) {
        const NUe &u = hue[i];
        if (u.RA == 1) { delay += u.timer; nTxP += u.nTxPreamble; }
        if (u.RaFailed) failed++;
        if (logs) { // (the field mapping of prach_noma.hip's dump)
            prach_ue_log &o = logs[i];
            o.idx = i; o.timer = u.timer; o.active = u.active; o.txTime = u.txTime; o.firstTxTime = u.firstTxTime; o.secondTxTime = u.secondTxTime;
            o.nowBackoff = u.nowBackoff; o.preamble = u.preamble; o.preambleChange = u.sector; o.rarWindow = u.rarWindow; o.maxRarCounter = u.msg1ReTx;
            o.preambleTxCounter = u.nTxPreamble; o.msg2Flag = u.msg2; o.connectionRequest = u.msg3Wait; o.msg4Flag = u.RA;
            o.failCount = u.RaFailed | (u.msg3Faile << 16);
        }
    }
    std::memset(res, 0, sizeof(*res));
    res->status = PRACH_OK;
    res->time_exit = time_exit;
    res->maxTime = 10000;
    res->nSuccessUE = nSuccess;
    res->failedUEs = nUE - nSuccess;
    res->preambleTxCount = nTxP;
    res->failCounts = failed;
    res->activeCheck = activeCheck;
    res->nAccessUE = nAccess;
    res->finalSuccessUEs = nSuccess;
    res->sumTimer = delay;
    res->totalDelay = (float)delay;
    res->draws = pos;
    res->steps = steps;
}

// One NOMA_C trial in the reference's rand() stream.  `hstream`: the host copy of the window [stream_offset, +len) (prach_glibc_stream).
// Returns PRACH_OK, PRACH_ERR_STREAM (window too small: the caller retries with a larger one) or a device error.
// This is the slot-by-slot form (the arrivals activated by the host between the launches); run_noma_glibc_batch below is the single-launch form.
int run_noma_glibc_trial(hipStream_t stream, const prach_cfg &c, const int32_t *hstream, unsigned long long len, prach_result *res, prach_ue_log *logs,
                         double *kernel_ms) {
#define NHIP(expr)                                                                                                            \
    do {                                                                                                                      \
        hipError_t e_ = (expr);                                                                                               \
        if (e_ != hipSuccess) {                                                                                               \
            std::fprintf(stderr, "[prach] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, __LINE__, hipGetErrorString(e_)); \
            rc = PRACH_ERR_DEVICE;                                                                                            \
            goto done;                                                                                                        \
        }                                                                                                                     \
    } while (0)
    int rc = PRACH_OK;
    const int nUE = c.nUE, aT = c.accessTime, maxTime = 10000;
    const int stop = (c.max_steps > 0 && c.max_steps < maxTime) ? c.max_steps : maxTime;
    NUe *d_ue = nullptr;
    int *d_stream = nullptr;
    SlotCtl *d_ctl = nullptr, *h_ctl = nullptr;
    int *d_live = nullptr;
    LiveState *d_ls = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<NUe> hue((size_t)nUE);
    std::vector<int32_t> sched((size_t)(maxTime / aT + 2), nUE);
    int32_t nAccess = 0;
    NParams K{nUE, c.nPreamble, c.backoff, c.nGrantUL, c.maxRarWindow, c.maxMsg2TxCount, aT, stop, (c.flags & PRACH_FLAG_NOMA_NONSECTOR) ? 1 : 0};
    unsigned long long pos = 0;
    int activeCheck = 0, time_exit = stop, nSuccess = 0;
    unsigned long long steps = 0;
    prach_arrival_schedule(&c, sched.data(), (int)sched.size(), &nAccess);
    std::memset(hue.data(), 0, sizeof(NUe) * (size_t)nUE);
    for (int i = 0; i < nUE; i++) hue[i].sector = -1; // NOMA.c:651-655: everything 0, sector -1
    NHIP(hipMalloc(reinterpret_cast<void **>(&d_ue), sizeof(NUe) * (size_t)nUE));
    NHIP(hipMalloc(reinterpret_cast<void **>(&d_stream), 4 * (size_t)(len + 2)));
    NHIP(hipMalloc(reinterpret_cast<void **>(&d_ctl), sizeof(SlotCtl)));
    NHIP(hipMalloc(reinterpret_cast<void **>(&d_live), 4 * (size_t)nUE + 64));
    NHIP(hipMalloc(reinterpret_cast<void **>(&d_ls), sizeof(LiveState)));
    NHIP(hipMemsetAsync(d_ls, 0, sizeof(LiveState), stream));
    NHIP(hipHostMalloc(reinterpret_cast<void **>(&h_ctl), sizeof(SlotCtl), hipHostMallocDefault));
    NHIP(hipEventCreate(&ev0));
    NHIP(hipEventCreate(&ev1));
    NHIP(hipMemcpyAsync(d_ue, hue.data(), sizeof(NUe) * (size_t)nUE, hipMemcpyHostToDevice, stream));
    NHIP(hipMemcpyAsync(d_stream, hstream, 4 * (size_t)len, hipMemcpyHostToDevice, stream));
    NHIP(hipEventRecord(ev0, stream));
    for (int s = 0, t = 0; t < stop; s++, t += aT) {
        // NOMA.c:675-686: this access slot's arrivals, activated on the host in stream order (index order)
        const int prevAC = activeCheck;
        activeCheck = sched[s];
        for (int i = prevAC; i < activeCheck; i++) {
            NUe &u = hue[i];
            int32_t pre0, sec;
            const int arc = prach_noma_activation_stream(&c, hstream, reinterpret_cast<uint64_t *>(&pos), len, &pre0, &sec, &u.gain, &u.lgain);
            if (arc != PRACH_OK) { rc = arc; goto done; }
            u.active = 1; u.preamble = pre0; u.nTxPreamble = 1; u.txTime = t + 1; u.timer = 0; u.rarWindow = 0; u.msg1ReTx = 0; u.nowBackoff = 0;
            u.firstTxTime = t + 1; u.sector = sec;
        }
        if (activeCheck > prevAC)
            NHIP(hipMemcpyAsync(d_ue + prevAC, hue.data() + prevAC, sizeof(NUe) * (size_t)(activeCheck - prevAC), hipMemcpyHostToDevice, stream));
        h_ctl->pos = pos; h_ctl->stream_len = len; h_ctl->time = t; h_ctl->activeCheck = activeCheck; h_ctl->nSuccess = 0; h_ctl->exit_time = -1;
        h_ctl->status = PRACH_OK; h_ctl->pad = 0;
        NHIP(hipMemcpyAsync(d_ctl, h_ctl, sizeof(SlotCtl), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(noma_glibc_slot_kernel, dim3(1), dim3(WG_THREADS), 0, stream, d_ue, d_stream, d_ctl, d_live, d_ls, K);
        NHIP(hipGetLastError());
        NHIP(hipMemcpyAsync(h_ctl, d_ctl, sizeof(SlotCtl), hipMemcpyDeviceToHost, stream));
        NHIP(hipStreamSynchronize(stream));
        if (h_ctl->status != PRACH_OK) { rc = h_ctl->status; goto done; }
        pos = h_ctl->pos;
        nSuccess = h_ctl->nSuccess;
        if (h_ctl->exit_time >= 0) { time_exit = h_ctl->exit_time; steps = (unsigned long long)h_ctl->exit_time + 1; break; }
        steps = (unsigned long long)std::min(t + aT, stop);
    }
    NHIP(hipEventRecord(ev1, stream));
    NHIP(hipMemcpyAsync(hue.data(), d_ue, sizeof(NUe) * (size_t)nUE, hipMemcpyDeviceToHost, stream));
    NHIP(hipStreamSynchronize(stream));
    {
        float ms = 0;
        NHIP(hipEventElapsedTime(&ms, ev0, ev1));
        if (kernel_ms) *kernel_ms += ms;
        noma_glibc_finish(c, hue.data(), pos, nSuccess, time_exit, steps, activeCheck, nAccess, res, logs);
    }
done:
    if (d_ue) (void)hipFree(d_ue);
    if (d_stream) (void)hipFree(d_stream);
    if (d_ctl) (void)hipFree(d_ctl);
    if (d_live) (void)hipFree(d_live);
    if (d_ls) (void)hipFree(d_ls);
    if (h_ctl) (void)hipHostFree(h_ctl);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    return rc;
#undef NHIP
}

// The single-launch form for the NOMA_C trials of a call (the seeds of one sweep point): one workgroup per trial in ONE launch, the stream windows
// [stream_offset, + lens[j]) generated on the device before it.  rcs[j]: PRACH_OK (res[j] / logs[j] filled), PRACH_ERR_STREAM (window too small),
// NOMA_GLIBC_AMBIGUOUS_RC (a value inside the device math library's error band: run that trial slot by slot, run_noma_glibc_trial) or a device error.
int run_noma_glibc_batch(hipStream_t stream, const prach_cfg *const *cfgs, int n, const unsigned long long *lens, prach_result *const *res, prach_ue_log *const *logs,
                         double *kernel_ms, int *rcs) {
#define BHIP(expr)                                                                                                            \
    do {                                                                                                                      \
        hipError_t e_ = (expr);                                                                                               \
        if (e_ != hipSuccess) {                                                                                               \
            std::fprintf(stderr, "[prach] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, __LINE__, hipGetErrorString(e_)); \
            rc = PRACH_ERR_DEVICE;                                                                                            \
            goto done;                                                                                                        \
        }                                                                                                                     \
    } while (0)
    int rc = PRACH_OK;
    const int maxTime = 10000;
    char *dbuf = nullptr, *hbuf = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    struct Off { size_t seeds, sched, ue, stream, live; size_t nsched, nchunks; int32_t nAccess; };
    std::vector<Off> off((size_t)n);
    std::vector<NUe> hue;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    // staged prefix (one copy): argument blocks, control blocks (zeroed), stream seeds, arrival tables; then the per-trial arrays
    size_t o = up(sizeof(TrialArgs) * (size_t)n);
    const size_t octl = o; o = up(o + sizeof(TrialCtl) * (size_t)n);
    const size_t ojobs = o; o = up(o + sizeof(StreamJob) * (size_t)n);
    for (int j = 0; j < n; j++) {
        const prach_cfg &c = *cfgs[j];
        off[j].nsched = (size_t)(maxTime / c.accessTime + 2);
        off[j].nchunks = (size_t)((lens[j] + STREAM_CHUNK - 1) / STREAM_CHUNK);
        off[j].seeds = o; o = up(o + 4 * 31 * (off[j].nchunks + 1));
        off[j].sched = o; o = up(o + 4 * off[j].nsched);
    }
    const size_t staged = o;
    for (int j = 0; j < n; j++) {
        const prach_cfg &c = *cfgs[j];
        off[j].ue = o; o = up(o + sizeof(NUe) * (size_t)c.nUE);
        off[j].stream = o; o = up(o + 4 * (size_t)(lens[j] + 2));
        off[j].live = o; o = up(o + 4 * (size_t)c.nUE + 64);
    }
    BHIP(hipMalloc(reinterpret_cast<void **>(&dbuf), o));
    BHIP(hipHostMalloc(reinterpret_cast<void **>(&hbuf), staged, hipHostMallocDefault));
    BHIP(hipEventCreate(&ev0));
    BHIP(hipEventCreate(&ev1));
    std::memset(hbuf, 0, staged);
    for (int j = 0; j < n; j++) {
        const prach_cfg &c = *cfgs[j];
        const int stop = (c.max_steps > 0 && c.max_steps < maxTime) ? c.max_steps : maxTime;
        int32_t *sched = reinterpret_cast<int32_t *>(hbuf + off[j].sched);
        for (size_t q = 0; q < off[j].nsched; q++) sched[q] = c.nUE;
        prach_arrival_schedule(&c, sched, (int)off[j].nsched, &off[j].nAccess);
        prach_internal_glibc_seeds((uint32_t)c.seed, c.stream_offset, off[j].nchunks, STREAM_CHUNK, reinterpret_cast<uint32_t *>(hbuf + off[j].seeds));
        TrialArgs &A = reinterpret_cast<TrialArgs *>(hbuf)[j];
        A.ue = reinterpret_cast<NUe *>(dbuf + off[j].ue); A.stream = reinterpret_cast<const int *>(dbuf + off[j].stream);
        A.sched = reinterpret_cast<const int *>(dbuf + off[j].sched); A.live = reinterpret_cast<int *>(dbuf + off[j].live);
        A.ctl = reinterpret_cast<TrialCtl *>(dbuf + octl) + j; A.slen = lens[j]; A.cell_radius = c.cellRadius; A.pad = 0;
        A.K = NParams{c.nUE, c.nPreamble, c.backoff, c.nGrantUL, c.maxRarWindow, c.maxMsg2TxCount, c.accessTime, stop, (c.flags & PRACH_FLAG_NOMA_NONSECTOR) ? 1 : 0};
        reinterpret_cast<StreamJob *>(hbuf + ojobs)[j] = StreamJob{reinterpret_cast<const unsigned *>(dbuf + off[j].seeds), reinterpret_cast<int *>(dbuf + off[j].stream), lens[j]};
        rcs[j] = PRACH_ERR_INTERNAL;
    }
    BHIP(hipMemcpyAsync(dbuf, hbuf, staged, hipMemcpyHostToDevice, stream));
    BHIP(hipEventRecord(ev0, stream));
    {
        unsigned long long max_n = 0;
        for (int j = 0; j < n; j++) max_n = std::max(max_n, lens[j]);
        BHIP(launch_glibc_stream_jobs(reinterpret_cast<const StreamJob *>(dbuf + ojobs), n, max_n, stream)); // every window, one launch
    }
    hipLaunchKernelGGL(noma_glibc_trial_kernel, dim3((unsigned)n), dim3(WG_THREADS), 0, stream, reinterpret_cast<const TrialArgs *>(dbuf));
    BHIP(hipGetLastError());
    BHIP(hipEventRecord(ev1, stream));
    BHIP(hipMemcpyAsync(hbuf + octl, dbuf + octl, sizeof(TrialCtl) * (size_t)n, hipMemcpyDeviceToHost, stream));
    BHIP(hipStreamSynchronize(stream));
    {
        float ms = 0;
        BHIP(hipEventElapsedTime(&ms, ev0, ev1));
        if (kernel_ms) *kernel_ms += ms;
    }
    for (int j = 0; j < n; j++) {
        const prach_cfg &c = *cfgs[j];
        const TrialCtl tc = reinterpret_cast<const TrialCtl *>(hbuf + octl)[j];
        if (tc.status == NOMA_GLIBC_AMBIGUOUS) { rcs[j] = NOMA_GLIBC_AMBIGUOUS_RC; continue; }
        if (tc.status != PRACH_OK) { rcs[j] = tc.status; continue; }
        hue.resize((size_t)c.nUE);
        BHIP(hipMemcpy(hue.data(), dbuf + off[j].ue, sizeof(NUe) * (size_t)c.nUE, hipMemcpyDeviceToHost));
        noma_glibc_finish(c, hue.data(), tc.pos, tc.nSuccess, tc.time_exit, tc.steps, tc.activeCheck, off[j].nAccess, res[j], logs ? logs[j] : nullptr);
        rcs[j] = PRACH_OK;
    }
done:
    if (dbuf) (void)hipFree(dbuf);
    if (hbuf) (void)hipHostFree(hbuf);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    return rc;
#undef BHIP
}

} // namespace prach
