// prach_noma_glibc.hip — NOMA.c (PRACH_VARIANT_NOMA_C) in the REFERENCE'S OWN rand() stream, on the GPU.
//
// Why this path is separate from prach_noma.hip: NOMA.c draws everything from one global stream, and activeUE (NOMA.c:131-192) consumes a
// DATA-DEPENDENT number of draws per UE (two rejection loops) through double-precision libm calls (cos, sin, sqrt, log, pow) whose results
// must be bit-identical to the reference's.  So in this mode the activation of an access slot's arrivals runs on the HOST, in stream order,
// with the libm the reference links (prach_noma_activation_stream), between two device steps; everything else of the slot — the
// transmitter gather, the per-sector grouping with its pair draws, msg2Results, and resourceRequestAllocation / timerIncrease /
// successUEs of the slot's accessTime subframes — is ONE kernel launch per access slot that consumes the device copy of the stream at
// exactly the positions the reference's index-ordered loops reach (block-wide exclusive prefix sums over the draw counts, chunk after
// chunk in index order, phase after phase in the reference's order).  One workgroup per trial: this is the bit-exact-vs-the-reference's-own-
// files mode of config 4 (2000 launches and host round trips per trial: ~0.3 s at nUE = 100 000), not the throughput mode (Philox,
// prach_noma.hip).  Reference: NOMA.c:131-192 (host), :194-324 / :325-447 (grouping), :449-498 (msg2Results), :499-546
// (resourceRequestAllocation), :665-711 (the time step).
#include "prach_device.h"
#include "prach_device_fn.h"

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

__global__ __launch_bounds__(WG_THREADS) void noma_glibc_slot_kernel(NUe *ue_, const int *stream_, SlotCtl *ctl_, const NParams K) {
    __shared__ int cnt[6 * 64], who[6 * 64], wtot[NW], sh[8];
    NG NUe *const ue = (NG NUe *)ue_;
    const NG int *const stream = (const NG int *)stream_;
    NG SlotCtl *const ctl = (NG SlotCtl *)ctl_;
    const int tid = threadIdx.x;
    const int nP = K.nP, aT = K.aT, time = ctl->time, ac = ctl->activeCheck;
    const unsigned long long slen = ctl->stream_len;
    unsigned long long pos = ctl->pos;
    const int nsect = K.nonsector ? 1 : 6;

    // ---- transmitter gather (NOMA.c:207-213 / :331-339): per (sector, preamble) count and lowest transmitter ----
    for (int k = tid; k < 6 * 64; k += WG_THREADS) { cnt[k] = 0; who[k] = INT_MAX; }
    __syncthreads();
    for (int i = tid; i < ac; i += WG_THREADS) {
        const NG NUe &u = ue[i];
        if (u.RA == 0 && u.txTime == time + 1 && u.msg2 == 0 && u.nowBackoff <= 0 && u.RaFailed == 0) {
            const int b = (K.nonsector ? 0 : u.sector) * nP + u.preamble;
            atomicAdd(&cnt[b], 1);
            atomicMin(&who[b], i);
        }
    }
    __syncthreads();
    // ---- grouping, sequentially like the reference (NOMA.c:214-309 per sector / :341-437 cell-wide): ONE thread; sectors in order, the
    // pair draws straight from the stream ----
    if (tid == 0) {
        int status = PRACH_OK;
        for (int s = 0; s < nsect; s++) {
            int tidx[64];
            double tg[64], tl[64];
            int count = 0;
            for (int p = 0; p < nP; p++)
                if (cnt[s * nP + p] == 1) { const int i = who[s * nP + p]; tidx[count] = i; tg[count] = ue[i].gain; tl[count] = ue[i].lgain; count++; }
            if (count <= 0) continue;
            int grantCheck = 0;
            if (count <= K.nGrantUL) {
                for (int i = 0; i < count; i++) {
                    if (K.nonsector) { if (grantCheck < K.nGrantUL) grantCheck++; ue[tidx[i]].msg2 = 1; } // NOMA.c:377-382
                    else if (grantCheck < K.nGrantUL) { grantCheck++; ue[tidx[i]].msg2 = 1; }             // NOMA.c:245-250
                }
                continue;
            }
            for (int i = 0; i < count; i++) // sortUE: bubble sort, strict < (stable), NOMA.c:90-103
                for (int j = 0; j < count - 1; j++)
                    if (tg[j + 1] < tg[j]) {
                        const int ti = tidx[j]; tidx[j] = tidx[j + 1]; tidx[j + 1] = ti;
                        const double a = tg[j]; tg[j] = tg[j + 1]; tg[j + 1] = a;
                        const double b = tl[j]; tl[j] = tl[j + 1]; tl[j + 1] = b;
                    }
            int pair = 0;
            for (int i = 0; i < count - 1; i++) {
                for (int j = 1; j < count; j++) { // (enNoma stays 0: NOMA.c:266,269)
                    const int rx0 = tidx[i], rx1 = tidx[j];
                    if (rx0 != -1 && rx1 != -1 && __dsub_rn(__dmul_rn(10.0, tl[j]), __dmul_rn(10.0, tl[i])) > 15.0) {
                        pair += 2;
                        tidx[i] = -1; tidx[j] = -1;
                        if (grantCheck < K.nGrantUL) {
                            grantCheck++;
                            if (pos >= slen) { status = PRACH_ERR_STREAM; break; }
                            const double pd = (double)stream[pos++] / (double)2147483647;
                            if (pd < 0.3) {
                                if (K.nonsector) ue[rx0].msg2 = 1; // NOMA.c:413-415
                                else {
                                    if (pos >= slen) { status = PRACH_ERR_STREAM; break; }
                                    const int which = stream[pos++] % 2; // NOMA.c:287-290
                                    ue[which ? rx1 : rx0].msg2 = 1;
                                }
                            } else { ue[rx0].msg2 = 1; ue[rx1].msg2 = 1; }
                        }
                        break;
                    }
                }
                if (status != PRACH_OK) break;
            }
            if (status != PRACH_OK) break;
            if (count - pair > 0)
                for (int i = 0; i < count; i++)
                    if (tidx[i] != -1 && grantCheck < K.nGrantUL) { grantCheck++; ue[tidx[i]].msg2 = 1; }
        }
        sh[0] = (int)(pos & 0xffffffffull); sh[1] = (int)(pos >> 32); sh[2] = status;
    }
    __syncthreads();
    pos = ((unsigned long long)(unsigned)sh[1] << 32) | (unsigned)sh[0];
    int status = sh[2];
    __syncthreads();

    // ---- msg2Results in index order (NOMA.c:692-696 -> :449-498): draws at pos + index-ordered prefix ----
    const bool rar_expires = 5 >= K.maxRarWindow; // rarWindow is SET to the literal 5, then compared (NOMA.c:453-455)
    for (int c0 = 0; c0 < ac && status == PRACH_OK; c0 += WG_THREADS) {
        const int i = c0 + tid;
        int need = 0;
        bool act = false;
        if (i < ac) {
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
        for (int c0 = 0; c0 < ac && status == PRACH_OK; c0 += WG_THREADS) {
            const int i = c0 + tid;
            int need = 0;
            bool due = false;
            if (i < ac) {
                const NG NUe &u = ue[i];
                due = u.txTime == tk && u.msg2 == 1 && u.active == 2 && u.RaFailed == 0;
                if (due) need = u.msg3Wait <= 48 ? 1 : 2;
            }
            int tot;
            const int off = block_excl_scan(need, wtot, tot);
            if (pos + (unsigned long long)tot > slen) { status = PRACH_ERR_STREAM; break; }
            if (i < ac) {
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
        nsucc = tot;
        if (nsucc == K.nUE) exit_time = tk;
    }
    __syncthreads();
    if (tid == 0) {
        ctl->pos = pos;
        ctl->nSuccess = nsucc;
        ctl->exit_time = exit_time;
        ctl->status = status;
    }
}

} // namespace

// One NOMA_C trial in the reference's rand() stream.  `hstream`: the host copy of the window [stream_offset, +len) (prach_glibc_stream).
// Returns PRACH_OK, PRACH_ERR_STREAM (window too small: the caller retries with a larger one) or a device error.
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
        hipLaunchKernelGGL(noma_glibc_slot_kernel, dim3(1), dim3(WG_THREADS), 0, stream, d_ue, d_stream, d_ctl, K);
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
        long long delay = 0;
        int nTxP = 0, failed = 0;
        for (int i = 0; i < nUE; i++) { // saveResult, NOMA.c:618-625
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
        res->maxTime = maxTime;
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
done:
    if (d_ue) (void)hipFree(d_ue);
    if (d_stream) (void)hipFree(d_stream);
    if (d_ctl) (void)hipFree(d_ctl);
    if (h_ctl) (void)hipHostFree(h_ctl);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    return rc;
#undef NHIP
}

} // namespace prach
