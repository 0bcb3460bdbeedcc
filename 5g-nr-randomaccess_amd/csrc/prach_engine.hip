// prach_engine.hip — host side of libprach_hip.so: the engine behind the C ABI in include/prach.h.
// Owns the HIP stream, one device arena (grown on demand, never allocated per subframe), stages
// the per-trial parameter blocks / arrival tables / draw streams, launches the trial kernel and
// turns DevResult into prach_result.  There is NO CPU fallback: without a gfx950 device every
// entry point that simulates returns PRACH_ERR_DEVICE.
#include "prach_device.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits.h>
#include <vector>

using namespace prach;

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            std::fprintf(stderr, "[prach] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, \
                         __LINE__, hipGetErrorString(e_));                                            \
            return PRACH_ERR_DEVICE;                                                                  \
        }                                                                                             \
    } while (0)

struct prach_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    char *arena = nullptr;
    size_t arena_cap = 0;
    prach_timing last{};
    int64_t opt_stream_factor = 0; // glibc: initial draws-per-UE budget override (0 = auto)
    int64_t opt_cluster = 0;       // workgroups per trial for the Philox cluster kernel (0 = auto)
    int64_t opt_legacy = 0;        // 1: run Philox trials on the one-workgroup trial_kernel as well
    int64_t opt_dense = 0;         // 1: cluster kernel without the compacted pass (diagnostic)
    int64_t opt_wide_records = 0;  // 1: 16-byte records also with one workgroup per trial (diagnostic)
    int64_t opt_pipeline = 1;      // 0: clusters do not run phase A ahead of the exchange (diagnostic)
    int last_G = 0;
    int num_cus = 256;            // co-residency budget of the cluster kernels: one 1024-thread workgroup per CU
};

namespace {

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct TrialLayout {
    size_t rec, ptc, ftt, stt, fcnt, nd, evbuf, evbuf2, sidx, sched, stream, logs, timers, out, mbox, bar, cand, end;
    size_t n_pre0, n_sector, n_gain, n_lgain, n_nd0;
    size_t seeds, nchunks; // glibc: one 31-word window per STREAM_CHUNK outputs (device-side generation)
    size_t stream_len, sched_len;
    int evw, mbstride;
};

TrialLayout layout_trial(const prach_cfg &c, size_t base, bool want_logs, size_t stream_len, int G) {
    TrialLayout L{};
    size_t o = base;
    const size_t n = (size_t)c.nUE;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    L.rec = take(16 * n);
    L.ptc = take(4 * n); L.ftt = take(4 * n); L.stt = take(4 * n); L.fcnt = take(4 * n); L.nd = take(4 * n);
    L.evbuf = take(sizeof(Event) * n); L.evbuf2 = take(sizeof(Event) * n);
    L.sidx = take(4 * (n + 256));
    L.sched_len = (size_t)(prach_max_time(&c) / c.accessTime + 2);
    L.sched = take(4 * L.sched_len);
    L.stream_len = stream_len;
    L.stream = take(4 * (stream_len + 2));
    L.nchunks = (stream_len + STREAM_CHUNK - 1) / STREAM_CHUNK;
    L.seeds = take(4 * 31 * (L.nchunks + 1));
    L.logs = want_logs ? take(sizeof(prach_ue_log) * n) : 0;
    L.timers = take(4 * n);
    L.out = take(sizeof(DevResult));
    L.evw = 0; L.mbstride = 0; L.mbox = L.bar = L.cand = 0;
    if (G > 0) { // cluster kernel: mailboxes, arrival counter, early-leaver candidate scratch
        L.evw = G == 1 ? 4096 : 512;
        const size_t lgroups = ((n + 63) / 64 + (size_t)G - 1) / (size_t)G;
        L.mbstride = (int)align_up((size_t)2 * (1 + c.nPreamble + L.evw + (lgroups + 1) / 2), 4); // 8-byte granules: header, buckets, events, (glibc) group draw counts
        if (c.variant == PRACH_VARIANT_NOMA_C) L.mbstride = (int)align_up((size_t)2 * (1 + 6 * c.nPreamble), 4); // header + 6 x nP bins
        L.mbox = take(4 * (size_t)2 * G * L.mbstride);
        L.bar = take(256);
        L.cand = take(8 * (n + 64 * (size_t)G + 64));
    }
    L.n_pre0 = L.n_sector = L.n_gain = L.n_lgain = L.n_nd0 = 0;
    if (c.variant == PRACH_VARIANT_NOMA_C) {
        L.n_pre0 = take(4 * n); L.n_sector = take(4 * n); L.n_gain = take(8 * n); L.n_lgain = take(8 * n); L.n_nd0 = take(4 * n);
    }
    L.end = o;
    return L;
}

// initial glibc stream budget (draws) for a trial; the kernel reports exhaustion and we retry bigger
uint64_t stream_budget(const prach_cfg &c, int attempt, int64_t factor) {
    uint64_t per = factor > 0 ? (uint64_t)factor : 450; // 100k UEs / 12 grants / backoff 20 consume ~325 per UE
    uint64_t b = (uint64_t)c.nUE * per + (1u << 20);
    return b << (2 * attempt);
}

} // namespace

extern "C" {

int prach_engine_create(int device, prach_engine **out) {
    if (!out) return PRACH_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "[prach] no HIP device: libprach_hip needs an MI355X (gfx950); there is no CPU fallback\n");
        return PRACH_ERR_DEVICE;
    }
    if (device < 0 || device >= ndev) return PRACH_ERR_ARG;
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "[prach] device %d is %s; this library is built for gfx950 only\n", device, prop.gcnArchName);
        return PRACH_ERR_DEVICE;
    }
    prach_engine *e = new prach_engine();
    e->device = device;
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&e->ev0));
    HIPCHK(hipEventCreate(&e->ev1));
    *out = e;
    return PRACH_OK;
}

void prach_engine_destroy(prach_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->arena) (void)hipFree(e->arena);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int prach_engine_set(prach_engine *e, const char *key, int64_t value) {
    if (!e || !key) return PRACH_ERR_ARG;
    if (std::strcmp(key, "stream_factor") == 0) { e->opt_stream_factor = value; return PRACH_OK; }
    if (std::strcmp(key, "cluster") == 0) { if (value < 0 || value > CLUSTER_MAX_G) return PRACH_ERR_ARG; e->opt_cluster = value; return PRACH_OK; }
    if (std::strcmp(key, "legacy") == 0) { e->opt_legacy = value != 0; return PRACH_OK; }
    if (std::strcmp(key, "dense") == 0) { e->opt_dense = value != 0; return PRACH_OK; }
    if (std::strcmp(key, "wide_records") == 0) { e->opt_wide_records = value != 0; return PRACH_OK; }
    if (std::strcmp(key, "pipeline") == 0) { e->opt_pipeline = value != 0; return PRACH_OK; }
    return PRACH_ERR_ARG;
}

int prach_device_glibc_stream(prach_engine *e, uint32_t seed, uint64_t first, uint64_t n, int32_t *out) {
    if (!e || !out || n == 0) return PRACH_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    const size_t nchunks = (n + STREAM_CHUNK - 1) / STREAM_CHUNK;
    const size_t need = align_up(4 * 31 * nchunks, 256) + 4 * (n + 2);
    if (need > e->arena_cap) {
        if (e->arena) HIPCHK(hipFree(e->arena));
        e->arena = nullptr; e->arena_cap = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&e->arena), need));
        e->arena_cap = need;
    }
    std::vector<uint32_t> seeds(31 * nchunks);
    prach_internal_glibc_seeds(seed, first, nchunks, STREAM_CHUNK, seeds.data());
    HIPCHK(hipMemcpy(e->arena, seeds.data(), 4 * 31 * nchunks, hipMemcpyHostToDevice));
    int *dout = reinterpret_cast<int *>(e->arena + align_up(4 * 31 * nchunks, 256));
    HIPCHK(launch_glibc_stream(reinterpret_cast<const unsigned *>(e->arena), dout, n, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out, dout, 4 * n, hipMemcpyDeviceToHost));
    return PRACH_OK;
}

int prach_last_timing(const prach_engine *e, prach_timing *out) {
    if (!e || !out) return PRACH_ERR_ARG;
    *out = e->last;
    return PRACH_OK;
}

// one launch over the trials idx[0..m) (all the same rng_mode); attempt = glibc stream retry level
static int run_group(prach_engine *e, const prach_cfg *cfgs, const int *idx, int m, prach_result *results,
                     prach_ue_log *const *ue_logs, int attempt, int G, double &kernel_ms, double &upload_ms) {
    const int rng_mode = cfgs[idx[0]].rng_mode;
    std::vector<TrialLayout> lay(m);
    size_t o = align_up(sizeof(TrialDev) * (size_t)m, 256);
    int maxP = 1;
    for (int k = 0; k < m; k++) {
        const prach_cfg &c = cfgs[idx[k]];
        const bool wl = ue_logs && ue_logs[idx[k]];
        const size_t sl = rng_mode == PRACH_RNG_GLIBC ? (size_t)stream_budget(c, attempt, e->opt_stream_factor) : 0;
        lay[k] = layout_trial(c, o, wl, sl, G);
        o = lay[k].end;
        if (c.nPreamble > maxP) maxP = c.nPreamble;
    }
    if (o > e->arena_cap) {
        if (e->arena) HIPCHK(hipFree(e->arena));
        e->arena = nullptr;
        e->arena_cap = 0;
        const size_t want = o + (o >> 3);
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&e->arena), want));
        e->arena_cap = want;
    }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<TrialDev> td(m);
    std::vector<int32_t> sched;
    std::vector<uint32_t> seeds;
    std::vector<int32_t> nAccess(m, 0);
    for (int k = 0; k < m; k++) {
        const prach_cfg &c = cfgs[idx[k]];
        const TrialLayout &L = lay[k];
        TrialDev &d = td[k];
        d.variant = c.variant; d.uniform = c.uniform; d.nUE = c.nUE; d.nP = c.nPreamble; d.backoff = c.backoff;
        d.nGrantUL = c.nGrantUL; d.maxRarWindow = c.maxRarWindow; d.maxMsg2 = c.maxMsg2TxCount; d.aT = c.accessTime;
        d.rng_mode = c.rng_mode; d.maxTime = prach_max_time(&c);
        d.stop = (c.max_steps > 0 && c.max_steps < d.maxTime) ? c.max_steps : d.maxTime;
        d.seed_lo = (unsigned)c.seed; d.seed_hi = (unsigned)(c.seed >> 32);
        d.stream_len = L.stream_len;
        d.dense_pass = e->opt_dense ? 1 : 0;
        d.pipeline = e->opt_pipeline ? 1 : 0;
        char *A = e->arena;
        d.rec = reinterpret_cast<int4 *>(A + L.rec);
        d.ptc = reinterpret_cast<int *>(A + L.ptc); d.ftt = reinterpret_cast<int *>(A + L.ftt);
        d.stt = reinterpret_cast<int *>(A + L.stt); d.fcnt = reinterpret_cast<int *>(A + L.fcnt);
        d.nd = reinterpret_cast<unsigned *>(A + L.nd);
        d.evbuf = reinterpret_cast<Event *>(A + L.evbuf); d.evbuf2 = reinterpret_cast<Event *>(A + L.evbuf2);
        d.sidx = reinterpret_cast<int *>(A + L.sidx);
        d.sched = reinterpret_cast<const int *>(A + L.sched);
        d.stream = reinterpret_cast<const int *>(A + L.stream);
        d.logs = L.logs ? reinterpret_cast<prach_ue_log *>(A + L.logs) : nullptr;
        d.timers = reinterpret_cast<int *>(A + L.timers);
        d.out = reinterpret_cast<DevResult *>(A + L.out);
        d.evw = L.evw; d.mbstride = L.mbstride;
        d.binshift = 0;
        while (((c.nUE - 1) >> d.binshift) >= 1024) d.binshift++;
        d.mbox = G > 0 ? reinterpret_cast<int *>(A + L.mbox) : nullptr;
        d.bar = G > 0 ? reinterpret_cast<unsigned *>(A + L.bar) : nullptr;
        d.cand = G > 0 ? reinterpret_cast<int2 *>(A + L.cand) : nullptr;
        if (G > 0) { // words polled / accumulated in-kernel are zeroed before EVERY launch
            HIPCHK(hipMemsetAsync(A + L.out, 0, sizeof(DevResult), e->stream));
            HIPCHK(hipMemsetAsync(A + L.bar, 0, 256, e->stream));
            HIPCHK(hipMemsetAsync(A + L.mbox, 0, 4 * (size_t)2 * G * L.mbstride, e->stream)); // tags: 0 never equals t+1
        }
        sched.assign(L.sched_len, c.nUE);
        prach_arrival_schedule(&c, sched.data(), (int)L.sched_len, &nAccess[k]);
        HIPCHK(hipMemcpyAsync(A + L.sched, sched.data(), 4 * L.sched_len, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream)); // staging vector is reused
        d.n_pre0 = d.n_sector = nullptr; d.n_gain = d.n_lgain = nullptr; d.n_nd0 = nullptr;
        if (c.variant == PRACH_VARIANT_NOMA_C) { // activeUE's per-UE attributes (double-precision libm work): host, once per trial
            const size_t nn = (size_t)c.nUE;
            std::vector<int32_t> pre0(nn), sec(nn);
            std::vector<double> gn(nn), lg(nn);
            std::vector<uint32_t> nd0(nn);
            int trc = prach_noma_activation_table(&c, pre0.data(), sec.data(), gn.data(), lg.data(), nd0.data());
            if (trc != PRACH_OK) return trc;
            HIPCHK(hipMemcpy(A + L.n_pre0, pre0.data(), 4 * nn, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(A + L.n_sector, sec.data(), 4 * nn, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(A + L.n_gain, gn.data(), 8 * nn, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(A + L.n_lgain, lg.data(), 8 * nn, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(A + L.n_nd0, nd0.data(), 4 * nn, hipMemcpyHostToDevice));
            d.n_pre0 = reinterpret_cast<const int *>(A + L.n_pre0); d.n_sector = reinterpret_cast<const int *>(A + L.n_sector);
            d.n_gain = reinterpret_cast<const double *>(A + L.n_gain); d.n_lgain = reinterpret_cast<const double *>(A + L.n_lgain);
            d.n_nd0 = reinterpret_cast<const unsigned *>(A + L.n_nd0);
        }
        if (rng_mode == PRACH_RNG_GLIBC) {
            // the reference's rand() stream window [stream_offset, +stream_len): the host only jumps ahead (31-word window
            // per chunk, cached matrix powers); the values themselves are generated on the device inside the timed region
            seeds.resize(31 * (L.nchunks + 1));
            prach_internal_glibc_seeds((uint32_t)c.seed, c.stream_offset, L.nchunks, STREAM_CHUNK, seeds.data());
            HIPCHK(hipMemcpyAsync(A + L.seeds, seeds.data(), 4 * 31 * L.nchunks, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
        }
    }
    HIPCHK(hipMemcpyAsync(e->arena, td.data(), sizeof(TrialDev) * (size_t)m, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    auto t1 = std::chrono::steady_clock::now();
    upload_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();

    HIPCHK(hipEventRecord(e->ev0, e->stream));
    if (rng_mode == PRACH_RNG_GLIBC)
        for (int k = 0; k < m; k++)
            HIPCHK(launch_glibc_stream(reinterpret_cast<const unsigned *>(e->arena + lay[k].seeds), reinterpret_cast<int *>(e->arena + lay[k].stream),
                                       (unsigned long long)lay[k].stream_len, e->stream));
    if (cfgs[idx[0]].variant == PRACH_VARIANT_NOMA_C) HIPCHK(launch_noma_kernel(reinterpret_cast<const TrialDev *>(e->arena), m, G, maxP, e->stream));
    else if (G > 0) {
        // one workgroup per trial = the streaming regime: 8 + 4 byte hot records, if every subframe number of every trial of
        // the launch fits 16 bits (txTime <= t + 59 + backoff + accessTime)
        bool compact = G == 1 && !e->opt_wide_records;
        for (int k = 0; k < m && compact; k++) {
            const prach_cfg &c = cfgs[idx[k]];
            compact = (int64_t)prach_max_time(&c) + c.backoff + c.accessTime + 64 < 63000;
        }
        HIPCHK(launch_cluster_kernel(reinterpret_cast<const TrialDev *>(e->arena), m, G, maxP, rng_mode, compact ? 1 : 0, e->stream));
    }
    else HIPCHK(launch_trial_kernel(reinterpret_cast<const TrialDev *>(e->arena), m, rng_mode, maxP, e->stream));
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    kernel_ms += ms;
    e->last.launches++;
    e->last.workgroups = m * (G > 0 ? G : 1);
    e->last_G = G;

    std::vector<int32_t> timers;
    for (int k = 0; k < m; k++) {
        const prach_cfg &c = cfgs[idx[k]];
        const TrialLayout &L = lay[k];
        DevResult dr;
        HIPCHK(hipMemcpy(&dr, e->arena + L.out, sizeof(dr), hipMemcpyDeviceToHost));
        prach_result &r = results[idx[k]];
        std::memset(&r, 0, sizeof(r));
        r.status = dr.status;
        r.time_exit = dr.time_exit;
        r.maxTime = prach_max_time(&c);
        r.nSuccessUE = dr.nSuccess;
        r.failedUEs = c.nUE - dr.nSuccess;
        r.preambleTxCount = dr.ptcSum;
        r.failCounts = dr.fcSum;
        r.collisionPreambles = dr.collisionPreambles;
        r.totalPreambleTxop = dr.totalPreambleTxop;
        r.activeCheck = dr.activeCheck;
        r.nAccessUE = nAccess[k];
        r.continueFaliedUEs = dr.continueFailed;
        r.finalSuccessUEs = dr.finalSuccess;
        r.sumTimer = dr.sumTimer;
        r.draws = dr.draws;
        r.steps = dr.steps;
        if (c.variant == PRACH_VARIANT_NOMA_C && dr.nSuccess == c.nUE && dr.dbg[0] > 0) { // all UEs succeeded: NOMA.c:707-710 breaks there
            r.time_exit = (int32_t)dr.dbg[0] - 1;
            r.steps = dr.dbg[0];
        }
        if (std::getenv("PRACH_PRINT_STAMPS"))
            std::fprintf(stderr, "[prach stamps/step] pass=%.0f publish=%.0f barrier=%.0f gather=%.0f r1b=%.0f leavers=%.0f checks=%.0f grants=%.0f cycles | N avg %.1f max %llu, resetcand avg %.2f, singles avg %.1f max %llu\n",
                         dr.stamps6[0] / (double)dr.steps, dr.stamps6[1] / (double)dr.steps, dr.stamps6[2] / (double)dr.steps, dr.stamps6[3] / (double)dr.steps,
                         dr.stamps6[4] / (double)dr.steps, dr.stamps6[5] / (double)dr.steps, dr.stamps6[6] / (double)dr.steps, dr.stamps6[7] / (double)dr.steps,
                         dr.dbg[0] / (double)dr.steps, dr.dbg[1], dr.dbg[2] / (double)dr.steps, (dr.dbg[3] >> 20) / (double)dr.steps, dr.dbg[3] & 0xfffff);
        if (dr.status != PRACH_OK) continue;
        // totalDelay is a FLOAT running sum in index order (Beta.c:186,193): exact in integer
        // arithmetic while it stays below 2^24, otherwise replay the float additions on the host.
        if (dr.sumTimer < (1ll << 24)) {
            r.totalDelay = (float)dr.sumTimer;
        } else {
            timers.resize((size_t)c.nUE);
            HIPCHK(hipMemcpy(timers.data(), e->arena + L.timers, 4 * (size_t)c.nUE, hipMemcpyDeviceToHost));
            float td_ = 0;
            for (int i = 0; i < c.nUE; i++)
                if (timers[i] != INT_MIN) td_ += (float)timers[i];
            r.totalDelay = td_;
        }
        if (L.logs)
            HIPCHK(hipMemcpy(ue_logs[idx[k]], e->arena + L.logs, sizeof(prach_ue_log) * (size_t)c.nUE, hipMemcpyDeviceToHost));
    }
    return PRACH_OK;
}

int prach_run_trials(prach_engine *e, const prach_cfg *cfgs, int n, prach_result *results, prach_ue_log *const *ue_logs) {
    if (!e || !cfgs || !results || n <= 0) return PRACH_ERR_ARG;
    for (int k = 0; k < n; k++) {
        int v = prach_cfg_validate(&cfgs[k]);
        if (v != PRACH_OK) return v;
        if (cfgs[k].variant == PRACH_VARIANT_NOMA_C && (cfgs[k].rng_mode != PRACH_RNG_PHILOX || cfgs[k].nPreamble > 64 || cfgs[k].uniform))
            return PRACH_ERR_UNSUPPORTED; // the rejection loops of activeUE make the glibc stream position data dependent
    }
    HIPCHK(hipSetDevice(e->device));
    auto t0 = std::chrono::steady_clock::now();
    e->last = prach_timing{};
    double kernel_ms = 0, upload_ms = 0;
    { // NOMA.c variant: its own kernel, G workgroups per trial like the production kernel
        std::vector<int> idx;
        for (int k = 0; k < n; k++)
            if (cfgs[k].variant == PRACH_VARIANT_NOMA_C) idx.push_back(k);
        if (!idx.empty()) {
            int minGroups = INT_MAX;
            bool small = true;
            for (int k : idx) { minGroups = std::min(minGroups, (cfgs[k].nUE + 63) / 64); small = small && cfgs[k].nUE < (1 << 20) - 1; }
            int G = (int)e->opt_cluster;
            const size_t resident = (size_t)e->num_cus * 3 / 4;
            // (measured at nUE = 100 000: 23.2 ms with 16 workgroups, 24.3 with 32, 26.6 with 8: 6 x nPreamble bins per mailbox)
            if (G <= 0) { G = 1; while (G * 2 <= 16 && (size_t)G * 2 * idx.size() <= resident / 2 + resident / 6 && G * 2 <= std::max(1, minGroups / 16)) G *= 2; }
            while (G > 1 && (size_t)G * idx.size() > resident) G /= 2;
            if (!small) G = 1; // 20-bit granule fields
            int rc = run_group(e, cfgs, idx.data(), (int)idx.size(), results, ue_logs, 0, G, kernel_ms, upload_ms);
            if (rc != PRACH_OK) return rc;
        }
    }
    for (int mode = 0; mode < 2; mode++) {
        std::vector<int> idx;
        for (int k = 0; k < n; k++)
            if (cfgs[k].rng_mode == mode && cfgs[k].variant != PRACH_VARIANT_NOMA_C) idx.push_back(k);
        if (idx.empty()) continue;
        // longest trials first: workgroups are dispatched in index order as CUs free up, so the tail of a
        // many-trial launch is made of the short trials
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
            auto work = [&](int k) { return (uint64_t)cfgs[k].nUE * (uint64_t)(cfgs[k].max_steps > 0 ? cfgs[k].max_steps : prach_max_time(&cfgs[k])); };
            return work(a) > work(b);
        });
        bool cluster_ok = !e->opt_legacy;
        for (int k : idx) {
            cluster_ok = cluster_ok && cfgs[k].nUE < (1 << 20) - 1; // 20-bit granule fields, dead-group bitmap

            if (mode == PRACH_RNG_GLIBC) cluster_ok = cluster_ok && cfgs[k].nUE <= CLUSTER_GLIBC_MAX_UE;
        }
        if (cluster_ok) {
            // production path: cluster kernel, G workgroups per trial (all clusters must be co-resident:
            // at most one 1024-thread workgroup per CU)
            int minGroups = INT_MAX;
            for (int k : idx) minGroups = std::min(minGroups, (cfgs[k].nUE + 63) / 64);
            int G = (int)e->opt_cluster;
            const size_t resident = (size_t)e->num_cus * 3 / 4; // every cluster must be co-resident: <= one workgroup per CU, with margin
            if (G <= 0) {
                G = 1;
                while (G * 2 <= 32 && (size_t)G * 2 * idx.size() <= resident / 2 + resident / 6 && G * 2 <= std::max(1, minGroups / 16)) G *= 2;
                // Uniform arrivals over 60 000 subframes (Beta.c:92-95): only nUE / 60 000 arrivals per subframe, a UE lives some
                // tens of subframes, finished groups are skipped 32 at a time — the live band is a few groups and one workgroup
                // steps through a subframe faster than a cluster exchanges (nUE = 100 000: 5.1 vs 6.2 us per subframe)
                bool light = mode == PRACH_RNG_PHILOX;
                for (int k : idx) light = light && cfgs[k].uniform && cfgs[k].nUE <= 2000000;
                if (light) G = 1;
            }
            while (G > 1 && (size_t)G * idx.size() > resident) G /= 2;
            std::vector<int> todo = idx, fallback;
            for (int attempt = 0; !todo.empty(); attempt++) {
                if (attempt > 6) return PRACH_ERR_STREAM;
                int rc = run_group(e, cfgs, todo.data(), (int)todo.size(), results, ue_logs, attempt, G, kernel_ms, upload_ms);
                if (rc != PRACH_OK) return rc;
                std::vector<int> again;
                for (int k : todo) {
                    if (results[k].status == PRACH_ERR_STREAM) again.push_back(k);         // glibc: draw-stream window ran out: larger one
                    else if (results[k].status == PRACH_ERR_INTERNAL) fallback.push_back(k); // a per-subframe capacity was exceeded:
                }                                                                              // exact rerun on trial_kernel
                todo.swap(again);
            }
            idx.swap(fallback);
            if (idx.empty()) continue;
        }
        int attempt = 0;
        while (!idx.empty()) {
            int rc = run_group(e, cfgs, idx.data(), (int)idx.size(), results, ue_logs, attempt, 0, kernel_ms, upload_ms);
            if (rc != PRACH_OK) return rc;
            std::vector<int> again; // glibc trials whose draw-stream window ran out: rerun with a larger one
            for (int k : idx)
                if (results[k].status == PRACH_ERR_STREAM) again.push_back(k);
            idx.swap(again);
            if (++attempt > 6) return PRACH_ERR_STREAM;
        }
    }
    uint64_t upd = 0;
    int worst = PRACH_OK;
    for (int k = 0; k < n; k++) {
        upd += (uint64_t)cfgs[k].nUE * results[k].steps;
        if (results[k].status != PRACH_OK) worst = results[k].status;
    }
    e->last.kernel_ms = kernel_ms;
    e->last.upload_ms = upload_ms;
    e->last.updates = upd;
    e->last.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return worst;
}

} // extern "C"
