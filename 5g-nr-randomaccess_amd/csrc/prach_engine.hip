// prach_engine.hip — host side of libprach_hip.so: the engine behind the C ABI in include/prach.h.
// Owns the HIP stream, one device arena and one pinned staging buffer (both grown on demand, nothing is
// allocated per subframe or per trial), stages the per-trial parameter blocks / arrival tables / draw-stream
// seeds of a whole launch with ONE copy, launches the trial kernels and turns DevResult into prach_result.
// There is NO CPU fallback: without a gfx950 device every entry point that simulates returns PRACH_ERR_DEVICE.
#include "prach_device.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits.h>
#include <new>
#include <thread>
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
    // the arena as ONE reserved virtual range that physical memory is mapped into piece by piece (hipMemAddressReserve / hipMemCreate / hipMemMap): growing it
    // maps the increment only — no hipFree + hipMalloc of the whole arena (measured 0.8-1.4 s for 10-16 GB when a sweep's calls grow point by point) — and
    // its address never changes.  vmm: 0 untried, 1 in use, -1 unavailable (plain hipMalloc arena).
    int vmm = 0;
    size_t vmm_reserved = 0, vmm_gran = 0;
    std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> vmm_parts;
    char *pinned = nullptr; // host staging mirror of the head of the arena (parameter blocks, arrival tables, stream seeds, results)
    size_t pinned_cap = 0;
    prach_timing last{};
    int64_t opt_stream_factor = 0; // glibc: initial draws-per-UE budget override (0 = auto)
    double draws_per_ue_seen = 0;  // glibc: the largest rand() consumption per UE of the last call's trials (0: none yet) — sizes the next call's stream windows
    int64_t opt_cluster = 0;       // workgroups per trial for the Philox cluster kernel (0 = auto)
    int64_t opt_legacy = 0;        // 1: run Philox trials on the one-workgroup trial_kernel as well
    int64_t opt_dense = 0;         // 1: cluster kernel without the compacted pass (diagnostic)
    int64_t opt_wide_records = 0;  // 1: 16-byte records also with one workgroup per trial (diagnostic)
    int64_t opt_pipeline = 1;      // 0: clusters do not run phase A ahead of the exchange (diagnostic)
    int64_t opt_resident = 0;      // test hook: pretend only this many workgroups can be co-resident (0 = ask the runtime)
    int64_t opt_host_threads = 0;  // host threads for the NOMA.c activation tables (0 = all cores)
#ifdef PRACH_QCAP                  // (the small-queue test build exists to exercise the queue-overflow path of the global-record kernels)
    int64_t opt_lds_records = 0;
#else
    int64_t opt_lds_records = 1;   // 0: clusters keep their UE records in global memory (diagnostic)
#endif
    int64_t opt_xcd_pack = 1;      // 1: lean clusters are launched XCD-packed (a cluster per XCD; granules stay in that XCD's L2 once verified)
    bool pack_off = false;         // (set for the rerun of a packed launch that timed out)
    int64_t opt_fast = 1;          // 0: LDS-resident clusters run on the general kernel (prach_cluster.hip) instead of prach_lcluster.hip
    int64_t opt_batch = 1;         // 0: one-workgroup-per-trial Philox launches run on the general kernel instead of prach_batch.hip
    int64_t opt_batch_waves = 0;   // wavefronts per batch-kernel workgroup: 8 (512 threads, two trials per CU), 16 (one), 0 = chosen per launch
    int64_t opt_plain_arena = 0;   // 1: the arena is one hipMalloc allocation, re-allocated when it grows (diagnostic)
    int64_t opt_vmm_fail_after = 0; // test hook: the reserved-range arena cannot map a further piece once it has this many (0: no limit)
    int64_t opt_noma_host_activation = 0; // 1: NOMA.c's activeUE table is built on the host (the reference's libm) instead of by noma_activation_kernel
    int64_t opt_noma_ambiguity_test = 0;  // test hook: the resolver reports every gain sort as ambiguous (exercises the rerun with the host-built table)
    bool force_host_act = false;   // (set for the rerun of trials whose device-built table left a gain comparison inside the error band)
    int64_t opt_calendar_cap = 0;  // test hook: entries per calendar list of the batch kernel (0 = sized per trial)
    bool full_calendars = false;   // (set for the rerun of batch-kernel trials that filled a calendar list: lists of nUE entries cannot fill)
    std::vector<int> cal_overflow; // last launch: the trials that filled a calendar list
    int noma_flagged = 0, noma_ambiguous = 0; // last call: UEs recomputed on the host, trials rerun with the host-built table
    int num_cus = 256;
    size_t mem_budget = (size_t)200 << 30; // arena bytes one launch may take (3/4 of the device's memory): a call that needs more runs as several launches
};

namespace {

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct TrialLayout {
    size_t rec, ptc, ftt, stt, fcnt, nd, evbuf, evbuf2, sidx, sched, stream, logs, timers, out, mbox, cand;
    size_t n_pre0, n_sector, n_gain, n_lgain, n_nd0, sector;
    size_t rec32, chunks, ctab, cpool, jcal, qov, evov; // batch kernel
    int calcap, calslots, npool, tcap;
    size_t diag;                 // diagnostic build: per-workgroup stamps (zeroed region)
    size_t seeds, nchunks; // glibc: one 31-word window per STREAM_CHUNK outputs (device-side generation)
    size_t stream_len, sched_len;
    int evw, mbstride;
};

// Arena of one launch:
//   [ TrialDev[m] | arrival tables | glibc stream seeds ]   staged region: built in pinned host memory, ONE copy
//   [ DevResult[m] | mailboxes ]                             zeroed region: ONE memset per launch (tags: 0 never equals t+1);
//                                                            the DevResult block also comes back with ONE copy
//   [ per-trial arrays ]                                     records, cold fields, scratch of the kernel that runs
struct LaunchLayout {
    std::vector<TrialLayout> t;
    size_t staged_end = 0, zero_begin = 0, zero_end = 0, out0 = 0, end = 0;
    size_t act_flags = 0; // NOMA.c: noma_activation_kernel's list of UEs for the host (in the zeroed region)
    size_t stream_jobs = 0; // glibc: StreamJob[m] (in the staged region)
};

size_t mbox_bytes(const prach_cfg &c, int G, int &evw, int &mbstride) {
    evw = 0; mbstride = 0;
    if (G <= 0) return 0;
    const size_t n = (size_t)c.nUE;
    evw = G == 1 ? 4096 : CLUSTER_EVW;
    const size_t lgroups = ((n + 63) / 64 + (size_t)G - 1) / (size_t)G;
    mbstride = (int)align_up((size_t)2 * (1 + c.nPreamble + evw + (lgroups + 1) / 2), 4); // 8-byte granules: header, buckets, events, (glibc) group draw counts
    if (c.variant == PRACH_VARIANT_NOMA_C) mbstride = (int)align_up((size_t)2 * (1 + 6 * c.nPreamble), 4); // header + 6 x nP bins
    if (G == 1) return 256; // one workgroup per trial exchanges nothing
    return 4 * (size_t)2 * G * mbstride;
}

// stream_len[k]: glibc draw-stream window of trial k (0 in Philox mode); G: workgroups per trial (0 = trial_kernel)
LaunchLayout layout_launch(const prach_cfg *cfgs, const int *idx, int m, prach_ue_log *const *ue_logs, const std::vector<size_t> &stream_len, int G, bool batch, bool full_calendars, int64_t calendar_cap) {
    LaunchLayout L;
    L.t.resize(m);
    size_t o = align_up(sizeof(TrialDev) * (size_t)m, 256);
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    for (int k = 0; k < m; k++) { // staged region
        const prach_cfg &c = cfgs[idx[k]];
        TrialLayout &T = L.t[k];
        T.sched_len = (size_t)(prach_max_time(&c) / c.accessTime + 2);
        T.sched = take(4 * T.sched_len);
        T.stream_len = stream_len[k];
        T.nchunks = (T.stream_len + STREAM_CHUNK - 1) / STREAM_CHUNK;
        T.seeds = T.stream_len ? take(4 * 31 * (T.nchunks + 1)) : 0;
    }
    bool any_stream = false;
    for (int k = 0; k < m; k++) any_stream = any_stream || stream_len[k] > 0;
    if (any_stream) L.stream_jobs = take(sizeof(StreamJob) * (size_t)m);
    L.staged_end = o;
    L.zero_begin = o;
    L.out0 = take(sizeof(DevResult) * (size_t)m);
    for (int k = 0; k < m; k++) {
        L.t[k].out = L.out0 + sizeof(DevResult) * (size_t)k;
        const size_t mb = mbox_bytes(cfgs[idx[k]], G, L.t[k].evw, L.t[k].mbstride);
        L.t[k].mbox = mb ? take(mb) : 0;
#ifdef PRACH_STAMPS
        L.t[k].diag = G > 1 ? take(8 * 32 * (size_t)CLUSTER_MAX_G) : 0;
#else
        L.t[k].diag = 0;
#endif
    }
    if (cfgs[idx[0]].variant == PRACH_VARIANT_NOMA_C) L.act_flags = take(4 * (2 + 2 * (size_t)NOMA_ACT_FLAG_CAP));
    L.zero_end = o;
    for (int k = 0; k < m; k++) {
        const prach_cfg &c = cfgs[idx[k]];
        TrialLayout &T = L.t[k];
        const size_t n = (size_t)c.nUE;
        T.rec = T.ptc = T.ftt = T.stt = T.fcnt = T.nd = T.rec32 = T.chunks = T.ctab = T.cpool = T.jcal = T.qov = T.evov = 0;
        T.calcap = T.calslots = T.npool = T.tcap = 0;
        if (batch) { // prach_batch.hip: 32-byte event records, the two calendars, the global parts of the candidate and event lists
            // The records of the UEs under way travel in 2 KB chunks of 64: every UE has one record, so ceil(nUE / 64) full chunks, plus the open chunks of the
            // wavefronts (at most 128 each), plus what sits in their stacks of free ids (64 each), plus slack: twice the full chunks + 3 136.  A join list holds
            // the UEs whose window opens in ONE subframe: txTime is aligned to the access slots (Beta.c:268-277), so an overloaded trial (more UEs than its UL
            // grants can serve: every UE cycles through backoff and window about every 17 subframes) puts ~0.3 nUE into one list, another trial far less.
            // A trial that exhausts either leaves with PRACH_ERR_INTERNAL and is rerun with a doubled pool and lists of nUE entries (run_trials_impl).
            T.calslots = batch_calendar_slots(c.backoff, c.accessTime, c.maxRarWindow);
            const double steps_ = (double)((c.max_steps > 0 && c.max_steps < prach_max_time(&c)) ? c.max_steps : prach_max_time(&c));
            const bool overloaded = (double)c.nUE > (double)std::max(0, c.nGrantUL - 1) * steps_ / 5.0;
            T.calcap = (int)(n <= 16384 ? n : std::max<size_t>(16384, overloaded ? n * 2 / 5 : n / 8)) + 64; // (a small trial gets lists that cannot fill)
            if (calendar_cap > 0) T.calcap = (int)calendar_cap;
            T.npool = 2 * (int)((n + 63) / 64) + 16 * (128 + 64) + 64;
            if (full_calendars) { T.calcap = (int)n + 64; T.npool *= 2; }
            T.tcap = (int)((n + 63) / 64) + 256;
            T.rec32 = take(32 * n); T.chunks = take((size_t)batch_chunk_bytes() * (size_t)T.npool); T.ctab = take(4 * (size_t)T.calslots * (size_t)T.tcap);
            T.cpool = take(8 * (size_t)T.npool); T.jcal = take(4 * ((size_t)T.calslots * (size_t)T.calcap + 64)); // (+ 64 words: where lanes without a join entry store, prach_batch.hip)
            T.qov = take(4 * n); T.evov = take(16 * n);
        } else {
            T.rec = take(16 * n);
            T.ptc = take(4 * n); T.ftt = take(4 * n); T.stt = take(4 * n); T.fcnt = take(4 * n); T.nd = take(4 * n);
        }
        T.evbuf = T.evbuf2 = T.sidx = T.cand = 0;
        if (G == 0) { // trial_kernel only: event / singleton scratch without a per-subframe capacity
            T.evbuf = take(sizeof(Event) * n); T.evbuf2 = take(sizeof(Event) * n);
            T.sidx = take(4 * (n + 256));
        } else if (c.variant != PRACH_VARIANT_NOMA_C) {
            T.cand = take(8 * (n + 64 * (size_t)G + 64)); // cluster kernel: early-leaver candidates
        }
        T.stream = T.stream_len ? take(4 * (T.stream_len + 2)) : 0;
        T.logs = (ue_logs && ue_logs[idx[k]]) ? take(sizeof(prach_ue_log) * n) : 0;
        T.timers = take(4 * n);
        T.sector = (c.flags & PRACH_FLAG_SECTOR_GRANTS) ? take(4 * n) : 0;
        T.n_pre0 = T.n_sector = T.n_gain = T.n_lgain = T.n_nd0 = 0;
        if (c.variant == PRACH_VARIANT_NOMA_C) {
            T.n_pre0 = take(4 * n); T.n_sector = take(4 * n); T.n_gain = take(8 * n); T.n_lgain = take(8 * n); T.n_nd0 = take(4 * n);
        }
    }
    L.end = o;
    return L;
}

// initial glibc stream budget (draws) for a trial; the kernel reports exhaustion and we retry bigger
// (a sweep is called point after point with growing nUE: what the previous point's trials consumed per UE, tripled, is the first window of the next —
//  a Beta.c point with 54 grants draws ~30 values per UE, not 450: 100 seeds x 100 000 UEs are then 3 GB of windows instead of 18 GB; a window that
//  turns out too small is the ordinary PRACH_ERR_STREAM retry, four times larger)
uint64_t stream_budget(const prach_cfg &c, int attempt, int64_t factor, double seen) {
    uint64_t per = factor > 0 ? (uint64_t)factor : seen > 0 ? (uint64_t)std::min(450.0, std::max(64.0, 3.0 * seen + 32.0)) : 450; // 100k UEs / 12 grants / backoff 20 consume ~325 per UE
    if (attempt > 0 && factor <= 0) per = 450; // (a window sized from a light previous call was too small: straight to the default, then four times larger per retry)
    uint64_t b = (uint64_t)c.nUE * per + (1u << 20);
    return b << (2 * std::max(0, attempt - (seen > 0 && factor <= 0 ? 1 : 0)));
}

// grows the reserved-range arena to at least `need` mapped bytes; false: the mechanism is not available here (nothing has been changed)
static bool grow_vmm_arena(prach_engine *e, size_t need) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = e->device;
    if (e->vmm == 0) {
        size_t gran = 0, free_b = 0, total_b = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) { (void)hipGetLastError(); e->vmm = -1; return false; }
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) { (void)hipGetLastError(); e->vmm = -1; return false; }
        const size_t reserve = align_up(total_b, gran); // (address space, not memory)
        void *p = nullptr;
        if (hipMemAddressReserve(&p, reserve, 0, nullptr, 0) != hipSuccess || !p) { (void)hipGetLastError(); e->vmm = -1; return false; }
        e->arena = static_cast<char *>(p); e->arena_cap = 0; e->vmm_reserved = reserve; e->vmm_gran = std::max(gran, (size_t)2 << 20); e->vmm = 1; // (pieces in multiples of 2 MB whatever the runtime recommends)
    }
    if (need > e->vmm_reserved) return false;
    // at least a quarter more than what is mapped, at least 256 MB: a sweep's growing calls map a handful of increments in all — each increment in pieces of
    // at most 1 GB (hipMemSetAccess refused pieces of 2 GB and more: hipErrorInvalidValue)
    size_t delta = align_up(std::max(need - e->arena_cap, std::max(e->arena_cap >> 2, (size_t)256 << 20)), e->vmm_gran);
    delta = std::min(delta, e->vmm_reserved - e->arena_cap);
    const size_t piece_max = align_up((size_t)1 << 30, e->vmm_gran); // (hipMemSetAccess refuses a piece of 2^31 bytes: hipErrorInvalidValue)
    while (delta > 0) {
        const size_t piece = std::min(delta, piece_max);
        if (e->opt_vmm_fail_after > 0 && (int64_t)e->vmm_parts.size() >= e->opt_vmm_fail_after) return false; // (test hook: as if the device had no more memory to map)
        hipMemGenericAllocationHandle_t h;
        auto why = [&](const char *what, hipError_t rc) { if (std::getenv("PRACH_VERBOSE")) std::fprintf(stderr, "[prach] %s of a %zu-byte piece at offset %zu: %s\n", what, piece, e->arena_cap, hipGetErrorName(rc)); (void)hipGetLastError(); };
        hipError_t rc = hipMemCreate(&h, piece, &prop, 0);
        if (rc != hipSuccess) { why("hipMemCreate", rc); return false; }
        rc = hipMemMap(e->arena + e->arena_cap, piece, 0, h, 0);
        if (rc != hipSuccess) { why("hipMemMap", rc); (void)hipMemRelease(h); return false; }
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        // (access is set for the whole mapped range from the base: inside this engine the runtime refuses the new piece alone — hipErrorInvalidValue for every
        //  piece behind the first, of any size — although profiles/tools/vmm_probe.hip, the same calls in a bare program, is granted it)
        rc = hipMemSetAccess(e->arena, e->arena_cap + piece, &acc, 1);
        if (rc != hipSuccess) { why("hipMemSetAccess", rc); (void)hipMemUnmap(e->arena + e->arena_cap, piece); (void)hipMemRelease(h); return false; }
        e->vmm_parts.push_back({h, piece});
        e->arena_cap += piece; // (pieces mapped before a failure stay: the caller sees the arena short of `need` and falls back)
        delta -= piece;
    }
    return true;
}
static void free_arena(prach_engine *e) {
    if (e->vmm == 1) {
        size_t at = 0;
        for (auto &pt : e->vmm_parts) { (void)hipMemUnmap(e->arena + at, pt.second); (void)hipMemRelease(pt.first); at += pt.second; }
        e->vmm_parts.clear();
        if (e->arena) (void)hipMemAddressFree(e->arena, e->vmm_reserved);
    } else if (e->arena) (void)hipFree(e->arena);
    e->arena = nullptr; e->arena_cap = 0;
}

int ensure_arena(prach_engine *e, size_t need) {
    if (need <= e->arena_cap) return PRACH_OK;
    if (e->vmm >= 0 && !e->opt_plain_arena) {
        if (grow_vmm_arena(e, need) && need <= e->arena_cap) return PRACH_OK;
        if (e->vmm == 1) { // the range is in use but could not grow (out of memory / of range): back to one plain allocation
            std::fprintf(stderr, "[prach] the reserved-range arena could not grow to %zu bytes: falling back to hipMalloc\n", need);
            free_arena(e);
        }
        e->vmm = -1;
    }
    const size_t had = e->arena_cap;
    if (e->arena) HIPCHK(hipFree(e->arena));
    e->arena = nullptr;
    e->arena_cap = 0;
    // (a sweep's calls grow point by point: growing by an eighth re-allocated at nearly every point, and one hipFree + hipMalloc of ~16 GB measured 1.4 s)
    const size_t want = std::max(need + (need >> 2), std::min<size_t>(2 * had, (size_t)64 << 30));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&e->arena), want));
    e->arena_cap = want;
    return PRACH_OK;
}
int ensure_pinned(prach_engine *e, size_t need) {
    if (need <= e->pinned_cap) return PRACH_OK;
    if (e->pinned) HIPCHK(hipHostFree(e->pinned));
    e->pinned = nullptr;
    e->pinned_cap = 0;
    const size_t want = need + (need >> 2) + 4096;
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&e->pinned), want, hipHostMallocDefault));
    e->pinned_cap = want;
    return PRACH_OK;
}

int host_threads(const prach_engine *e) {
    int n = e->opt_host_threads > 0 ? (int)e->opt_host_threads : (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(n, 64));
}

// Workgroups of the given cluster launch that can be resident at once on this device: the runtime's occupancy answer for
// the kernel and its dynamic LDS size, times the CU count (MI355X_MICROARCH.md "Residency and cooperative launch": a
// plain launch has the same residency as a cooperative one; the query is what a cooperative launch would check).
int resident_workgroups(const prach_engine *e, int per_cu) {
    if (e->opt_resident > 0) return (int)e->opt_resident;
    return std::max(1, per_cu) * e->num_cus;
}

} // namespace

static int engine_create_impl(int device, prach_engine **out) {
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
    prach_engine *e = new (std::nothrow) prach_engine();
    if (!e) return PRACH_ERR_DEVICE;
    e->device = device;
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e->mem_budget = (size_t)prop.totalGlobalMem / 4 * 3;
    int rc = [&]() -> int {
        HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&e->ev0));
        HIPCHK(hipEventCreate(&e->ev1));
        return PRACH_OK;
    }();
    if (rc != PRACH_OK) { prach_engine_destroy(e); return rc; } // (nothing half-built is leaked)
    *out = e;
    return PRACH_OK;
}

// one launch over the trials idx[0..m) (all the same rng_mode); attempt = glibc stream retry level
// LDS-resident records (prach_cluster.hip REC_L16): a Philox cluster whose owned UE slots (the launch's maximum) fit LDS next to
// the per-subframe structures.  Returns the slot count per workgroup, 0 = records stay in global memory.
static int lds_record_slots(const prach_engine *e, const prach_cfg *cfgs, const int *idx, int m, int G, int maxP, int *maxgroups) {
    *maxgroups = 0;
    if (G <= 1 || !e->opt_lds_records || e->opt_dense || cfgs[idx[0]].variant == PRACH_VARIANT_NOMA_C) return 0;
    int lslots = 0;
    for (int k = 0; k < m; k++) {
        const int groups = (cfgs[idx[k]].nUE + 63) / 64;
        lslots = std::max(lslots, (groups + G - 1) / G * 64);
        *maxgroups = std::max(*maxgroups, groups);
    }
    if (cfgs[idx[0]].rng_mode == PRACH_RNG_GLIBC) // the reference's rand() stream: only the lean kernel keeps such a cluster's records in LDS
        return (lslots <= CLUSTER_LQCAP && e->opt_fast && e->opt_pipeline && maxP <= lcluster_max_preambles() && *maxgroups <= lcluster_max_groups_glibc() &&
                lcluster_kernel_lds_bytes(lslots, true, *maxgroups) <= CLUSTER_LDS_LIMIT) ? lslots : 0;
    return (lslots <= CLUSTER_LQCAP && cluster_kernel_lds_bytes(maxP, false, lslots) <= CLUSTER_LDS_LIMIT) ? lslots : 0;
}
// ... on the lean kernel (prach_lcluster.hip): pipelined compacted pass only, nPreamble <= 64
static bool use_fast_kernel(const prach_engine *e, int lslots, int maxP) {
    return lslots > 0 && e->opt_fast && e->opt_pipeline && maxP <= lcluster_max_preambles() && lcluster_kernel_lds_bytes(lslots) <= CLUSTER_LDS_LIMIT;
}

// one workgroup per trial: can prach::batch_kernel (prach_batch.hip) run this trial?  (Philox: both workgroup shapes; the reference's own rand()
// stream: the 1024-thread shape with its per-group call marks in LDS, up to 131 072 UEs)
static bool batch_eligible(const prach_engine *e, const prach_cfg &c) {
    if (c.variant == PRACH_VARIANT_NOMA_C || !e->opt_batch || e->opt_dense || e->opt_wide_records) return false;
    const bool glibc = c.rng_mode == PRACH_RNG_GLIBC;
    return c.nPreamble <= batch_max_preambles() && c.maxRarWindow <= batch_max_rar_window() && batch_calendar_slots(c.backoff, c.accessTime, c.maxRarWindow) <= batch_max_calendar_slots() &&
           (int64_t)prach_max_time(&c) + c.backoff + c.accessTime + 128 < batch_max_subframes() && c.nUE < (1 << 20) - 1 && (c.nUE + 63) / 64 <= batch_max_groups(glibc);
}

// NOMA.c's activeUE table on the device: noma_activation_kernel for every UE of the launch, then the few UEs whose value sits within the
// device math library's error band of a rounding / comparison boundary (~1e-6 of them) are recomputed with the host's libm
// (prach_noma_activation_range) and patched in.  dparams: the launch's TrialDev blocks (device), tabs[k]: trial k's table arrays (device).
struct ActTab { char *pre0, *sec, *gain, *lgain, *nd0; };
constexpr int NOMA_ACT_OVERFLOW_RC = -1099; // (internal) noma_activation_kernel flagged more UEs than its list holds
static int noma_device_activation(prach_engine *e, const TrialDev *dparams, const prach_cfg *cfgs, const int *idx, int m, const std::vector<ActTab> &tabs,
                                  unsigned *dflags, std::vector<std::pair<int, int>> *flagged) {
    int maxUE = 0;
    for (int k = 0; k < m; k++) maxUE = std::max(maxUE, cfgs[idx[k]].nUE);
    HIPCHK(launch_noma_activation(dparams, m, maxUE, dflags, e->stream));
    unsigned nflag = 0;
    HIPCHK(hipMemcpyAsync(&nflag, dflags, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (nflag > (unsigned)NOMA_ACT_FLAG_CAP) { // (e.g. a cell radius just above the 35 m exclusion zone: the redraw loop's iteration limit flags every tenth UE)
        std::fprintf(stderr, "[prach] noma_activation_kernel flagged %u UEs (more than its list holds): the activation table of this launch is built on the host\n", nflag);
        return NOMA_ACT_OVERFLOW_RC;
    }
    if (!nflag) return PRACH_OK;
    std::vector<unsigned> fl(2 * (size_t)nflag);
    HIPCHK(hipMemcpyAsync(fl.data(), dflags + 2, 8 * (size_t)nflag, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    // the recomputed values are staged in buffers that live until the copies have completed ON THE ENGINE'S STREAM (the simulation kernel is launched on that
    // stream, which is non-blocking: a copy on the null stream would not be ordered in front of it)
    struct Patch { int32_t pre0, sec; double gn, lg; uint32_t nd0; };
    std::vector<Patch> pt(nflag);
    for (unsigned q = 0; q < nflag; q++) {
        const int k = (int)fl[2 * q], i = (int)fl[2 * q + 1];
        if (k < 0 || k >= m || i < 0 || i >= cfgs[idx[k]].nUE) return PRACH_ERR_INTERNAL;
        const int arc = prach_noma_activation_range(&cfgs[idx[k]], i, i + 1, &pt[q].pre0, &pt[q].sec, &pt[q].gn, &pt[q].lg, &pt[q].nd0);
        if (arc != PRACH_OK) return arc;
        const ActTab &T = tabs[k];
        HIPCHK(hipMemcpyAsync(T.pre0 + 4 * (size_t)i, &pt[q].pre0, 4, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(T.sec + 4 * (size_t)i, &pt[q].sec, 4, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(T.gain + 8 * (size_t)i, &pt[q].gn, 8, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(T.lgain + 8 * (size_t)i, &pt[q].lg, 8, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(T.nd0 + 4 * (size_t)i, &pt[q].nd0, 4, hipMemcpyHostToDevice, e->stream));
        if (flagged) flagged->push_back({k, i});
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    e->noma_flagged += (int)nflag;
    return PRACH_OK;
}

static int run_group(prach_engine *e, const prach_cfg *cfgs, const int *idx, int m, prach_result *results,
                     prach_ue_log *const *ue_logs, int attempt, int G, double &kernel_ms, double &upload_ms) {
    const int rng_mode = cfgs[idx[0]].rng_mode;
    const bool noma = cfgs[idx[0]].variant == PRACH_VARIANT_NOMA_C;
    std::vector<size_t> slen(m, 0);
    int maxP = 1;
    for (int k = 0; k < m; k++) {
        const prach_cfg &c = cfgs[idx[k]];
        if (rng_mode == PRACH_RNG_GLIBC) slen[k] = (size_t)stream_budget(c, attempt, e->opt_stream_factor, e->draws_per_ue_seen);
        if (c.nPreamble > maxP) maxP = c.nPreamble;
    }
    // one workgroup per trial, Philox: the batch kernel (prach_batch.hip), within its limits
    bool batch = G == 1;
    for (int k = 0; k < m && batch; k++) batch = batch_eligible(e, cfgs[idx[k]]);
    for (int k = 0; k < m; k++) // the dormant per-sector grant path exists in trial_kernel (G == 0) and in the batch kernel only
        if ((cfgs[idx[k]].flags & PRACH_FLAG_SECTOR_GRANTS) && G > 0 && !batch) return PRACH_ERR_INTERNAL;
    // NOMA.c's activeUE table: built by the device (noma_activation_kernel) unless the option or a rerun asks for the host's libm
    const bool host_act = noma && (e->opt_noma_host_activation || e->force_host_act);
    const LaunchLayout LL = layout_launch(cfgs, idx, m, ue_logs, slen, G, batch, e->full_calendars, e->opt_calendar_cap);
    if (LL.end > e->mem_budget && m > 1) { // (e.g. the 10 000-trial grid with its calendars on ONE GPU: two or three launches instead of one)
        const int h = m / 2;
        int rc = run_group(e, cfgs, idx, h, results, ue_logs, attempt, G, kernel_ms, upload_ms);
        if (rc != PRACH_OK) return rc;
        return run_group(e, cfgs, idx + h, m - h, results, ue_logs, attempt, G, kernel_ms, upload_ms);
    }
    { int rc = ensure_arena(e, LL.end); if (rc != PRACH_OK) return rc; }
    { int rc = ensure_pinned(e, std::max(LL.staged_end, sizeof(DevResult) * (size_t)m)); if (rc != PRACH_OK) return rc; }
    auto t0 = std::chrono::steady_clock::now();
    char *const A = e->arena;
    char *const H = e->pinned;
    TrialDev *const td = reinterpret_cast<TrialDev *>(H);
    std::vector<int32_t> nAccess(m, 0);
    for (int k = 0; k < m; k++) {
        const prach_cfg &c = cfgs[idx[k]];
        const TrialLayout &L = LL.t[k];
        TrialDev &d = td[k];
        std::memset(&d, 0, sizeof(d));
        d.variant = c.variant; d.uniform = c.uniform; d.nUE = c.nUE; d.nP = c.nPreamble; d.backoff = c.backoff;
        d.nGrantUL = c.nGrantUL; d.maxRarWindow = c.maxRarWindow; d.maxMsg2 = c.maxMsg2TxCount; d.aT = c.accessTime;
        d.rng_mode = c.rng_mode; d.maxTime = prach_max_time(&c);
        d.stop = (c.max_steps > 0 && c.max_steps < d.maxTime) ? c.max_steps : d.maxTime;
        d.seed_lo = (unsigned)c.seed; d.seed_hi = (unsigned)(c.seed >> 32);
        d.stream_len = L.stream_len;
        d.dense_pass = e->opt_dense ? 1 : 0;
        d.pipeline = e->opt_pipeline ? 1 : 0;
        if (batch) {
            d.rec32 = reinterpret_cast<int4 *>(A + L.rec32); d.chunks = reinterpret_cast<int4 *>(A + L.chunks); d.ctab = reinterpret_cast<int *>(A + L.ctab); d.cpool = reinterpret_cast<int *>(A + L.cpool); d.nchunks = L.npool; d.tcap = L.tcap;
            d.jcal = reinterpret_cast<int *>(A + L.jcal); d.calcap = L.calcap; d.calmask = L.calslots - 1;
            d.qov = reinterpret_cast<int *>(A + L.qov); d.evov = reinterpret_cast<int2 *>(A + L.evov);
        } else {
            d.rec = reinterpret_cast<int4 *>(A + L.rec);
            d.ptc = reinterpret_cast<int *>(A + L.ptc); d.ftt = reinterpret_cast<int *>(A + L.ftt);
            d.stt = reinterpret_cast<int *>(A + L.stt); d.fcnt = reinterpret_cast<int *>(A + L.fcnt);
            d.nd = reinterpret_cast<unsigned *>(A + L.nd);
        }
        d.evbuf = L.evbuf ? reinterpret_cast<Event *>(A + L.evbuf) : nullptr;
        d.evbuf2 = L.evbuf2 ? reinterpret_cast<Event *>(A + L.evbuf2) : nullptr;
        d.sidx = L.sidx ? reinterpret_cast<int *>(A + L.sidx) : nullptr;
        d.sched = reinterpret_cast<const int *>(A + L.sched);
        d.stream = L.stream ? reinterpret_cast<const int *>(A + L.stream) : nullptr;
        d.logs = L.logs ? reinterpret_cast<prach_ue_log *>(A + L.logs) : nullptr;
        d.timers = reinterpret_cast<int *>(A + L.timers);
        d.out = reinterpret_cast<DevResult *>(A + L.out);
        d.evw = L.evw; d.mbstride = L.mbstride;
        d.binshift = 0;
        while (((c.nUE - 1) >> d.binshift) >= 1024) d.binshift++;
        d.mbox = L.mbox ? reinterpret_cast<int *>(A + L.mbox) : nullptr;
        d.cand = L.cand ? reinterpret_cast<int2 *>(A + L.cand) : nullptr;
        d.flags = c.flags;
        d.diag = L.diag ? reinterpret_cast<unsigned long long *>(A + L.diag) : nullptr;
        d.sector = L.sector ? reinterpret_cast<int *>(A + L.sector) : nullptr;
        int32_t *sched = reinterpret_cast<int32_t *>(H + L.sched);
        // the arrival table depends on (nUE, traffic law, accessTime) only: a sweep x seeds batch has a handful of distinct ones
        // (2000 pow() calls each) — copy the previous trial's table when its key is the same
        int same = -1;
        for (int q = k - 1; q >= 0 && q >= k - 16; q--) {
            const prach_cfg &o = cfgs[idx[q]];
            if (o.nUE == c.nUE && o.uniform == c.uniform && o.accessTime == c.accessTime && o.variant == c.variant) { same = q; break; }
        }
        if (same >= 0) {
            std::memcpy(sched, H + LL.t[same].sched, 4 * L.sched_len);
            nAccess[k] = nAccess[same];
        } else {
            for (size_t s = 0; s < L.sched_len; s++) sched[s] = c.nUE;
            prach_arrival_schedule(&c, sched, (int)L.sched_len, &nAccess[k]);
        }
        if (rng_mode == PRACH_RNG_GLIBC)
            // the reference's rand() stream window [stream_offset, +stream_len): the host only jumps ahead (31-word window
            // per chunk, cached matrix powers); the values themselves are generated on the device inside the timed region
            reinterpret_cast<StreamJob *>(H + LL.stream_jobs)[k] = StreamJob{reinterpret_cast<const unsigned *>(A + L.seeds), reinterpret_cast<int *>(A + L.stream), (unsigned long long)L.stream_len};
        if (noma) {
            d.n_pre0 = reinterpret_cast<const int *>(A + L.n_pre0); d.n_sector = reinterpret_cast<const int *>(A + L.n_sector);
            d.n_gain = reinterpret_cast<const double *>(A + L.n_gain); d.n_lgain = reinterpret_cast<const double *>(A + L.n_lgain);
            d.n_nd0 = reinterpret_cast<const unsigned *>(A + L.n_nd0);
            d.cell_radius = c.cellRadius;
            d.n_devact = host_act ? 0 : (e->opt_noma_ambiguity_test ? 2 : 1);
        }
    }
    if (rng_mode == PRACH_RNG_GLIBC) { // one 31-word window per chunk of every trial's stream window: the trials dealt to the host cores (100 trials x 200 jump-aheads: 25 ms on one)
        const int nth = std::min(host_threads(e), std::max(1, m));
        auto work = [&](int tix) {
            for (int k = tix; k < m; k += nth) {
                const prach_cfg &c = cfgs[idx[k]];
                const TrialLayout &L = LL.t[k];
                prach_internal_glibc_seeds((uint32_t)c.seed, c.stream_offset, L.nchunks, STREAM_CHUNK, reinterpret_cast<uint32_t *>(H + L.seeds));
            }
        };
        std::vector<std::thread> th;
        for (int tix = 1; tix < nth; tix++) th.emplace_back(work, tix);
        work(0);
        for (auto &x : th) x.join();
    }
    if (noma && host_act) {
        // activeUE's per-UE attributes (NOMA.c:131-192: double-precision libm work, once per UE): built on the host with the
        // libm the reference links, UE ranges of all trials dealt to all host cores, then copied trial by trial
        const int nth = host_threads(e);
        struct Tab { std::vector<int32_t> pre0, sec; std::vector<double> gn, lg; std::vector<uint32_t> nd0; };
        std::vector<Tab> tabs(m);
        struct Job { int k, lo, hi; };
        std::vector<Job> jobs;
        for (int k = 0; k < m; k++) {
            const int n = cfgs[idx[k]].nUE;
            tabs[k].pre0.resize(n); tabs[k].sec.resize(n); tabs[k].gn.resize(n); tabs[k].lg.resize(n); tabs[k].nd0.resize(n);
            const int step = std::max(2048, (n + nth - 1) / nth);
            for (int lo = 0; lo < n; lo += step) jobs.push_back({k, lo, std::min(n, lo + step)});
        }
        std::vector<int> jrc(jobs.size(), PRACH_OK);
        auto work = [&](int tix) {
            for (size_t j = (size_t)tix; j < jobs.size(); j += (size_t)nth) {
                const Job &J = jobs[j];
                Tab &T = tabs[J.k];
                jrc[j] = prach_noma_activation_range(&cfgs[idx[J.k]], J.lo, J.hi, T.pre0.data() + J.lo, T.sec.data() + J.lo, T.gn.data() + J.lo,
                                                     T.lg.data() + J.lo, T.nd0.data() + J.lo);
            }
        };
        std::vector<std::thread> th;
        for (int tix = 1; tix < nth; tix++) th.emplace_back(work, tix);
        work(0);
        for (auto &x : th) x.join();
        for (int r : jrc) if (r != PRACH_OK) return r;
        for (int k = 0; k < m; k++) {
            const size_t nn = (size_t)cfgs[idx[k]].nUE;
            const TrialLayout &L = LL.t[k];
            HIPCHK(hipMemcpyAsync(A + L.n_pre0, tabs[k].pre0.data(), 4 * nn, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(A + L.n_sector, tabs[k].sec.data(), 4 * nn, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(A + L.n_gain, tabs[k].gn.data(), 8 * nn, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(A + L.n_lgain, tabs[k].lg.data(), 8 * nn, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(A + L.n_nd0, tabs[k].nd0.data(), 4 * nn, hipMemcpyHostToDevice, e->stream));
        }
        HIPCHK(hipStreamSynchronize(e->stream)); // (the tables are pageable host memory that goes away)
    }
    // ONE copy stages every parameter block, arrival table and stream seed of the launch; ONE memset zeroes everything the
    // kernels accumulate into or poll (DevResult blocks; mailbox tags: 0 never equals t + 1)
    HIPCHK(hipMemcpyAsync(A, H, LL.staged_end, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemsetAsync(A + LL.zero_begin, 0, LL.zero_end - LL.zero_begin, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    auto t1 = std::chrono::steady_clock::now();
    upload_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();

    HIPCHK(hipEventRecord(e->ev0, e->stream));
    if (rng_mode == PRACH_RNG_GLIBC) { // every trial's stream window, ONE launch (a launch per trial: 0.45 ms each, prach_stream.hip)
        unsigned long long max_n = 0;
        for (int k = 0; k < m; k++) max_n = std::max(max_n, (unsigned long long)LL.t[k].stream_len);
        HIPCHK(launch_glibc_stream_jobs(reinterpret_cast<const StreamJob *>(A + LL.stream_jobs), m, max_n, e->stream));
    }
    if (noma && !host_act) { // activeUE for every UE of the launch, inside the timed region
        std::vector<ActTab> tabs(m);
        for (int k = 0; k < m; k++) tabs[k] = {A + LL.t[k].n_pre0, A + LL.t[k].n_sector, A + LL.t[k].n_gain, A + LL.t[k].n_lgain, A + LL.t[k].n_nd0};
        int rc = noma_device_activation(e, reinterpret_cast<const TrialDev *>(A), cfgs, idx, m, tabs, reinterpret_cast<unsigned *>(A + LL.act_flags), nullptr);
        if (rc == NOMA_ACT_OVERFLOW_RC && !e->force_host_act) { // the same launch once more with the host-built table (the path a NOMA_AMBIGUOUS rerun takes)
            e->force_host_act = true;
            rc = run_group(e, cfgs, idx, m, results, ue_logs, attempt, G, kernel_ms, upload_ms);
            e->force_host_act = false;
            return rc;
        }
        if (rc != PRACH_OK) return rc;
    }
    if (noma) {
        const int xpack = e->opt_xcd_pack && !e->pack_off && G > 1 && ((m + 7) / 8) * G <= e->num_cus / 8;
        e->last.xcd_packed = xpack;
        HIPCHK(launch_noma_kernel(reinterpret_cast<const TrialDev *>(A), m, G, maxP, xpack, e->stream));
    }
    else if (batch) {
        e->last.rec_mode = CLUSTER_REC_BATCH;
        e->last.xcd_packed = 0;
        // Workgroup shape (speed only, same results).  A launch is as long as its longest trial alone or as its work needs, whichever is more: a 100 000-UE
        // trial runs 127 ms on 16 wavefronts and 152 ms on 8, so while the trials fit the CUs in two rounds the 1024-thread shape wins (510 sweep trials:
        // 159 / 272 ms — Beta.c / WithNOMA — against 163 / 290 ms); past two per CU a third round starts, and two 512-thread workgroups = two trials per CU,
        // which hide each other's barriers, win (560 trials: 165 / 288 against 171 / 294 ms; 740: 171 / 327 against 218 / 395; 1000: 207 / 442 against
        // 288 / 517; 2000: 381 / 811 against 566 / 1 005): scripts/gpu_shape_probe.sh.
        int waves = (int)e->opt_batch_waves;
        if (rng_mode == PRACH_RNG_GLIBC) waves = 16;
        if (waves == 0) waves = m > 2 * e->num_cus ? 8 : 16;
        e->last.workgroups = m;
        HIPCHK(launch_batch_kernel(reinterpret_cast<const TrialDev *>(A), m, waves, rng_mode == PRACH_RNG_GLIBC, e->stream));
    }
    else if (G > 0) {
        // one workgroup per trial in the reference's rand() stream: 8 + 4 byte hot records, if every subframe number of every trial of
        // the launch fits 16 bits (txTime <= t + 59 + backoff + accessTime).  (Philox with one workgroup per trial is the batch kernel's.)
        bool compact = G == 1 && !e->opt_wide_records && rng_mode == PRACH_RNG_GLIBC;
        for (int k = 0; k < m && compact; k++) {
            const prach_cfg &c = cfgs[idx[k]];
            compact = (int64_t)prach_max_time(&c) + c.backoff + c.accessTime + 64 < 63000;
        }
        int maxgroups = 0;
        const int lslots = lds_record_slots(e, cfgs, idx, m, G, maxP, &maxgroups);
        const bool glibc = rng_mode == PRACH_RNG_GLIBC;
        const int rec_mode = (glibc ? lslots > 0 : use_fast_kernel(e, lslots, maxP)) ? CLUSTER_REC_LFAST : (lslots > 0 ? CLUSTER_REC_L16 : (compact ? CLUSTER_REC_H8 : CLUSTER_REC_G16));
        e->last.rec_mode = rec_mode;
        // XCD-packed launch (prach_lcluster.hip): each cluster on one XCD, eight clusters side by side — when the clusters of the launch fit
        // the XCDs' CUs that way (budgeted at one workgroup per CU: LDS-resident state fills a CU, the general layouts take more than half)
        const int xpack = e->opt_xcd_pack && !e->pack_off && G > 1 && rec_mode != CLUSTER_REC_H8 && ((m + 7) / 8) * G <= e->num_cus / 8;
        e->last.xcd_packed = xpack;
        if (rec_mode == CLUSTER_REC_LFAST) HIPCHK(launch_lcluster_kernel(reinterpret_cast<const TrialDev *>(A), m, G, lslots, xpack, glibc, maxgroups, e->stream));
        else {
            HIPCHK(launch_cluster_kernel(reinterpret_cast<const TrialDev *>(A), m, G, maxP, rng_mode, rec_mode, lslots, xpack, e->stream));
        }
    }
    else HIPCHK(launch_trial_kernel(reinterpret_cast<const TrialDev *>(A), m, rng_mode, maxP, e->stream));
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    HIPCHK(hipMemcpyAsync(H, A + LL.out0, sizeof(DevResult) * (size_t)m, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    kernel_ms += ms;
    e->last.launches++;
    e->last.workgroups = m * (G > 0 ? G : 1);
    e->last.cluster_size = G;

    const DevResult *const drs = reinterpret_cast<const DevResult *>(H);
    std::vector<int32_t> timers;
    for (int k = 0; k < m; k++) {
        const prach_cfg &c = cfgs[idx[k]];
        const TrialLayout &L = LL.t[k];
        const DevResult &dr = drs[k];
        prach_result &r = results[idx[k]];
        std::memset(&r, 0, sizeof(r));
        r.status = (dr.hard_error && dr.status == PRACH_ERR_TIMEOUT) ? PRACH_ERR_INTERNAL : dr.status; // (a capacity overflow somewhere in the cluster is the cause, a peer's time-out its effect)
        if (noma && dr.hard_error == NOMA_AMBIGUOUS && r.status == PRACH_OK) { r.status = PRACH_ERR_INTERNAL; e->noma_ambiguous++; } // (rerun with the host-built table: run_trials)
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
        e->last.group_visits += dr.visits;
        e->last.event_ues += dr.events;
        if (c.variant == PRACH_VARIANT_NOMA_C && dr.nSuccess == c.nUE && dr.dbg[0] > 0) { // all UEs succeeded: NOMA.c:707-710 breaks there
            r.time_exit = (int32_t)dr.dbg[0] - 1;
            r.steps = dr.dbg[0];
        }
        if (std::getenv("PRACH_PRINT_STAMPS"))
            std::fprintf(stderr, "[prach stamps/step] pass=%.0f publish=%.0f barrier=%.0f gather=%.0f r1b=%.0f leavers=%.0f checks=%.0f grants=%.0f cycles | N avg %.1f max %llu, resetcand avg %.2f, singles avg %.1f max %llu\n",
                         dr.stamps6[0] / (double)dr.steps, dr.stamps6[1] / (double)dr.steps, dr.stamps6[2] / (double)dr.steps, dr.stamps6[3] / (double)dr.steps,
                         dr.stamps6[4] / (double)dr.steps, dr.stamps6[5] / (double)dr.steps, dr.stamps6[6] / (double)dr.steps, dr.stamps6[7] / (double)dr.steps,
                         dr.dbg[0] / (double)dr.steps, dr.dbg[1], dr.dbg[2] / (double)dr.steps, (dr.dbg[3] >> 20) / (double)dr.steps, dr.dbg[3] & 0xfffff);
        if (std::getenv("PRACH_PRINT_STAMPS")) {
            static const char *const nm[24] = {"head", "phaseB", "S1", "leavers", "S2", "publish", "window-rest", "take1", "S3", "round2", "S4", "calls", "S5", "grants", "S6", "phaseA",
                                               "w:phaseA|A-barrier", "w:loads", "w:refill", "t:buckets", "-", "-", "-", "-"};
            static const char *const nmb[24] = {"head", "joins", "-", "body", "S1", "leavers", "S2", "classify", "S4", "calls", "S5", "grants", "S6", "-", "-", "-",
                                                "-", "-", "-", "-", "-", "-", "-", "-"}; // prach_batch.hip
            static const char *const nmn[24] = {"head", "publish+gather", "gather-barrier", "resolve(w0)", "resolve-barrier", "passB+A(w0)", "pass-barrier", "-", "r:to-gains", "r:rank+sort", "r:pairing", "-", "-", "-", "-", "-",
                                                "-", "-", "-", "-", "-", "-", "-", "-"}; // prach_noma.hip, per SUBFRAME (x accessTime = per slot)
            const char *const *const names = noma ? nmn : e->last.rec_mode == CLUSTER_REC_BATCH ? nmb : nm;
            std::fprintf(stderr, "[prach fine stamps/step] nUE=%d steps=%llu", c.nUE, (unsigned long long)dr.steps);
            for (int q = 0; q < 20; q++) if (names[q][0] != '-') std::fprintf(stderr, " %s=%.0f", names[q], dr.fstamps[q] / (double)dr.steps);
            std::fprintf(stderr, "\n");
        }
        if (L.diag && e->last.rec_mode == CLUSTER_REC_LFAST && std::getenv("PRACH_PRINT_STAMPS") && std::atoi(std::getenv("PRACH_PRINT_STAMPS")) >= 2) {
            // every workgroup of the cluster on its own clock (thread PRACH_STAMP_TID): cycles per subframe and phase, then its event UEs per subframe
            std::vector<unsigned long long> dg(32 * (size_t)G);
            HIPCHK(hipMemcpy(dg.data(), A + L.diag, 8 * dg.size(), hipMemcpyDeviceToHost));
            static const char *const nm[20] = {"head", "phaseB", "S1", "leavers", "S2", "publish", "window-rest", "take1", "S3", "round2", "S4", "calls", "S5", "grants", "S6", "-",
                                               "w:phaseA", "w:loads", "w:refill", "t:buckets"};
            for (int b_ = 0; b_ < G; b_++) {
                std::fprintf(stderr, "[prach wg stamps/step] nUE=%d steps=%llu b=%d", c.nUE, (unsigned long long)dr.steps, b_);
                for (int q = 0; q < 20; q++) if (nm[q][0] != '-') std::fprintf(stderr, " %s=%.0f", nm[q], dg[32 * (size_t)b_ + q] / (double)dr.steps);
                std::fprintf(stderr, " queued=%.2f late-buckets=%.3f late-header=%.3f refills=%.2f\n", dg[32 * (size_t)b_ + 24] / (double)dr.steps, dg[32 * (size_t)b_ + 25] / (double)dr.steps,
                             dg[32 * (size_t)b_ + 26] / (double)dr.steps, dg[32 * (size_t)b_ + 27] / (double)dr.steps);
            }
        }
        if (dr.status != PRACH_OK && e->last.rec_mode == CLUSTER_REC_LFAST && std::getenv("PRACH_VERBOSE"))
            std::fprintf(stderr, "[prach] lcluster_kernel: trial nUE=%d left at subframe %d with status %d, capacity code %d\n", c.nUE, dr.time_exit, dr.status, dr.hard_error);
        if (batch && dr.status == PRACH_ERR_INTERNAL && (dr.hard_error == 5 || dr.hard_error == 8 || dr.hard_error == 9) && !e->full_calendars) e->cal_overflow.push_back(idx[k]);
        if (dr.status == PRACH_ERR_INTERNAL && e->last.rec_mode == CLUSTER_REC_BATCH && std::getenv("PRACH_VERBOSE"))
            std::fprintf(stderr, "[prach] batch_kernel: trial nUE=%d (P %d B %d G %d R %d M %d A %d u %d) left at subframe %d: capacity %d (2 reset-cycle candidates, 3 singleton callers, 4 crossing bin, 5 a join list, 6 the grant notes, 8 a chunk table, 9 the chunk pool)\n", c.nUE, c.nPreamble, c.backoff, c.nGrantUL, c.maxRarWindow, c.maxMsg2TxCount, c.accessTime, c.uniform, dr.time_exit, dr.hard_error);
        if (dr.status != PRACH_OK) continue;
        // totalDelay is a FLOAT running sum in index order (Beta.c:186,193): exact in integer
        // arithmetic while it stays below 2^24, otherwise replay the float additions on the host.
        if (dr.sumTimer < (1ll << 24)) {
            r.totalDelay = (float)dr.sumTimer;
        } else {
            timers.resize((size_t)c.nUE);
            HIPCHK(hipMemcpy(timers.data(), A + L.timers, 4 * (size_t)c.nUE, hipMemcpyDeviceToHost));
            float td_ = 0;
            for (int i = 0; i < c.nUE; i++)
                if (timers[i] != INT_MIN) td_ += (float)timers[i];
            r.totalDelay = td_;
        }
        if (L.logs)
            HIPCHK(hipMemcpy(ue_logs[idx[k]], A + L.logs, sizeof(prach_ue_log) * (size_t)c.nUE, hipMemcpyDeviceToHost));
    }
    return PRACH_OK;
}

// A trial that a cluster launch could not finish — a per-subframe LDS capacity exceeded (PRACH_ERR_INTERNAL) or a workgroup
// that waited too long for a peer (PRACH_ERR_TIMEOUT: the cluster's workgroups were not all resident, e.g. another process
// holds CUs) — is rerun, exactly, on a kernel that needs neither.  Never silently: counted in prach_timing and reported.
static void note_fallback(prach_engine *e, const char *what, size_t ntrials, size_t ntimeouts, int G) {
    e->last.fallback_trials += (int32_t)ntrials;
    e->last.spin_timeouts += (int32_t)ntimeouts;
    std::fprintf(stderr, "[prach] %zu trial(s) of a %d-workgroup cluster launch are rerun on %s (%zu exceeded a per-subframe capacity, %zu timed out "
                         "waiting for a peer workgroup: cluster not co-resident?)\n", ntrials, G, what, ntrials - ntimeouts, ntimeouts);
}

// Batch-kernel trials that filled a calendar list (a parameter set that synchronises more UEs onto one subframe than the list was sized for): once more on the
// same kernel with lists of nUE entries, which cannot fill.  Never silently: counted as fallback trials and reported.
static int rerun_full_calendars(prach_engine *e, const prach_cfg *cfgs, prach_result *results, prach_ue_log *const *ue_logs, double &kernel_ms, double &upload_ms) {
    if (e->cal_overflow.empty()) return PRACH_OK;
    std::vector<int> todo;
    todo.swap(e->cal_overflow);
    e->last.fallback_trials += (int32_t)todo.size();
    std::fprintf(stderr, "[prach] %zu trial(s) filled a calendar list of prach::batch_kernel: rerun with lists of nUE entries\n", todo.size());
    e->full_calendars = true;
    int rc = PRACH_OK;
    for (int attempt = 0; !todo.empty() && rc == PRACH_OK; attempt++) {
        if (attempt > 6) { rc = PRACH_ERR_STREAM; break; }
        rc = run_group(e, cfgs, todo.data(), (int)todo.size(), results, ue_logs, attempt, 1, kernel_ms, upload_ms);
        std::vector<int> again;
        for (int k : todo) if (results[k].status == PRACH_ERR_STREAM) again.push_back(k); // (the reference's stream: a larger window)
        todo.swap(again);
    }
    e->full_calendars = false;
    e->cal_overflow.clear();
    return rc;
}

static int run_trials_impl(prach_engine *e, const prach_cfg *cfgs, int n, prach_result *results, prach_ue_log *const *ue_logs) {
    for (int k = 0; k < n; k++) { // nothing is left uninitialised on an early error return
        std::memset(&results[k], 0, sizeof(results[k]));
        results[k].status = PRACH_ERR_INTERNAL;
    }
    for (int k = 0; k < n; k++) {
        int v = prach_cfg_validate(&cfgs[k]);
        if (v != PRACH_OK) return v;
        if (cfgs[k].variant == PRACH_VARIANT_NOMA_C && (cfgs[k].nPreamble > 64 || cfgs[k].uniform)) return PRACH_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(e->device));
    auto t0 = std::chrono::steady_clock::now();
    e->last = prach_timing{};
    e->noma_flagged = e->noma_ambiguous = 0;
    double kernel_ms = 0, upload_ms = 0;
    // NOMA.c in the reference's OWN rand() stream: activeUE's rejection loops make every stream position data dependent and its libm
    // calls must be the reference's, so the arrivals are activated on the host between device steps (prach_noma_glibc.hip): one trial
    // at a time, one launch per access slot — the bit-exact-vs-the-reference's-files mode, not the throughput mode
    {
        std::vector<int> todo;
        for (int k = 0; k < n; k++)
            if (cfgs[k].variant == PRACH_VARIANT_NOMA_C && cfgs[k].rng_mode == PRACH_RNG_GLIBC) todo.push_back(k);
        // window of trial k at retry level a: 48, 192, 768, 3072 values per UE (1.2 GB at nUE = 100 000 at most, then PRACH_ERR_STREAM)
        auto window = [&](int k, int a) { return ((unsigned long long)cfgs[k].nUE * 48ull + (1ull << 18)) << (2 * a); };
        // slot by slot, the arrivals activated by the host with the reference's libm (prach_noma_glibc.hip): the exact form a trial falls back to
        auto host_form = [&](int k) -> int {
            int rc = PRACH_ERR_STREAM;
            for (int attempt = 0; attempt < 4 && rc == PRACH_ERR_STREAM; attempt++) {
                const unsigned long long len = window(k, attempt);
                std::vector<int32_t> hs((size_t)len);
                prach_glibc_stream((uint32_t)cfgs[k].seed, cfgs[k].stream_offset, len, hs.data());
                rc = run_noma_glibc_trial(e->stream, cfgs[k], hs.data(), len, &results[k], ue_logs ? ue_logs[k] : nullptr, &kernel_ms);
                e->last.launches++;
            }
            return rc;
        };
        if (e->opt_noma_host_activation) {
            for (int k : todo) { int rc = host_form(k); if (rc != PRACH_OK) return rc; }
            todo.clear();
        }
        // the single-launch form: every trial of the call (the seeds of a sweep point) side by side, one workgroup each, activeUE on the device; a value
        // inside the device math library's error band (prach_noma_act.h: a few percent of the trials) sends that trial to the host form
        for (int attempt = 0; !todo.empty(); attempt++) {
            if (attempt >= 4) return PRACH_ERR_STREAM;
            const int m = (int)todo.size();
            std::vector<const prach_cfg *> pc(m);
            std::vector<unsigned long long> lens(m);
            std::vector<prach_result *> pr(m);
            std::vector<prach_ue_log *> pl(m);
            std::vector<int> rcs(m, PRACH_ERR_INTERNAL);
            for (int j = 0; j < m; j++) { pc[j] = &cfgs[todo[j]]; lens[j] = window(todo[j], attempt); pr[j] = &results[todo[j]]; pl[j] = ue_logs ? ue_logs[todo[j]] : nullptr; }
            if (e->opt_noma_ambiguity_test) std::fill(rcs.begin(), rcs.end(), NOMA_GLIBC_AMBIGUOUS_RC); // (test hook: as if the kernel had found a value inside the band)
            else {
                int rc = run_noma_glibc_batch(e->stream, pc.data(), m, lens.data(), pr.data(), pl.data(), &kernel_ms, rcs.data());
                e->last.launches++;
                if (rc != PRACH_OK) return rc;
            }
            std::vector<int> again;
            for (int j = 0; j < m; j++) {
                const int k = todo[j];
                if (rcs[j] == PRACH_OK) continue;
                if (rcs[j] == PRACH_ERR_STREAM) { again.push_back(k); continue; }
                if (rcs[j] != NOMA_GLIBC_AMBIGUOUS_RC) return rcs[j];
                e->noma_ambiguous++;
                e->last.fallback_trials++;
                if (std::getenv("PRACH_VERBOSE")) std::fprintf(stderr, "[prach] NOMA.c trial nUE=%d in the reference's stream: a value inside the device libm's error band, rerun with host-side activation\n", cfgs[k].nUE);
                int rc = host_form(k);
                if (rc != PRACH_OK) return rc;
            }
            todo.swap(again);
        }
    }
    { // NOMA.c variant (Philox): its own kernel, G workgroups per trial like the production kernel
        std::vector<int> idx;
        for (int k = 0; k < n; k++)
            if (cfgs[k].variant == PRACH_VARIANT_NOMA_C && cfgs[k].rng_mode == PRACH_RNG_PHILOX) idx.push_back(k);
        if (!idx.empty()) {
            int minGroups = INT_MAX, maxP = 1;
            bool small = true;
            for (int k : idx) {
                minGroups = std::min(minGroups, (cfgs[k].nUE + 63) / 64);
                small = small && cfgs[k].nUE < (1 << 20) - 1;
                maxP = std::max(maxP, cfgs[k].nPreamble);
            }
            int G = (int)e->opt_cluster;
            const size_t resident = (size_t)resident_workgroups(e, noma_kernel_blocks_per_cu(maxP));
            e->last.resident_limit = (int32_t)resident;
            // (measured at nUE = 100 000: 23.2 ms with 16 workgroups, 24.3 with 32, 26.6 with 8: 6 x nPreamble bins per mailbox)
            // (NOMA.c's own experiment — 10 seeds x the sweep = 100 trials in one call — measured 106 ms with one workgroup per trial, 61 ms with two:
            //  the clusters may fill the CUs the occupancy query admits, not half of them)
            if (G <= 0) { G = 1; while (G * 2 <= 16 && (size_t)G * 2 * idx.size() <= resident && G * 2 <= std::max(1, minGroups / 16)) G *= 2; }
            while (G > 1 && (size_t)G * idx.size() > resident) G /= 2;
            if (!small) G = 1; // 20-bit granule fields
            int rc = run_group(e, cfgs, idx.data(), (int)idx.size(), results, ue_logs, 0, G, kernel_ms, upload_ms);
            if (rc != PRACH_OK) return rc;
            std::vector<int> again;
            size_t nto = 0;
            for (int k : idx)
                if (results[k].status == PRACH_ERR_TIMEOUT || results[k].status == PRACH_ERR_INTERNAL) { again.push_back(k); nto += results[k].status == PRACH_ERR_TIMEOUT; }
            if (!again.empty() && (G > 1 || e->noma_ambiguous > 0)) { // one workgroup per trial waits for nobody; the host-built table needs no error band
                note_fallback(e, "noma_kernel with one workgroup per trial and the host-built activation table", again.size(), nto, G);
                e->force_host_act = true;
                rc = run_group(e, cfgs, again.data(), (int)again.size(), results, ue_logs, 0, 1, kernel_ms, upload_ms);
                e->force_host_act = false;
                if (rc != PRACH_OK) return rc;
            }
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
        // trials only trial_kernel runs (index-ordered, exact, one workgroup each): the dormant per-sector grant path, sizes beyond the
        // cluster kernels' 20-bit granule fields / group tables — they leave the others on the cluster kernels
        std::vector<int> solo;
        {
            std::vector<int> rest, sect;
            for (int k : idx) {
                const bool sector = (cfgs[k].flags & PRACH_FLAG_SECTOR_GRANTS) != 0;
                const bool only_trial_kernel = sector || cfgs[k].nUE >= (1 << 20) - 1 || (mode == PRACH_RNG_GLIBC && cfgs[k].nUE > CLUSTER_GLIBC_MAX_UE);
                (sector && !e->opt_legacy && batch_eligible(e, cfgs[k]) ? sect : only_trial_kernel ? solo : rest).push_back(k);
            }
            idx.swap(rest);
            if (!sect.empty()) {
                // the per-sector grant path in Philox mode: the batch kernel keeps the six budgets (one launch for these trials, one workgroup each;
                // what exceeds a capacity of its resolver goes on to trial_kernel like any other trial)
                int rc = run_group(e, cfgs, sect.data(), (int)sect.size(), results, ue_logs, 0, 1, kernel_ms, upload_ms);
                if (rc != PRACH_OK) return rc;
                rc = rerun_full_calendars(e, cfgs, results, ue_logs, kernel_ms, upload_ms);
                if (rc != PRACH_OK) return rc;
                size_t nbad = 0;
                for (int k : sect) if (results[k].status != PRACH_OK) { solo.push_back(k); nbad++; }
                if (nbad) { note_fallback(e, "trial_kernel (one workgroup per trial, no per-subframe capacity)", nbad, 0, 1); e->last.trial_kernel_reruns += (int32_t)nbad; }
            }
        }
        const bool cluster_ok = !e->opt_legacy && !idx.empty();
        int maxP = 1;
        for (int k : idx) maxP = std::max(maxP, cfgs[k].nPreamble);
        if (!cluster_ok) { idx.insert(idx.end(), solo.begin(), solo.end()); solo.clear(); }
        if (cluster_ok) {
            // production path: cluster kernel, G workgroups per trial.  The workgroups of a cluster wait for each other, so
            // every cluster of the launch must be resident: G x trials <= what the occupancy query admits for this kernel
            // and LDS size (members of a cluster are consecutive workgroups: in-order dispatch completes whole clusters)
            int minGroups = INT_MAX;
            for (int k : idx) minGroups = std::min(minGroups, (cfgs[k].nUE + 63) / 64);
            int G = (int)e->opt_cluster;
            // (every cluster layout of this library takes more than half a CU's LDS: one workgroup per CU.  Asked of the runtime for the
            //  general kernel's smallest layout; the lean kernel and the LDS-resident layouts only ever need more LDS, never admit more)
            const size_t resident = (size_t)resident_workgroups(e, std::min(cluster_kernel_blocks_per_cu(maxP, mode, CLUSTER_REC_G16, 0), 1));
            e->last.resident_limit = (int32_t)resident;
            if (G <= 0) {
                G = 1;
                // (the clusters of a call may fill the CUs: 8 trials of 100 000 UEs run 52 ms on 8 x 32 workgroups — the lean kernel, one XCD each —
                //  against 78 ms on 8 x 16, 16 trials 78 ms on 16 x 16 against 96 ms on 16 x 8: scripts/gpu_probe_cluster_sizes.py)
                while (G * 2 <= 32 && (size_t)G * 2 * idx.size() <= resident && G * 2 <= std::max(1, minGroups / 16)) G *= 2;
                // Uniform arrivals over 60 000 subframes (Beta.c:92-95): only nUE / 60 000 arrivals per subframe, a UE lives some
                // tens of subframes, finished groups are skipped 32 at a time — the live band is a few groups and one workgroup
                // steps through a subframe faster than a cluster exchanges (nUE = 100 000: 5.1 vs 6.2 us per subframe)
                // two workgroups per trial are not worth their exchange: 100 sweep trials run 151 / 242 ms (Beta.c / WithNOMA) on the batch kernel, one
                // workgroup each, against 221 / 471 ms on 2-workgroup clusters, whose halves of a 100 000-UE trial also overflow the 512 event
                // granules of a mailbox (32 of 100 trials rerun); from four workgroups per trial on, clusters win (scripts/gpu_probe_mid_batches.py)
                // (the reference's own stream, 100 trials of one sweep point: 246 / 431 ms on batch_kernel<16, true> against 657 / 2 126 ms — Beta.c / WithNOMA,
                //  nUE = 100 000, the latter with every trial overflowing its mailboxes — on 2-workgroup clusters: scripts/gpu_probe_glibc_batches.py)
                bool all_batch = e->opt_batch != 0;
                for (int k : idx) all_batch = all_batch && batch_eligible(e, cfgs[k]);
                if (G == 2 && all_batch) G = 1;
                // (in the reference's stream also against four workgroups per trial, which run on the general kernel: 50 trials 210 / 338 ms against 351 / 501 ms
                //  at nUE = 100 000, 73 / 100 against 84 / 108 ms at 20 000; from eight workgroups per trial on the clusters are level or ahead)
                if (G == 4 && all_batch && mode == PRACH_RNG_GLIBC) G = 1;
                // (round 4's batch kernel is level with four workgroups per trial in Philox mode as well: 60 sweep trials 126 / 185 ms — Beta.c / WithNOMA — against
                //  136 / 190 ms on 4-workgroup clusters; eight workgroups per trial stay ahead: 30 trials 95 / 118 against 126 / 184 ms)
                if (G == 4 && all_batch) G = 1;
                bool light = mode == PRACH_RNG_PHILOX;
                for (int k : idx) light = light && cfgs[k].uniform && cfgs[k].nUE <= 2000000;
                if (light) G = 1;
            }
            while (G > 1 && (size_t)G * idx.size() > resident) G /= 2;
            std::vector<int> todo = idx, fallback;
            size_t nto = 0;
            for (int attempt = 0; !todo.empty(); attempt++) {
                if (attempt > 6) return PRACH_ERR_STREAM;
                int rc = run_group(e, cfgs, todo.data(), (int)todo.size(), results, ue_logs, attempt, G, kernel_ms, upload_ms);
                if (rc != PRACH_OK) return rc;
                rc = rerun_full_calendars(e, cfgs, results, ue_logs, kernel_ms, upload_ms); // (one workgroup per trial on the batch kernel: a full calendar list)
                if (rc != PRACH_OK) return rc;
                std::vector<int> again;
                const bool was_packed = e->last.xcd_packed != 0;
                e->pack_off = false;
                for (int k : todo) {
                    if (results[k].status == PRACH_ERR_STREAM) again.push_back(k);         // glibc: draw-stream window ran out: larger one
                    else if (results[k].status == PRACH_ERR_TIMEOUT && was_packed) {       // the packed placement did not hold: plain cluster launch
                        again.push_back(k);
                        e->pack_off = true;
                    }
                    else if (results[k].status == PRACH_ERR_INTERNAL || results[k].status == PRACH_ERR_TIMEOUT) {
                        fallback.push_back(k);
                        nto += results[k].status == PRACH_ERR_TIMEOUT;
                    }
                }
                if (e->pack_off) {
                    std::fprintf(stderr, "[prach] %zu trial(s) of an XCD-packed cluster launch timed out waiting for a peer workgroup: rerun as a plain cluster launch\n", again.size());
                    e->last.spin_timeouts += (int32_t)again.size();
                }
                todo.swap(again);
            }
            e->pack_off = false;
            idx.swap(fallback);
            bool to_batch = G > 1 && e->opt_batch && !idx.empty();
            for (int k : idx) to_batch = to_batch && batch_eligible(e, cfgs[k]);
            if (!idx.empty()) note_fallback(e, to_batch ? "prach::batch_kernel (one workgroup per trial, event queue without a capacity)"
                                                        : "trial_kernel (one workgroup per trial, no per-subframe capacity)", idx.size(), nto, G);
            if (to_batch) {
                // Overflow without the cliff: most capacities a cluster trips over are PER WORKGROUP (512 event granules per mailbox, the
                // candidate list) or come with its LDS-resident layout; prach::batch_kernel has neither (one workgroup, the event queue
                // continues in global memory) and still runs all 16 wavefronts on the trial, so such trials go there first — the
                // one-workgroup, index-ordered trial_kernel (no per-subframe capacity at all, ~10x slower) only gets what is left.
                int rc = run_group(e, cfgs, idx.data(), (int)idx.size(), results, ue_logs, 0, 1, kernel_ms, upload_ms);
                if (rc != PRACH_OK) return rc;
                rc = rerun_full_calendars(e, cfgs, results, ue_logs, kernel_ms, upload_ms);
                if (rc != PRACH_OK) return rc;
                std::vector<int> still;
                for (int k : idx) if (results[k].status != PRACH_OK) still.push_back(k);
                idx.swap(still);
                if (!idx.empty()) std::fprintf(stderr, "[prach] %zu of them exceeded a capacity of the batch kernel's resolver too: rerun on trial_kernel\n", idx.size());
            }
            e->last.trial_kernel_reruns += (int32_t)idx.size();
            idx.insert(idx.end(), solo.begin(), solo.end());
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
    double seen = 0;
    for (int k = 0; k < n; k++) {
        upd += (uint64_t)cfgs[k].nUE * results[k].steps;
        if (results[k].status != PRACH_OK) worst = results[k].status;
        else if (cfgs[k].rng_mode == PRACH_RNG_GLIBC && cfgs[k].variant != PRACH_VARIANT_NOMA_C) seen = std::max(seen, (double)results[k].draws / (double)cfgs[k].nUE);
    }
    if (seen > 0) e->draws_per_ue_seen = seen; // (sizes the stream windows of the next call: stream_budget)
    e->last.kernel_ms = kernel_ms;
    e->last.noma_host_ues = e->noma_flagged;
    e->last.upload_ms = upload_ms;
    e->last.updates = upd;
    e->last.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return worst;
}

// C++ exceptions (std::bad_alloc from the staging vectors) never cross the C boundary
#define PRACH_GUARD(body)                                                                              \
    try { body } catch (const std::bad_alloc &) {                                                       \
        std::fprintf(stderr, "[prach] out of host memory\n");                                          \
        return PRACH_ERR_DEVICE;                                                                       \
    } catch (...) { return PRACH_ERR_INTERNAL; }

extern "C" {

int prach_engine_create(int device, prach_engine **out) {
    if (!out) return PRACH_ERR_ARG;
    *out = nullptr;
    PRACH_GUARD(return engine_create_impl(device, out);)
}

void prach_engine_destroy(prach_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    free_arena(e);
    if (e->pinned) (void)hipHostFree(e->pinned);
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
    if (std::strcmp(key, "resident") == 0) { if (value < 0) return PRACH_ERR_ARG; e->opt_resident = value; return PRACH_OK; }
    if (std::strcmp(key, "host_threads") == 0) { if (value < 0) return PRACH_ERR_ARG; e->opt_host_threads = value; return PRACH_OK; }
    if (std::strcmp(key, "lds_records") == 0) { e->opt_lds_records = value != 0; return PRACH_OK; }
    if (std::strcmp(key, "fast") == 0) { e->opt_fast = value != 0; return PRACH_OK; }
    if (std::strcmp(key, "batch") == 0) { e->opt_batch = value != 0; return PRACH_OK; }
    if (std::strcmp(key, "vmm_fail_after") == 0) { if (value < 0) return PRACH_ERR_ARG; e->opt_vmm_fail_after = value; return PRACH_OK; }
    if (std::strcmp(key, "plain_arena") == 0) { if (e->arena_cap) return PRACH_ERR_ARG; e->opt_plain_arena = value != 0; return PRACH_OK; } // (before the first call only)
    if (std::strcmp(key, "noma_ambiguity_test") == 0) { e->opt_noma_ambiguity_test = value != 0; return PRACH_OK; }
    if (std::strcmp(key, "noma_host_activation") == 0) { e->opt_noma_host_activation = value != 0; return PRACH_OK; }
    if (std::strcmp(key, "mem_budget_mb") == 0) { if (value <= 0) return PRACH_ERR_ARG; e->mem_budget = (size_t)value << 20; return PRACH_OK; } // (test hook: split launches)
    if (std::strcmp(key, "calendar_cap") == 0) { if (value < 0) return PRACH_ERR_ARG; e->opt_calendar_cap = value; return PRACH_OK; }
    if (std::strcmp(key, "batch_waves") == 0) { if (value != 0 && value != 8 && value != 16) return PRACH_ERR_ARG; e->opt_batch_waves = value; return PRACH_OK; }
    if (std::strcmp(key, "xcd_pack") == 0) { e->opt_xcd_pack = value != 0; return PRACH_OK; }
    return PRACH_ERR_ARG;
}

int prach_device_glibc_stream(prach_engine *e, uint32_t seed, uint64_t first, uint64_t n, int32_t *out) {
    if (!e || !out || n == 0) return PRACH_ERR_ARG;
    PRACH_GUARD(
        HIPCHK(hipSetDevice(e->device));
        const size_t nchunks = (n + STREAM_CHUNK - 1) / STREAM_CHUNK;
        const size_t need = align_up(4 * 31 * nchunks, 256) + 4 * (n + 2);
        { int rc = ensure_arena(e, need); if (rc != PRACH_OK) return rc; }
        std::vector<uint32_t> seeds(31 * nchunks);
        prach_internal_glibc_seeds(seed, first, nchunks, STREAM_CHUNK, seeds.data());
        HIPCHK(hipMemcpy(e->arena, seeds.data(), 4 * 31 * nchunks, hipMemcpyHostToDevice));
        int *dout = reinterpret_cast<int *>(e->arena + align_up(4 * 31 * nchunks, 256));
        HIPCHK(launch_glibc_stream(reinterpret_cast<const unsigned *>(e->arena), dout, n, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipMemcpy(out, dout, 4 * n, hipMemcpyDeviceToHost));
        return PRACH_OK;
    )
}

static int activation_table_device_impl(prach_engine *e, const prach_cfg *cfg, int32_t *preamble0, int32_t *sector, double *gain, double *lgain,
                                   uint32_t *ndraws, uint8_t *flagged) {
    if (!e || !cfg || !preamble0 || !sector || !gain || !lgain || !ndraws) return PRACH_ERR_ARG;
    { int v = prach_cfg_validate(cfg); if (v != PRACH_OK) return v; }
    if (cfg->variant != PRACH_VARIANT_NOMA_C || cfg->rng_mode != PRACH_RNG_PHILOX) return PRACH_ERR_ARG;
    HIPCHK(hipSetDevice(e->device));
    const size_t n = (size_t)cfg->nUE;
    size_t o = align_up(sizeof(TrialDev), 256);
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    const size_t oflags = take(4 * (2 + 2 * (size_t)NOMA_ACT_FLAG_CAP)), opre = take(4 * n), osec = take(4 * n), ogain = take(8 * n), olg = take(8 * n), ond = take(4 * n);
    { int rc = ensure_arena(e, o); if (rc != PRACH_OK) return rc; }
    char *const A = e->arena;
    TrialDev d;
    std::memset(&d, 0, sizeof(d));
    d.variant = cfg->variant; d.nUE = cfg->nUE; d.nP = cfg->nPreamble; d.seed_lo = (unsigned)cfg->seed; d.seed_hi = (unsigned)(cfg->seed >> 32);
    d.cell_radius = cfg->cellRadius;
    d.n_pre0 = reinterpret_cast<const int *>(A + opre); d.n_sector = reinterpret_cast<const int *>(A + osec);
    d.n_gain = reinterpret_cast<const double *>(A + ogain); d.n_lgain = reinterpret_cast<const double *>(A + olg);
    d.n_nd0 = reinterpret_cast<const unsigned *>(A + ond);
    HIPCHK(hipMemcpyAsync(A, &d, sizeof(d), hipMemcpyHostToDevice, e->stream)); // (on the engine's stream, like the kernel behind them)
    HIPCHK(hipMemsetAsync(A + oflags, 0, 8, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const int idx0 = 0;
    std::vector<ActTab> tabs(1, ActTab{A + opre, A + osec, A + ogain, A + olg, A + ond});
    std::vector<std::pair<int, int>> fl;
    e->noma_flagged = 0;
    { int rc = noma_device_activation(e, reinterpret_cast<const TrialDev *>(A), cfg, &idx0, 1, tabs, reinterpret_cast<unsigned *>(A + oflags), &fl); if (rc != PRACH_OK) return rc == NOMA_ACT_OVERFLOW_RC ? PRACH_ERR_INTERNAL : rc; }
    HIPCHK(hipMemcpy(preamble0, A + opre, 4 * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sector, A + osec, 4 * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(gain, A + ogain, 8 * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lgain, A + olg, 8 * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ndraws, A + ond, 4 * n, hipMemcpyDeviceToHost));
    if (flagged) {
        std::memset(flagged, 0, n);
        for (const auto &f : fl) flagged[f.second] = 1;
    }
    return PRACH_OK;
}

int prach_noma_activation_table_device(prach_engine *e, const prach_cfg *cfg, int32_t *preamble0, int32_t *sector, double *gain, double *lgain,
                                       uint32_t *ndraws, uint8_t *flagged) {
    PRACH_GUARD(return activation_table_device_impl(e, cfg, preamble0, sector, gain, lgain, ndraws, flagged);)
}

int prach_last_timing(const prach_engine *e, prach_timing *out) {
    if (!e || !out) return PRACH_ERR_ARG;
    *out = e->last;
    return PRACH_OK;
}

int prach_run_trials(prach_engine *e, const prach_cfg *cfgs, int n, prach_result *results, prach_ue_log *const *ue_logs) {
    if (!e || !cfgs || !results || n <= 0) return PRACH_ERR_ARG;
    PRACH_GUARD(return run_trials_impl(e, cfgs, n, results, ue_logs);)
}

} // extern "C"
