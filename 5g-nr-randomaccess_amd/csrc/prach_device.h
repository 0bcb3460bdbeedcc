// prach_device.h — device-side data layout shared by the kernels (prach_kernels.hip) and the
// engine (prach_engine.hip).  gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/prach.h"

namespace prach {

// ---- per-UE hot record: ONE 16-byte word (a single global_load_dwordx4 per UE per subframe) -------
//   x  txTime                       (Beta.c:15)   [PEND_RESET: the drawn backoff `tmp`, until applied]
//   y  tb  timer base: the subframe at which `timer` was last zeroed (timer = executed - tb), or the
//          final timer value once the UE has succeeded (Beta.c:13,378)
//   z  bo  nowBackoff, stored as its expiry subframe when positive (value = max(bo - t, 0)), as the
//          literal value when <= 0 (Beta.c:25)     [PEND_RESET: the UE's previous preamble]
//   w  pk  packed small fields, see below
// timer and nowBackoff are therefore never rewritten while a UE merely waits (timerIncrease,
// Beta.c:413-419, costs no memory traffic).
constexpr unsigned PK_ACT_SHIFT = 0, PK_CONN_SHIFT = 2, PK_PRE_SHIFT = 4, PK_RAR_SHIFT = 12, PK_MRC_SHIFT = 20,
                   PK_PEND_SHIFT = 28;
constexpr int ACT_IDLE = 0; // active == -1 (not yet arrived)
constexpr int ACT_DONE = 1; // active ==  0 (msg4Flag == 1)
constexpr int ACT_M1 = 2;   // active ==  1 (Msg1/Msg2 phase)
constexpr int ACT_M3 = 3;   // active ==  2 (Msg3/Msg4 phase)

// what the deferred "apply" of the previous subframe still owes this UE
constexpr int PEND_NONE = 0, PEND_STAY = 1, PEND_CALLER = 2, PEND_RESET = 3, PEND_PASSIVE = 4, PEND_RJOIN = 5;
// bit 31 of the packed word: the resolver granted this (singleton-calling) UE an UL grant — set with ONE
// fire-and-forget atomicOr by the UE's owner workgroup, consumed by the next pass's apply
constexpr unsigned PK_GRANT_BIT = 0x80000000u;
// ordered "special" events handed to the resolver
constexpr int EV_CALLER = 1, EV_RESETCAND = 2, EV_PASSIVE = 3, EV_RJOIN = 4;

struct alignas(16) Event {
    int idx;  // UE index
    int info; // type | bucket p << 8 | old bucket q << 16
    int le;   // #pre-members of bucket p with index <= idx
    int pad;
};

struct DevResult {
    int status, time_exit, nSuccess, collisionPreambles, totalPreambleTxop, activeCheck, continueFailed,
        finalSuccess, ptcSum, fcSum;
    unsigned long long draws, steps;
    long long sumTimer;
    unsigned long long dbg[4];
    unsigned long long stamps6[8]; // diagnostic build: cycles per phase (pass, publish, barrier, gather, resolve, grants)
    unsigned long long fstamps[24]; // diagnostic build: finer split (prach_cluster.hip FSTAMP)
    unsigned long long visits, events; // one workgroup per trial: 64-UE group visits of the pass, UEs through the event body (the kernel's OWN memory work)
    int hard_error;                    // a workgroup of the trial left with PRACH_ERR_INTERNAL (a per-subframe capacity): wins over a peer's PRACH_ERR_TIMEOUT in `status`
    int pad_;
};

struct TrialDev {
    int variant, uniform, nUE, nP, backoff, nGrantUL, maxRarWindow, maxMsg2, aT, rng_mode, maxTime, stop;
    unsigned seed_lo, seed_hi;
    unsigned long long stream_len;
    int4 *rec;
    int *ptc, *ftt, *stt, *fcnt; // cold per-UE fields: preambleTxCounter, firstTxTime, secondTxTime, failCount
    unsigned *nd;                // philox: per-UE draw index
    Event *evbuf, *evbuf2;       // event scratch (per-wave segments / compacted)
    int *sidx;                   // singleton-caller scratch
    const int *sched;            // arrival schedule (activeCheck per access slot)
    const int *stream;           // glibc mode: draw stream window
    prach_ue_log *logs;          // nullable
    int *timers;                 // per UE: final timer if succeeded, else INT_MIN
    DevResult *out;
    // cluster kernel (prach_cluster.hip) only
    int evw, mbstride;           // events per mailbox, mailbox stride (ints)
    int binshift;                // grant selection: UE index >> binshift < 1024
    int *mbox;                   // [2][G][mbstride] write-through mailboxes
    int2 *cand;                  // early-leaver candidate scratch, nUE + 64*G entries
    int dense_pass;              // 1: every group through the full per-UE body (diagnostic option); 0: compacted pass
    int pipeline;                // 1: clusters run phase A of the next subframe during the exchange of the current one
    // NOMA.c variant (prach_noma.hip) only: the host-built activation table
    const int *n_pre0, *n_sector;
    const double *n_gain, *n_lgain;
    const unsigned *n_nd0;
    float cell_radius;           // NOMA.c:56,168 (the UE drop of activeUE)
    int n_devact;                // 1: the table was built on the device (noma_activation_kernel): the resolver checks its gain comparisons against the error band
    int flags;                   // PRACH_FLAG_* (include/prach.h)
    int *sector;                 // PRACH_FLAG_SECTOR_GRANTS: per UE, the sector drawn by activateUEs (WithNOMA:393-410); else null
    // batch kernel (prach_batch.hip) only
    int4 *rec32;                 // [nUE][2] the 32-byte event record
    int4 *chunks;                // [nchunks][128] the pool of 2 KB chunks: the records of 64 UEs whose next event falls into one subframe
    int *ctab;                   // [calmask + 1][tcap] chunk table: the chunks of every future subframe's event list
    int *cpool;                  // [2][nchunks] shared pool of free chunk ids
    int nchunks, tcap;
    int *jcal;                   // [calmask + 1][calcap] join lists: the UEs whose contention window opens in a subframe
    int calcap, calmask;         // entries per join list; calendar slots - 1 (slot = subframe & calmask)
    int *qov;                    // [nUE] early-leaver candidates of a subframe beyond their LDS part
    int2 *evov;                  // [2 nUE] the resolver's event list of a subframe beyond its LDS part
    unsigned long long *diag;    // diagnostic build (PRACH_STAMPS) only: [CLUSTER_MAX_G][32] per-workgroup phase stamps of a cluster trial; else null
};

constexpr int WG_THREADS = 1024;
constexpr int NW = WG_THREADS / 64; // waves per workgroup
constexpr int EVCAP = 1024;         // events staged in LDS (more: resolved from global scratch)
constexpr int SCAP = 1024;          // singleton callers staged in LDS
constexpr int RCCAP = 256;          // reset-cycle re-join candidates per subframe

size_t trial_kernel_lds_bytes(int nP);
hipError_t launch_trial_kernel(const TrialDev *params, int ntrials, int rng_mode, int maxP, hipStream_t stream);
size_t cluster_kernel_lds_bytes(int nP, bool glibc, int lslots);
// rec_mode: where / how a trial's hot records are kept (prach_cluster.hip): 0 global 16 B, 1 global 8 + 4 B (one workgroup per
// trial, the reference's rand() stream), 2 LDS-resident (clusters, Philox; lslots = owned UE slots per workgroup, the launch's maximum)
constexpr int CLUSTER_REC_G16 = 0, CLUSTER_REC_H8 = 1, CLUSTER_REC_L16 = 2;
constexpr int CLUSTER_EVW = 512;   // special-event granules per cluster mailbox and subframe
constexpr int CLUSTER_REC_LFAST = 3; // prach_lcluster.hip: the lean LDS-resident kernel (Philox clusters, nPreamble <= 64)
size_t lcluster_kernel_lds_bytes(int lslots, bool glibc = false, int groups = 0);
int lcluster_group_capacity(int groups);
int lcluster_max_groups_glibc();
int lcluster_max_preambles();
hipError_t launch_lcluster_kernel(const TrialDev *params, int ntrials, int G, int lslots, int xpack, bool glibc, int groups, hipStream_t stream);
constexpr int CLUSTER_LQCAP = 4096; // LDS-resident clusters: owned UE slots per workgroup at most (= the event queue)
constexpr size_t CLUSTER_LDS_LIMIT = 160 * 1024; // LDS per CU (MI355X_MICROARCH.md): one LDS-resident cluster workgroup per CU
hipError_t launch_cluster_kernel(const TrialDev *params, int ntrials, int G, int maxP, int rng_mode, int rec_mode, int lslots, int xpack, hipStream_t stream);
int cluster_kernel_blocks_per_cu(int maxP, int rng_mode, int rec_mode, int lslots); // occupancy query for the kernel and its dynamic LDS size
constexpr int CLUSTER_REC_BATCH = 4; // prach_batch.hip: one workgroup per trial, 4-byte pass words + 32-byte event records (Philox)
size_t batch_kernel_lds_bytes(int waves, bool glibc = false);
int batch_max_preambles();
int batch_max_rar_window();
int batch_max_subframes();
int batch_max_groups(bool glibc = false);
int batch_max_rar_window_two_per_cu();
int batch_chunk_bytes();
int batch_calendar_slots(int backoff, int accessTime, int maxRarWindow);
int batch_max_calendar_slots();
hipError_t launch_batch_kernel(const TrialDev *params, int ntrials, int waves, bool glibc, hipStream_t stream);
constexpr int STREAM_CHUNK = 31 * 2048; // rand() outputs generated per wavefront (prach_stream.hip)
hipError_t launch_glibc_stream(const unsigned *seeds, int *out, unsigned long long n, hipStream_t stream);
struct StreamJob { const unsigned *seeds; int *out; unsigned long long n; }; // one trial's window: a 31-word seed window per chunk, n values out
hipError_t launch_glibc_stream_jobs(const StreamJob *jobs, int njobs, unsigned long long max_n, hipStream_t stream); // (jobs: device memory)
extern "C" void prach_internal_glibc_seeds(uint32_t seed, uint64_t first, uint64_t nchunks, uint64_t chunk, uint32_t *out);
constexpr int CLUSTER_GLIBC_MAX_UE = 4096 * 64; // glibc mode on the cluster kernel: per-group draw counts live in LDS
constexpr int CLUSTER_MAX_G = 64;
size_t noma_kernel_lds_bytes(int nP);
hipError_t launch_noma_kernel(const TrialDev *params, int ntrials, int G, int maxP, int xpack, hipStream_t stream);
int noma_kernel_blocks_per_cu(int maxP);
// activeUE (NOMA.c:131-192) for every UE of every trial of a Philox launch, on the device.  flags[0]: number of UEs whose result may differ
// from the host libm's (a value within the error band of a rounding or comparison boundary); flags[2 + 2 q], flags[3 + 2 q]: trial, UE index
// of the q-th (q < cap).  The engine recomputes those with prach_noma_activation_range before the simulation kernel starts.
constexpr int NOMA_ACT_FLAG_CAP = 8192;
constexpr int NOMA_AMBIGUOUS = 77; // DevResult::hard_error: a gain comparison of the resolver fell inside the error band (the trial is rerun with the host-built table)
hipError_t launch_noma_activation(const TrialDev *params, int ntrials, int maxUE, unsigned *flags, hipStream_t stream);
// NOMA_C in the reference's own rand() stream (prach_noma_glibc.hip): one trial, host-activated arrivals + one device step per access slot
constexpr int NOMA_GLIBC_AMBIGUOUS_RC = -1077; // run_noma_glibc_batch: a value inside the device libm's error band — run that trial on the host-activated path
int run_noma_glibc_trial(hipStream_t stream, const prach_cfg &c, const int32_t *hstream, unsigned long long len, prach_result *res, prach_ue_log *logs,
                         double *kernel_ms);
int run_noma_glibc_batch(hipStream_t stream, const prach_cfg *const *cfgs, int n, const unsigned long long *lens, prach_result *const *res, prach_ue_log *const *logs,
                         double *kernel_ms, int *rcs);

} // namespace prach
