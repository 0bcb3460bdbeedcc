/* include/prach.h — C ABI of libprach_hip.so, the MI355X-native replacement for the reference's
 * per-subframe PRACH random-access simulation loop.
 *
 * The reference (yaki-toki/5G-NR-RandomAccess) has no library/FFI seam: each program inlines the
 * trial body in main() between calloc and free (RandomAccessSimulatorBeta.c:75-210,
 * RandomAccessWithNOMA.c:226-368, NOMA.c:650-714).  This header is the seam a maintainer would cut
 * there: ONE call per batch of (seed, nUE) trials replaces
 *     initialUE            Beta.c:220   WithNOMA:374
 *     activation loop      Beta.c:121-147 / activateUEs WithNOMA:383
 *     selectPreamble       Beta.c:229   WithNOMA:475
 *     preambleCollision    Beta.c:315   WithNOMA:607
 *     requestResourceAllocation Beta.c:371 WithNOMA:667
 *     timerIncrease        Beta.c:413   WithNOMA:712
 *     successUEs           Beta.c:421   WithNOMA:720
 *     end-of-trial sums    Beta.c:185-197 WithNOMA:337-351
 * and the writers reproduce saveSimulationLog / saveResult (Beta.c:432-514, WithNOMA:731-825).
 * Plain C types only; no torch / HIP types cross this boundary.  INTEGRATION.md shows the stub a
 * maintainer adds on the reference side.
 */
#ifndef PRACH_H
#define PRACH_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* which program's loop is reproduced (they differ: SURVEY.md §7.3) */
#define PRACH_VARIANT_BETA_C      0 /* RandomAccessSimulatorBeta.c */
#define PRACH_VARIANT_WITHNOMA_C  1 /* RandomAccessWithNOMA.c      */
#define PRACH_VARIANT_NOMA_C      2 /* NOMA.c (sector power-level grouping) */

/* prach_cfg.flags: variants the reference carries as commented-out code (never executed by the programs as committed) */
#define PRACH_FLAG_SECTOR_GRANTS   1 /* WITHNOMA_C: one UL-grant budget per 60-degree sector (sectorGrants[6], RandomAccessWithNOMA.c:260,
                                        271-273; the call of :312 and the grantCheck[sector] test of :626-637 un-commented); the sector
                                        comes from activateUEs' first draw (WithNOMA:393-410).  Runs on prach::batch_kernel (one workgroup
                                        per trial) in both RNG modes */
#define PRACH_FLAG_NOMA_NONSECTOR  2 /* NOMA_C: the cell-wide grouping preambleCollisionDetection (NOMA.c:325-447, call of :688
                                        un-commented) instead of preambleSectorCollisionDetection: one nGrantUL budget per access slot */

#define PRACH_RNG_GLIBC  0 /* the reference's own draw stream: srand(seed)/rand(), bit-exact vs the reference */
#define PRACH_RNG_PHILOX 1 /* Philox4x32-10, counter = (ue, draw#, nUE, variant), key = seed */

/* status codes (0 = ok).  The reference checks nothing and returns void everywhere; its only
 * error behaviour is the CLI's message + exit(-1) (WithNOMA:94-204), reproduced by prach_cli. */
#define PRACH_OK                 0
#define PRACH_ERR_ARG           -1 /* NULL pointer / non-positive nUE, nPreamble, backoff, accessTime ... */
#define PRACH_ERR_UNSUPPORTED   -2 /* nPreamble > 254, maxRarWindow > 255, maxMsg2TxCount > 255 */
#define PRACH_ERR_DEVICE        -3 /* HIP runtime error, no gfx950 device */
#define PRACH_ERR_STREAM        -4 /* glibc draw stream exhausted (engine retries internally) */
#define PRACH_ERR_INTERNAL      -5 /* device-side consistency check failed */
#define PRACH_ERR_IO            -6
#define PRACH_ERR_TIMEOUT       -7 /* a workgroup of a trial's cluster waited too long for a peer workgroup (the cluster was not
                                      wholly resident, e.g. another process holds CUs): the engine reruns such a trial on a
                                      kernel that waits for nobody and counts it in prach_timing; callers never see it */

typedef struct prach_cfg {
    int32_t variant;        /* PRACH_VARIANT_* */
    int32_t uniform;        /* 1: Uniform arrivals over 60 000 ms (Beta.c:92); 0: Beta(3,4) over 10 000 ms (Beta.c:103) */
    int32_t nUE;            /* Beta.c:75 */
    int32_t nPreamble;      /* Beta.c:47 */
    int32_t backoff;        /* backoffIndicator, Beta.c:48 */
    int32_t nGrantUL;       /* Beta.c:49; effective grants per 5 ms = value-1 (Beta.c:336-338) */
    int32_t maxRarWindow;   /* Beta.c:52 */
    int32_t maxMsg2TxCount; /* Beta.c:53 */
    int32_t accessTime;     /* Beta.c:54 */
    int32_t rng_mode;       /* PRACH_RNG_* */
    uint64_t seed;          /* srand(seed), Beta.c:69 */
    uint64_t stream_offset; /* glibc mode: rand() calls already consumed from this seed's stream
                               (the reference seeds once per seed and lets the stream run on across
                               the nUE sweep: Beta.c:66-71) */
    int32_t max_steps;      /* 0 = run to maxTime; >0 = stop after that many subframes */
    int32_t flags;          /* PRACH_FLAG_*: code paths the reference's author left commented out */
    float cellRadius, hBS, hUT; /* parsed by the CLI, never read by the simulation (WithNOMA:80-82);
                                   NOMA_C: cellRadius scales the UE drop (NOMA.c:56,168) */
} prach_cfg;

typedef struct prach_result {
    int32_t status;            /* PRACH_OK or error for this trial */
    int32_t time_exit;         /* `time` after the loop: printed "Total simulation time" (Beta.c:441) */
    int32_t maxTime;
    int32_t nSuccessUE;        /* Beta.c:178 */
    int32_t failedUEs;         /* Beta.c:185 */
    int32_t preambleTxCount;   /* Beta.c:194 */
    int32_t failCounts;        /* WithNOMA:348 */
    int32_t collisionPreambles, totalPreambleTxop; /* globals Beta.c:41-42 */
    int32_t activeCheck;       /* Beta.c:200 */
    int32_t nAccessUE;         /* Beta.c:95 */
    int32_t continueFaliedUEs, finalSuccessUEs; /* WithNOMA:84-85 */
    float totalDelay;          /* float running sum in index order, Beta.c:193 */
    int64_t sumTimer;          /* the same sum in exact integer arithmetic */
    uint64_t draws;            /* rand() calls consumed (glibc: add to stream_offset for the next trial) */
    uint64_t steps;            /* subframes executed; UE-subframe updates = nUE * steps */
} prach_result;

/* the 15 fields saveResult logs per UE (Beta.c:501-508) + failCount (WithNOMA:33) */
typedef struct prach_ue_log {
    int32_t idx, timer, active, txTime, firstTxTime, secondTxTime, nowBackoff, preamble, preambleChange,
        rarWindow, maxRarCounter, preambleTxCounter, msg2Flag, connectionRequest, msg4Flag, failCount;
} prach_ue_log;

typedef struct prach_timing {
    double kernel_ms;     /* HIP-event time of the simulation kernel launch(es) of the last call, on the engine's stream */
    double upload_ms;     /* host->device staging (schedules, parameters, glibc stream seeds; NOMA_C: the activation tables) */
    double total_ms;      /* wall time of the whole call */
    int32_t launches;     /* kernel launches in the last call */
    int32_t workgroups;   /* workgroups of the last launch */
    uint64_t updates;     /* sum over trials of nUE * steps */
    int32_t cluster_size;    /* workgroups per trial of the last launch (0: the one-workgroup fallback kernel) */
    int32_t resident_limit;  /* workgroups of a cluster launch that can be resident at once: one per CU (every cluster layout takes
                                more than half a CU's LDS; the runtime's occupancy query for the smallest one is the upper bound);
                                a cluster launch never exceeds it (its workgroups wait for each other) */
    int32_t fallback_trials; /* trials of the last call that a cluster launch could not finish and that were rerun (exactly) on
                                a kernel that waits for nobody: a per-subframe capacity exceeded, or a peer wait timed out.  Philox
                                trials go to prach::batch_kernel first (one workgroup per trial, event queue without a capacity);
                                trial_kernel_reruns (below) counts the ones that ended on the slow index-ordered trial_kernel */
    int32_t spin_timeouts;   /* ... of which: peer waits that timed out (PRACH_ERR_TIMEOUT) */
    int32_t rec_mode;        /* which kernel / record form the last cluster launch used: 0 prach::cluster_kernel, 16-byte records in global
                                memory; 1 the same with 8 + 4 byte records (one workgroup per trial: the glibc modes); 2 the same with
                                records resident in LDS (clusters, Philox); 3 prach::lcluster_kernel (the lean LDS-resident cluster
                                kernel: single Philox trials, the bench's N = 1 workload); 4 prach::batch_kernel (one workgroup per
                                trial, 4-byte pass words + 32-byte event records: the batched sweeps, Philox) */
    int32_t xcd_packed;   /* 1: the last cluster launch was XCD-packed (each cluster on the CUs of one XCD; prach_engine_set "xcd_pack") */
    uint64_t group_visits;   /* one workgroup per trial: 64-UE group visits of the pass, summed over the call */
    uint64_t event_ues;      /* ... and UEs that went through the full event body: the kernel's OWN memory work, for a roofline built from
                                the bytes it really moves (batch_kernel: a visit reads one 4-byte pass word per lane, an event reads and
                                writes one 32-byte record and one pass word) */
    int32_t trial_kernel_reruns; /* of fallback_trials: trials that were (also) rerun on the one-workgroup, index-ordered trial_kernel */
    int32_t noma_host_ues;       /* NOMA_C, Philox: UEs of the device-built activeUE table that the host recomputed with its libm (a value inside the
                                    device math library's error band of a rounding / comparison boundary: ~1e-6 of the UEs) */
} prach_timing;

typedef struct prach_engine prach_engine;

/* Engine lifetime: owns the HIP stream and every device buffer (nothing is allocated per subframe,
 * unlike preambleCollision's malloc per call, Beta.c:319).  device = HIP ordinal. */
int prach_engine_create(int device, prach_engine **out);
void prach_engine_destroy(prach_engine *);

/* Run n independent trials concurrently on the device (one workgroup cluster per trial).
 * ue_logs may be NULL; otherwise ue_logs[k] is NULL or a caller-owned array of cfgs[k].nUE entries. */
int prach_run_trials(prach_engine *, const prach_cfg *cfgs, int n, prach_result *results,
                     prach_ue_log *const *ue_logs);
int prach_last_timing(const prach_engine *, prach_timing *out);

/* engine tunables; none changes a result, all are covered by parity tests:
 *   "cluster"       workgroups cooperating on one trial (1..64; 0 = auto)
 *   "stream_factor" glibc mode: initial draws-per-UE budget of the rand() stream window (0 = auto; it grows on demand)
 *   "legacy"        1: run on the one-workgroup-per-trial kernel (the exact fallback of every capacity check)
 *   "dense"         1: cluster kernel without the compacted two-phase pass
 *   "wide_records"  1: 16-byte hot records also with one workgroup per trial
 *   "pipeline"      0: a cluster does not run phase A of the next subframe during the exchange of the current one
 *   "resident"      test hook: treat only this many workgroups as co-resident (0 = ask the runtime's occupancy query)
 *   "host_threads"  NOMA_C: host threads that build the activation tables (0 = all cores)
 *   "lds_records"   0: clusters keep their UE records in global memory instead of LDS
 *   "fast"          0: LDS-resident clusters run on the general cluster kernel instead of prach::lcluster_kernel
 *   "batch"         0: one-workgroup-per-trial Philox launches run on the general cluster kernel instead of prach::batch_kernel
 *   "xcd_pack"      0: clusters are not launched XCD-packed
 *   "batch_waves"   prach::batch_kernel's workgroup shape: 8 (512 threads, two trials per CU), 16 (1024 threads), 0 = chosen per launch
 *   "plain_arena"   1: the device arena is one hipMalloc allocation, re-allocated when it grows (before the first call only; diagnostic)
 *   "mem_budget_mb" arena megabytes one launch may take (default: three quarters of the device's memory): a call that needs more runs as several launches
 *   "calendar_cap", "vmm_fail_after", "noma_ambiguity_test", "noma_host_activation"   test hooks (prach_engine.hip) */
int prach_engine_set(prach_engine *, const char *key, int64_t value);

/* Host-side pieces of the same seam (no device needed) */
void prach_cfg_defaults(prach_cfg *cfg, int variant); /* Beta.c:47-57 / WithNOMA:70-88 */
int prach_cfg_validate(const prach_cfg *cfg);
int prach_max_time(const prach_cfg *cfg);
/* What this trial costs inside a batched launch, in kernel microseconds on an MI355X (a measured table, prach_host.c): the weight the multi-GPU dealing of the
   --times x sweep grid balances — the reference runs that grid serially (RandomAccessWithNOMA.c:216-221), so there is nothing there to replace. */
double prach_trial_cost(const prach_cfg *cfg);
/* out[s] = activeCheck after the arrival update of access slot s (Beta.c:121-134); returns #slots */
int prach_arrival_schedule(const prach_cfg *cfg, int32_t *out, int cap, int32_t *nAccessUE);
/* k-th .. k+n-th values of srand(seed)/rand() */
void prach_glibc_stream(uint32_t seed, uint64_t first, uint64_t n, int32_t *out);
/* the same stream generated ON THE DEVICE (what glibc-mode trials consume): the host jumps ahead with 31x31 matrix
 * powers of the lagged-Fibonacci recurrence, one wavefront per 63 488-value chunk rolls it forward */
int prach_device_glibc_stream(prach_engine *, uint32_t seed, uint64_t first, uint64_t n, int32_t *out);
const char *prach_strerror(int status);

/* NOMA.c variant (PRACH_VARIANT_NOMA_C): per-UE attributes fixed at activation (activeUE, NOMA.c:131-192):
 * first preamble, sector, Rayleigh channel gain and its natural log (the pairing test of NOMA.c:276 uses
 * 10*log(high)-10*log(low)), and the number of draws the activation consumed.  This is the HOST form, with the libm the
 * reference links (cos, sin, log, pow).  The engine builds the table on the DEVICE (prach_noma_activation_table_device below is that
 * kernel on its own) and calls the host form only for the UEs the kernel flags: wherever a last-bits difference between the device's
 * math library and the host's could change a float rounding, the rejection test or the draw count.  Engine option
 * "noma_host_activation" = 1 builds the whole table with the functions here instead (the round-1/2 behaviour).
 * This table form is the Philox mode's (draw k of UE i is independent of every other UE).  In glibc mode the rejection loops make
 * every stream position data dependent: there the whole trial is one launch with activeUE inside it (same error bands); a trial that hits
 * a band is run again slot by slot, its arrivals activated one by one with prach_noma_activation_stream below. */
int prach_noma_activation_table(const prach_cfg *cfg, int32_t *preamble0, int32_t *sector, double *gain, double *lgain,
                                uint32_t *ndraws);
/* the same for the UEs [lo, hi) only (outputs indexed from lo): ranges are independent, the engine builds them on all host cores */
int prach_noma_activation_range(const prach_cfg *cfg, int lo, int hi, int32_t *preamble0, int32_t *sector, double *gain, double *lgain,
                                uint32_t *ndraws);
/* The device-built table as the engine uses it (Philox mode), copied back: preamble0 / sector / ndraws of every UE equal the host form's;
 * gain / lgain of an unflagged UE are within a few ulp of it (never read except through comparisons that carry an error band), those of a
 * flagged UE (flagged[i] != 0; nullable) are the host form's, recomputed by this call.  For tests and diagnosis. */
int prach_noma_activation_table_device(prach_engine *, const prach_cfg *cfg, int32_t *preamble0, int32_t *sector, double *gain, double *lgain,
                                       uint32_t *ndraws, uint8_t *flagged);
/* activeUE for ONE UE in the reference's own rand() stream (glibc mode of the NOMA_C variant): draws stream[*pos...] in the reference's
 * order, *pos advances; PRACH_ERR_STREAM when the window of `avail` values is exhausted */
int prach_noma_activation_stream(const prach_cfg *cfg, const int32_t *stream, uint64_t *pos, uint64_t avail, int32_t *preamble0, int32_t *sector,
                                 double *gain, double *lgain);
size_t prach_format_noma_line(const prach_cfg *, const prach_result *, char *buf, size_t cap); /* NOMA.c:606-632 */

/* Text surfaces, byte-compatible with the reference (latency values excepted) */
size_t prach_format_logs(const prach_ue_log *ue, int nUE, char *buf, size_t cap);           /* Beta.c:501 */
size_t prach_format_results(const prach_cfg *, const prach_result *, double latency_s, char *buf, size_t cap); /* Beta.c:460-482 / WithNOMA:762-793 */
size_t prach_format_stdout(const prach_cfg *, const prach_result *, double latency_s, char *buf, size_t cap);  /* Beta.c:200-206,440-446 / WithNOMA:354-361,741-748 */
int prach_result_file_name(const prach_cfg *, int is_log, char *buf, size_t cap);            /* Beta.c:452-456,490-494 / WithNOMA:754-758,801-805 */
int prach_write_trial_files(const prach_cfg *, const prach_result *, const prach_ue_log *ue, double latency_s,
                            const char *root_dir);

/* results.csv (AveragePerformance.py:7-24): per nUE point the six lines of every seed's Beta.c Results.txt
 * are summed in seed order (Python floats = doubles), divided by the number of seeds, rounded with
 * np.around(x, 3) and written by csv.writer (shortest float repr, CRLF line ends, no header).
 * prach_results_csv_accumulate adds one Results.txt text to acc[6]; prach_results_csv_row formats one row. */
int prach_results_csv_accumulate(double acc[6], const char *results_txt);
size_t prach_results_csv_row(const double acc[6], int nseeds, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
