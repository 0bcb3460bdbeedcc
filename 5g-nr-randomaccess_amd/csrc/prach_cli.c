/* prach_cli.c — `prach_sim`: host-C driver with the reference's command-line surface
 * (RandomAccessWithNOMA.c:90-206, README.md:43-87) on top of the C ABI (include/prach.h).
 *
 * Same (flag, value) pair parsing, same messages and exit(-1) on bad values, same seed loop x
 * nUE sweep (WithNOMA:216-221), same stdout block and result files; the per-subframe loop itself
 * runs on the MI355X through prach_run_trials().  Both the spellings the code accepts (-rc, -mrc,
 * -bs, -ut) and the ones README.md documents (-r, -m, -u) are taken; `-d 2` selects Beta as the
 * README says (the reference's validation rejects it).
 *
 * `--program noma` runs NOMA.c's loop instead (NOMA.c:644-717: 10 seeds by default, one result line per
 * (seed, nUE) on stdout and appended to TestResults/Sector_{nUE}_Result.txt, "Done" per seed); that
 * variant draws from Philox by default; `--rng glibc` runs it in the reference's own rand() stream (prach_noma_glibc.hip: one launch per
 * sweep point, the seeds side by side), chained over the sweep of a seed like NOMA.c:644-647.
 *
 * Extensions (not in the reference): --program beta|withnoma|noma, --rng glibc|philox, --nue N,
 * --sweep LO:HI:STEP, --out DIR, --logs 0|1, --device N, --csv FILE (--program beta: the results.csv of
 * AveragePerformance.py over the --times seeds of every sweep point, written by prach_results_csv_*),
 * --sector-grants 1 (--program withnoma) / --nonsector 1 (--program noma): the two code paths the reference carries
 * commented out (WithNOMA:312,626-637 / NOMA.c:325-447,688);
 * --devices LIST: the same with explicit HIP ordinals (an ordinal may repeat);
 * --gpus N: the --times x sweep grid sharded over N devices of the node by host C — one forked child per device,
 * forked BEFORE any HIP call, trials dealt by descending cost (Philox: any trial anywhere; glibc: whole seeds, because
 * the sweep of a seed is chained through its rand() stream), results merged by the parent through shared memory; the
 * reference runs the grid serially (RandomAccessWithNOMA.c:216-221).
 */
#define _GNU_SOURCE
#include "../../include/prach.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

static void usage_and_exit(void) { /* text of WithNOMA:160-202 */
    printf("--times         -t : Simulation times (int)\n");
    printf("                     Simulation count must be greater than zero.\n");
    printf("                     Default 1\n\n");
    printf("--distribution  -d : Traffic model (1 or 2)\n");
    printf("                     1: traffic model 1 (Uniform distribution)\n");
    printf("                     2: traffic model 2 (Beta distribution)\n\n");
    printf("--preambles     -p : Number of preambles (int)\n");
    printf("                     Number of preamble must be greater than zero.\n");
    printf("                     Default 54\n\n");
    printf("--backoff       -b : Backoff indicator (int)\n");
    printf("                     Backoff indicator must be greater than zero.\n");
    printf("                     Default 20\n\n");
    printf("--grant         -g : The number of Up Link Grant per RAR (int)\n");
    printf("                     The number of Up Link Grant per RAR must be greater than zero.\n");
    printf("                     Default 12\n\n");
    printf("--rarCount      -r : RAR window size (int)\n");
    printf("                     The maximum RAR window size must be greater than zero.\n");
    printf("                     Default 5\n\n");
    printf("--maxRar        -m : Maximum retransmission (int)\n");
    printf("                     Maximum retransmissions must be greater than zero.\n");
    printf("                     Default 10\n\n");
    printf("--subframe      -s : Subframe units (int)\n");
    printf("                     The size of the subframe must be at least 5. (float)\n");
    printf("                     Default 5\n\n");
    printf("--cell          -c : Cell radius Size\n");
    printf("                     The radius of the cell is entered in diameter units and must be greater than 400m.\n");
    printf("                     Default 400.0\n\n");
    printf("--hbs           -b : Height of BS from ground (float)\n");
    printf("                     The height of the BS must be between 10m and 20m.\n");
    printf("                     Default 10.0\n\n");
    printf("--hut           -u : Height of UE from ground (float)\n");
    printf("                     The height of the UE must be between 1.5m and 22.5m.\n");
    printf("                     Default 1.8\n\n");
    exit(-1);
}

static int is(const char *a, const char *l, const char *s1, const char *s2) {
    return strcmp(a, l) == 0 || (s1 && strcmp(a, s1) == 0) || (s2 && strcmp(a, s2) == 0);
}
static void die(const char *msg) { printf("%s", msg); exit(-1); }


static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* One worker = one device: runs the trials idx[0..m) of the grid on `device` and leaves every prach_result in the (shared)
 * array `res`; per-trial files are written by the worker itself (independent files).  Philox trials go out in calls of up to
 * 1024 trials (one workgroup or cluster per trial); in glibc mode `idx` holds whole seeds in grid order and the sweep of a
 * seed is chained through cfg.stream_offset like the reference's single srand() per seed (WithNOMA:219-221): one call per
 * sweep point with all the worker's seeds in it.  Returns 0 or an exit code. */
static int run_worker(int device, const prach_cfg *cfgs, const int *idx, int m, prach_result *res, double *lat_out, int want_logs,
                      const char *outdir, int glibc, int npts) {
    prach_engine *eng = NULL;
    int rc = prach_engine_create(device, &eng);
    if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: device %d: %s\n", device, prach_strerror(rc)); return 2; }
    const double t0 = now_s();
    const int CH = glibc ? m : 1024;
    prach_cfg *c = (prach_cfg *)malloc(sizeof(prach_cfg) * (size_t)(m > 0 ? m : 1));
    prach_result *r = (prach_result *)malloc(sizeof(prach_result) * (size_t)(m > 0 ? m : 1));
    prach_ue_log **logs = want_logs ? (prach_ue_log **)calloc((size_t)(m > 0 ? m : 1), sizeof(prach_ue_log *)) : NULL;
    if (!c || !r || (want_logs && !logs)) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
    if (!glibc) {
        for (int a = 0; a < m; a += CH) {
            const int n = m - a < CH ? m - a : CH;
            for (int k = 0; k < n; k++) {
                c[k] = cfgs[idx[a + k]];
                if (want_logs) {
                    logs[k] = (prach_ue_log *)malloc(sizeof(prach_ue_log) * (size_t)c[k].nUE);
                    if (!logs[k]) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
                }
            }
            rc = prach_run_trials(eng, c, n, r, logs);
            if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }
            const double lat = now_s() - t0;
            for (int k = 0; k < n; k++) {
                res[idx[a + k]] = r[k];
                lat_out[idx[a + k]] = lat;
                if (cfgs[idx[a + k]].variant != PRACH_VARIANT_NOMA_C) {
                    rc = prach_write_trial_files(&c[k], &r[k], want_logs ? logs[k] : NULL, lat, outdir);
                    if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }
                }
                if (want_logs) { free(logs[k]); logs[k] = NULL; }
            }
        }
    } else { /* idx = seeds' trials in grid order: idx[s * npts + k]; chained per seed */
        const int nseeds = m / npts;
        uint64_t *offset = (uint64_t *)calloc((size_t)(nseeds > 0 ? nseeds : 1), sizeof(uint64_t));
        if (!offset) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
        for (int k = 0; k < npts; k++) {
            for (int s_ = 0; s_ < nseeds; s_++) {
                c[s_] = cfgs[idx[s_ * npts + k]];
                c[s_].stream_offset = offset[s_];
                if (want_logs) {
                    logs[s_] = (prach_ue_log *)malloc(sizeof(prach_ue_log) * (size_t)c[s_].nUE);
                    if (!logs[s_]) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
                }
            }
            rc = prach_run_trials(eng, c, nseeds, r, logs);
            if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }
            const double lat = now_s() - t0;
            for (int s_ = 0; s_ < nseeds; s_++) {
                offset[s_] += r[s_].draws;
                res[idx[s_ * npts + k]] = r[s_];
                lat_out[idx[s_ * npts + k]] = lat;
                if (c[s_].variant != PRACH_VARIANT_NOMA_C) { /* (NOMA.c's lines are printed and appended by the parent) */
                    rc = prach_write_trial_files(&c[s_], &r[s_], want_logs ? logs[s_] : NULL, lat, outdir);
                    if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }
                }
                if (want_logs) { free(logs[s_]); logs[s_] = NULL; }
            }
        }
        free(offset);
    }
    free(c); free(r); free(logs);
    prach_engine_destroy(eng);
    return 0;
}

int main(int argc, char *argv[]) {
    int randomMax = 1, variant = PRACH_VARIANT_WITHNOMA_C, rng = PRACH_RNG_GLIBC, device = 0, want_logs = 1, gpus = 1, rng_given = 0;
    int sweep_lo = 10000, sweep_hi = 100000, sweep_step = 10000; /* WithNOMA:221 */
    const char *outdir = ".", *csv_path = NULL, *devlist = NULL;
    int devs[64];
    /* --program must be known before the defaults are laid down */
    for (int i = 1; i + 1 < argc; i += 2)
        if (strcmp(argv[i], "--program") == 0)
            variant = strcmp(argv[i + 1], "beta") == 0 ? PRACH_VARIANT_BETA_C : (strcmp(argv[i + 1], "noma") == 0 ? PRACH_VARIANT_NOMA_C : PRACH_VARIANT_WITHNOMA_C);
    prach_cfg base;
    prach_cfg_defaults(&base, variant);
    if (variant == PRACH_VARIANT_NOMA_C) { randomMax = 10; rng = PRACH_RNG_PHILOX; want_logs = 0; } /* NOMA.c:644 */

    for (int i = 1; i < argc; i += 2) {
        const char *a = argv[i];
        if (i + 1 >= argc) usage_and_exit(); /* the reference dereferences NULL here; we print the usage */
        const char *v = argv[i + 1];
        if (is(a, "--times", "-t", NULL)) {
            if (atoi(v) < 1) die("Simulation count must be greater than zero.");
            randomMax = atoi(v);
        } else if (is(a, "--distribution", "-d", NULL)) {
            if (atoi(v) != 1 && atoi(v) != 0 && atoi(v) != 2) die("Traffic model just choose 1 or 2");
            base.uniform = atoi(v) == 1;
        } else if (is(a, "--preambles", "-p", NULL)) {
            if (atoi(v) < 1) die("Number of preamble must be greater than zero.");
            base.nPreamble = atoi(v);
        } else if (is(a, "--backoff", "-b", NULL)) {
            if (atoi(v) < 1) die("Backoff indicator must be greater than zero.");
            base.backoff = atoi(v);
        } else if (is(a, "--grant", "-g", NULL)) {
            if (atoi(v) < 1) die("The number of Up Link Grant per RAR must be greater than zero.");
            base.nGrantUL = atoi(v);
        } else if (is(a, "--rarCount", "-rc", "-r")) {
            if (atoi(v) < 1) die("The maximum RAR window size must be greater than zero.");
            base.maxRarWindow = atoi(v) + 1; /* WithNOMA:128 */
        } else if (is(a, "--maxRar", "-mrc", "-m")) {
            if (atoi(v) < 1) die("Maximum retransmissions must be greater than zero.");
            base.maxMsg2TxCount = atoi(v) - 1; /* WithNOMA:134 */
        } else if (is(a, "--subframe", "-s", NULL)) {
            if (atoi(v) < 5) die("The size of the subframe must be at least 5.");
            base.accessTime = atoi(v);
        } else if (is(a, "--cell", "-c", NULL)) {
            if (atof(v) < 400.0) die("The radius of the cell is entered in diameter units and must be greater than 400m.");
            base.cellRadius = (float)atof(v);
        } else if (is(a, "--hbs", "-bs", NULL)) {
            if (atof(v) < 10.0 || atof(v) > 20.0) die("The height of the BS must be between 10m and 20m.");
            base.hBS = (float)atof(v);
        } else if (is(a, "--hut", "-ut", "-u")) {
            if (atof(v) < 1.5 || atof(v) > 22.5) die("The height of the UE must be between 1.5m and 22.5m.");
            base.hUT = (float)atof(v);
        } else if (strcmp(a, "--program") == 0) {
            /* handled above */
        } else if (strcmp(a, "--rng") == 0) {
            rng = strcmp(v, "philox") == 0 ? PRACH_RNG_PHILOX : PRACH_RNG_GLIBC;
            rng_given = 1;
        } else if (strcmp(a, "--nue") == 0) {
            if (atoi(v) < 1) die("Number of UEs must be greater than zero.");
            sweep_lo = sweep_hi = atoi(v); sweep_step = 1;
        } else if (strcmp(a, "--sweep") == 0) {
            if (sscanf(v, "%d:%d:%d", &sweep_lo, &sweep_hi, &sweep_step) != 3 || sweep_lo < 1 || sweep_step < 1 || sweep_hi < sweep_lo)
                die("--sweep LO:HI:STEP");
        } else if (strcmp(a, "--out") == 0) {
            outdir = v;
        } else if (strcmp(a, "--logs") == 0) {
            want_logs = atoi(v) != 0;
        } else if (strcmp(a, "--device") == 0) {
            device = atoi(v);
        } else if (strcmp(a, "--sector-grants") == 0) { /* the author's commented-out per-sector grant path (WithNOMA:312,626-637) */
            if (atoi(v)) base.flags |= PRACH_FLAG_SECTOR_GRANTS;
        } else if (strcmp(a, "--nonsector") == 0) { /* the author's commented-out cell-wide NOMA grouping (NOMA.c:688) */
            if (atoi(v)) base.flags |= PRACH_FLAG_NOMA_NONSECTOR;
        } else if (strcmp(a, "--gpus") == 0) {
            if (atoi(v) < 1 || atoi(v) > 64) die("--gpus N: 1..64 devices of this node");
            gpus = atoi(v);
        } else if (strcmp(a, "--devices") == 0) { /* explicit HIP ordinals, e.g. 0,2,3 (an ordinal may repeat: workers then share that device) */
            devlist = v;
        } else if (strcmp(a, "--csv") == 0) {
            csv_path = v;
        } else {
            usage_and_exit();
        }
    }
    base.rng_mode = rng;
    if (csv_path && variant != PRACH_VARIANT_BETA_C) die("--csv needs --program beta (AveragePerformance.py reads its six-number Results.txt)");
    if (variant == PRACH_VARIANT_NOMA_C) { want_logs = 0; if (!rng_given) rng = PRACH_RNG_PHILOX; base.rng_mode = rng; } /* --rng glibc: NOMA.c's own rand() stream */

    /* the grid: trial (seed s, sweep point k) = cfgs[s * npts + k], the reference's loop order (WithNOMA:216-221 / NOMA.c:644-647) */
    const int npts = (sweep_hi - sweep_lo) / sweep_step + 1;
    const int ntr = randomMax * npts;
    const int glibc = rng == PRACH_RNG_GLIBC;
    prach_cfg *cfgs = (prach_cfg *)malloc(sizeof(prach_cfg) * (size_t)ntr);
    /* results and per-trial latencies live in shared memory: the workers (children) fill them, the parent prints and merges */
    prach_result *res = (prach_result *)mmap(NULL, sizeof(prach_result) * (size_t)ntr, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    double *lat = (double *)mmap(NULL, sizeof(double) * (size_t)ntr, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (!cfgs || res == MAP_FAILED || lat == MAP_FAILED) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
    for (int s_ = 0; s_ < randomMax; s_++)
        for (int k = 0; k < npts; k++) {
            prach_cfg *c = &cfgs[s_ * npts + k];
            *c = base;
            c->nUE = sweep_lo + k * sweep_step;
            c->seed = (uint64_t)s_;
        }
    if (variant == PRACH_VARIANT_NOMA_C) {
        char path[1024];
        snprintf(path, sizeof path, "%s/TestResults", outdir);
        mkdir(path, 0755);
    } else {
        if (base.uniform) printf("Traffic model: Uniform\n\n"); /* WithNOMA:208-213 */
        else printf("Traffic model: Beta\n\n");
        fflush(stdout);
    }

    /* deal the trials to the workers: longest first onto the least loaded (cost = prach_trial_cost, a measured table); glibc mode deals whole seeds */
    if (devlist) {
        gpus = 0;
        for (const char *q = devlist; *q && gpus < 64;) {
            devs[gpus++] = atoi(q);
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
        if (gpus < 1) die("--devices LIST: comma-separated HIP ordinals");
    } else {
        for (int w = 0; w < 64; w++) devs[w] = device + w;
    }
    if (gpus > ntr) gpus = ntr;
    if (glibc && gpus > randomMax) gpus = randomMax;
    int **widx = (int **)calloc((size_t)gpus, sizeof(int *));
    int *wn = (int *)calloc((size_t)gpus, sizeof(int));
    double *wload = (double *)calloc((size_t)gpus, sizeof(double));
    if (!widx || !wn || !wload) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
    for (int w = 0; w < gpus; w++) {
        widx[w] = (int *)malloc(sizeof(int) * (size_t)ntr);
        if (!widx[w]) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
    }
    if (glibc) {
        for (int s_ = 0; s_ < randomMax; s_++) { /* seeds cost the same: round robin keeps every worker's seeds in grid order */
            const int w = s_ % gpus;
            for (int k = 0; k < npts; k++) widx[w][wn[w]++] = s_ * npts + k;
        }
    } else {
        for (int k = npts - 1; k >= 0; k--) /* sweep points are ascending in nUE: descending cost */
            for (int s_ = 0; s_ < randomMax; s_++) {
                int best = 0;
                for (int w = 1; w < gpus; w++) if (wload[w] < wload[best]) best = w;
                widx[best][wn[best]++] = s_ * npts + k;
                wload[best] += prach_trial_cost(&cfgs[s_ * npts + k]);
            }
    }

    if (gpus == 1) {
        int rcw = run_worker(devs[0], cfgs, widx[0], wn[0], res, lat, want_logs, outdir, glibc, npts);
        if (rcw) return rcw;
    } else {
        /* one child per device, forked BEFORE this process touches HIP (a forked copy of an initialised runtime is not usable) */
        pid_t *pid = (pid_t *)calloc((size_t)gpus, sizeof(pid_t));
        if (!pid) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
        for (int w = 0; w < gpus; w++) {
            pid[w] = fork();
            if (pid[w] < 0) { perror("prach_sim: fork"); return 2; }
            if (pid[w] == 0) _exit(run_worker(devs[w], cfgs, widx[w], wn[w], res, lat, want_logs, outdir, glibc, npts));
        }
        int bad = 0;
        for (int w = 0; w < gpus; w++) {
            int st = 0;
            if (waitpid(pid[w], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                fprintf(stderr, "prach_sim: the worker of device %d failed\n", devs[w]);
                bad = 1;
            }
        }
        free(pid);
        if (bad) return 2;
    }

    /* the parent prints in the reference's order and merges */
    char text[4096];
    if (variant == PRACH_VARIANT_NOMA_C) { /* NOMA.c main: no banner, one line per trial, "Done" per seed */
        char line[256], path[1024];
        for (int s_ = 0; s_ < randomMax; s_++) {
            for (int k = 0; k < npts; k++) {
                const int q = s_ * npts + k;
                prach_format_noma_line(&cfgs[q], &res[q], line, sizeof line);
                fputs(line, stdout);
                snprintf(path, sizeof path, "%s/TestResults/Sector_%d_Result.txt", outdir, cfgs[q].nUE); /* NOMA.c:603-605 */
                FILE *fp = fopen(path, "a");
                if (!fp) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(PRACH_ERR_IO)); return 2; }
                fputs(line, fp);
                fclose(fp);
            }
            printf("Done\n"); /* NOMA.c:716 */
        }
        return 0;
    }
    for (int q = 0; q < ntr; q++) {
        prach_format_stdout(&cfgs[q], &res[q], lat[q], text, sizeof text);
        fputs(text, stdout);
    }
    if (csv_path) { /* one row per sweep point: mean over the seeds in seed order (AveragePerformance.py:8-24), np.around(., 3), csv.writer's float repr and CRLF */
        FILE *fp = fopen(csv_path, "wb");
        if (!fp) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(PRACH_ERR_IO)); return 2; }
        for (int k = 0; k < npts; k++) {
            double acc[6] = {0, 0, 0, 0, 0, 0};
            for (int s_ = 0; s_ < randomMax; s_++) {
                prach_format_results(&cfgs[s_ * npts + k], &res[s_ * npts + k], lat[s_ * npts + k], text, sizeof text);
                prach_results_csv_accumulate(acc, text);
            }
            const size_t n = prach_results_csv_row(acc, randomMax, text, sizeof text);
            fwrite(text, 1, n, fp);
        }
        fclose(fp);
    }
    for (int w = 0; w < gpus; w++) free(widx[w]);
    free(widx); free(wn); free(wload); free(cfgs);
    return 0;
}
