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
 * variant draws from Philox (the reference's rand() stream position is data dependent there).
 *
 * Extensions (not in the reference): --program beta|withnoma|noma, --rng glibc|philox, --nue N,
 * --sweep LO:HI:STEP, --out DIR, --logs 0|1, --device N, --csv FILE (--program beta: the results.csv of
 * AveragePerformance.py over the --times seeds of every sweep point, written by prach_results_csv_*).
 */
#define _GNU_SOURCE
#include "../../include/prach.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

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

int main(int argc, char *argv[]) {
    int randomMax = 1, variant = PRACH_VARIANT_WITHNOMA_C, rng = PRACH_RNG_GLIBC, device = 0, want_logs = 1;
    int sweep_lo = 10000, sweep_hi = 100000, sweep_step = 10000; /* WithNOMA:221 */
    const char *outdir = ".", *csv_path = NULL;
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
        } else if (strcmp(a, "--csv") == 0) {
            csv_path = v;
        } else {
            usage_and_exit();
        }
    }
    base.rng_mode = rng;
    if (csv_path && variant != PRACH_VARIANT_BETA_C) die("--csv needs --program beta (AveragePerformance.py reads its six-number Results.txt)");

    if (variant == PRACH_VARIANT_NOMA_C) { /* NOMA.c main: no banner, one line per trial, "Done" per seed */
        prach_engine *eng = NULL;
        int rc = prach_engine_create(device, &eng);
        if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }
        char line[256], path[1024];
        snprintf(path, sizeof path, "%s/TestResults", outdir);
        mkdir(path, 0755);
        for (int seed = 0; seed < randomMax; seed++) {
            for (int n = sweep_lo; n <= sweep_hi; n += sweep_step) {
                prach_cfg c = base;
                c.nUE = n; c.seed = (uint64_t)seed; c.rng_mode = PRACH_RNG_PHILOX;
                prach_result r;
                rc = prach_run_trials(eng, &c, 1, &r, NULL);
                if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }
                prach_format_noma_line(&c, &r, line, sizeof line);
                fputs(line, stdout);
                snprintf(path, sizeof path, "%s/TestResults/Sector_%d_Result.txt", outdir, n); /* NOMA.c:603-605 */
                FILE *fp = fopen(path, "a");
                if (!fp) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(PRACH_ERR_IO)); return 2; }
                fputs(line, fp);
                fclose(fp);
            }
            printf("Done\n"); /* NOMA.c:716 */
        }
        prach_engine_destroy(eng);
        return 0;
    }

    if (base.uniform) printf("Traffic model: Uniform\n\n"); /* WithNOMA:208-213 */
    else printf("Traffic model: Beta\n\n");

    prach_engine *eng = NULL;
    int rc = prach_engine_create(device, &eng);
    if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }

    const int npts = (sweep_hi - sweep_lo) / sweep_step + 1;
    char text[4096];
    double (*csv_acc)[6] = csv_path ? (double (*)[6])calloc((size_t)npts, sizeof(double[6])) : NULL; /* AveragePerformance.py:8 */
    if (csv_path && !csv_acc) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
    if (rng == PRACH_RNG_PHILOX) {
        /* Philox trials are independent: the whole --times x sweep grid runs concurrently in ONE call
         * (one workgroup cluster per trial); output order and files are the reference's. */
        const int ntr = randomMax * npts;
        prach_cfg *cfgs = (prach_cfg *)malloc(sizeof(prach_cfg) * (size_t)ntr);
        prach_result *res = (prach_result *)malloc(sizeof(prach_result) * (size_t)ntr);
        prach_ue_log **logs = want_logs ? (prach_ue_log **)calloc((size_t)ntr, sizeof(prach_ue_log *)) : NULL;
        if (!cfgs || !res || (want_logs && !logs)) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
        for (int s_ = 0; s_ < randomMax; s_++)
            for (int k = 0; k < npts; k++) {
                prach_cfg *c = &cfgs[s_ * npts + k];
                *c = base;
                c->nUE = sweep_lo + k * sweep_step;
                c->seed = (uint64_t)s_;
                if (want_logs) {
                    logs[s_ * npts + k] = (prach_ue_log *)malloc(sizeof(prach_ue_log) * (size_t)c->nUE);
                    if (!logs[s_ * npts + k]) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
                }
            }
        struct timespec ts0, ts1;
        clock_gettime(CLOCK_MONOTONIC, &ts0);
        rc = prach_run_trials(eng, cfgs, ntr, res, logs);
        if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }
        clock_gettime(CLOCK_MONOTONIC, &ts1);
        const double lat = (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
        for (int k = 0; k < ntr; k++) {
            prach_format_stdout(&cfgs[k], &res[k], lat, text, sizeof text);
            fputs(text, stdout);
            rc = prach_write_trial_files(&cfgs[k], &res[k], want_logs ? logs[k] : NULL, lat, outdir);
            if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }
            if (want_logs) free(logs[k]);
        }
        if (csv_acc)
            for (int k = 0; k < npts; k++)
                for (int s_ = 0; s_ < randomMax; s_++) { /* seed order inside a point, AveragePerformance.py:10-19 */
                    prach_format_results(&cfgs[s_ * npts + k], &res[s_ * npts + k], lat, text, sizeof text);
                    prach_results_csv_accumulate(csv_acc[k], text);
                }
        free(cfgs); free(res); free(logs);
        goto write_csv;
    }
    /* glibc mode: within one seed the sweep is chained through the rand() stream (one srand() per seed, WithNOMA:219-221),
     * but different seeds are independent — so the loop nest is turned inside out: for every nUE point, the trials of ALL
     * seeds run concurrently in one call, each continuing its own seed's stream.  Output is buffered per seed and
     * printed in the reference's order. */
    {
        uint64_t *offset = (uint64_t *)calloc((size_t)randomMax, sizeof(uint64_t));
        prach_cfg *cfgs = (prach_cfg *)malloc(sizeof(prach_cfg) * (size_t)randomMax);
        prach_result *res = (prach_result *)malloc(sizeof(prach_result) * (size_t)randomMax);
        prach_ue_log **logs = want_logs ? (prach_ue_log **)calloc((size_t)randomMax, sizeof(prach_ue_log *)) : NULL;
        char **outtxt = (char **)calloc((size_t)randomMax, sizeof(char *));
        size_t *outlen = (size_t *)calloc((size_t)randomMax, sizeof(size_t));
        if (!offset || !cfgs || !res || !outtxt || !outlen || (want_logs && !logs)) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
        for (int s_ = 0; s_ < randomMax; s_++) {
            outtxt[s_] = (char *)malloc((size_t)npts * 1024);
            if (want_logs) logs[s_] = (prach_ue_log *)malloc(sizeof(prach_ue_log) * (size_t)sweep_hi);
            if (!outtxt[s_] || (want_logs && !logs[s_])) { fprintf(stderr, "prach_sim: out of memory\n"); return 2; }
        }
        struct timespec ts0, ts1;
        clock_gettime(CLOCK_MONOTONIC, &ts0);
        for (int k = 0; k < npts; k++) {
            for (int s_ = 0; s_ < randomMax; s_++) {
                cfgs[s_] = base;
                cfgs[s_].nUE = sweep_lo + k * sweep_step;
                cfgs[s_].seed = (uint64_t)s_;
                cfgs[s_].stream_offset = offset[s_];
            }
            rc = prach_run_trials(eng, cfgs, randomMax, res, logs);
            if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }
            clock_gettime(CLOCK_MONOTONIC, &ts1);
            const double lat = (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
            for (int s_ = 0; s_ < randomMax; s_++) {
                offset[s_] += res[s_].draws;
                outlen[s_] += prach_format_stdout(&cfgs[s_], &res[s_], lat, outtxt[s_] + outlen[s_], 1024);
                rc = prach_write_trial_files(&cfgs[s_], &res[s_], want_logs ? logs[s_] : NULL, lat, outdir);
                if (rc != PRACH_OK) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(rc)); return 2; }
                if (csv_acc) { prach_format_results(&cfgs[s_], &res[s_], lat, text, sizeof text); prach_results_csv_accumulate(csv_acc[k], text); }
            }
        }
        for (int s_ = 0; s_ < randomMax; s_++) { fwrite(outtxt[s_], 1, outlen[s_], stdout); free(outtxt[s_]); if (want_logs) free(logs[s_]); }
        free(offset); free(cfgs); free(res); free(logs); free(outtxt); free(outlen);
    }
write_csv:
    if (csv_acc) { /* one row per sweep point: mean over the seeds, np.around(., 3), csv.writer's float repr and CRLF */
        FILE *fp = fopen(csv_path, "wb");
        if (!fp) { fprintf(stderr, "prach_sim: %s\n", prach_strerror(PRACH_ERR_IO)); return 2; }
        for (int k = 0; k < npts; k++) {
            const size_t n = prach_results_csv_row(csv_acc[k], randomMax, text, sizeof text);
            fwrite(text, 1, n, fp);
        }
        fclose(fp);
        free(csv_acc);
    }
    prach_engine_destroy(eng);
    return 0;
}
