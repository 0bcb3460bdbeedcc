/* prach_host.c — host-side half of the C ABI that needs no device: parameter defaults and
 * validation, the deterministic arrival schedule, the glibc rand() stream, and the reference's
 * text surfaces (stdout block, Results.txt, Logs.txt).  Plain C like the reference.
 *
 * Reference lines are cited per function; nothing here is copied from the reference — the
 * formats are reproduced because the drop-in boundary of this project IS those files
 * (SURVEY.md §8b).
 */
#define _GNU_SOURCE
#include "../../include/prach.h"

#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

/* Beta.c:47-57 (BETA_C) / WithNOMA:70-88 (WITHNOMA_C): the hard-coded / default parameters */
void prach_cfg_defaults(prach_cfg *c, int variant) {
    memset(c, 0, sizeof(*c));
    c->variant = variant;
    c->uniform = 0;
    c->nUE = 100000;
    c->nPreamble = 54;
    c->backoff = 20;
    c->nGrantUL = variant == PRACH_VARIANT_BETA_C ? 54 : 12;
    c->maxRarWindow = 6;
    c->maxMsg2TxCount = 9;
    c->accessTime = 5;
    c->rng_mode = PRACH_RNG_GLIBC;
    c->seed = 0;
    c->stream_offset = 0;
    c->cellRadius = 400.0f;
    c->hBS = 10.0f;
    c->hUT = 1.8f;
    if (variant == PRACH_VARIANT_NOMA_C) { /* NOMA.c:41-57 */
        c->nGrantUL = 2;        /* per sector */
        c->maxRarWindow = 5;
        c->maxMsg2TxCount = 10; /* maxMsg1ReTx */
        c->cellRadius = 500.0f;
        c->rng_mode = PRACH_RNG_PHILOX;
    }
}

int prach_cfg_validate(const prach_cfg *c) {
    if (!c) return PRACH_ERR_ARG;
    if (c->variant != PRACH_VARIANT_BETA_C && c->variant != PRACH_VARIANT_WITHNOMA_C && c->variant != PRACH_VARIANT_NOMA_C)
        return PRACH_ERR_ARG;
    if (c->nUE < 1 || c->nPreamble < 1 || c->backoff < 1 || c->nGrantUL < 1 || c->accessTime < 1) return PRACH_ERR_ARG;
    if (c->maxRarWindow < 1 || c->maxMsg2TxCount < 0 || c->max_steps < 0) return PRACH_ERR_ARG;
    if (c->rng_mode != PRACH_RNG_GLIBC && c->rng_mode != PRACH_RNG_PHILOX) return PRACH_ERR_ARG;
    if (c->nPreamble > 254 || c->maxRarWindow > 255 || c->maxMsg2TxCount > 255) return PRACH_ERR_UNSUPPORTED;
    if (c->nUE > (1 << 24)) return PRACH_ERR_UNSUPPORTED;
    if (c->flags & ~(PRACH_FLAG_SECTOR_GRANTS | PRACH_FLAG_NOMA_NONSECTOR)) return PRACH_ERR_ARG;
    if ((c->flags & PRACH_FLAG_SECTOR_GRANTS) && c->variant != PRACH_VARIANT_WITHNOMA_C) return PRACH_ERR_ARG; /* sectors exist in that program only */
    if ((c->flags & PRACH_FLAG_NOMA_NONSECTOR) && c->variant != PRACH_VARIANT_NOMA_C) return PRACH_ERR_ARG;
    /* NOMA.c:167-172 redraws the UE's distance until it exceeds 35 m: with a cell radius of 35 m or less the reference itself never returns */
    if (c->variant == PRACH_VARIANT_NOMA_C && !(c->cellRadius > 35.0f && c->cellRadius < 1e9f)) return PRACH_ERR_ARG;
    return PRACH_OK;
}

int prach_max_time(const prach_cfg *c) { return c->uniform ? 60000 : 10000; } /* Beta.c:92,103 */

/* What a trial costs inside a batched launch, in kernel microseconds: the unit the multi-GPU dealing balances (dist.py shard_trials, prach_sim --gpus).
   Measured on an MI355X, 1024 trials of one size per call (scripts/gpu_cost_table.py -> profiles/r04_cost_table.json): the Beta.c program as committed (54 UL
   grants: every UE is served, cost grows with the contention of the big points) and RandomAccessWithNOMA's defaults (12 grants: overloaded from 20 000 UEs
   on, linear in nUE); linear between the sweep's points, proportional beyond them, scaled by the subframes a shortened trial runs.  Uniform arrivals
   (60 000 subframes, a handful of live UEs each) and NOMA.c are costed by size alone: their sweeps deal evenly whatever the constant. */
double prach_trial_cost(const prach_cfg *c) {
    static const double beta_us[10] = {55.67, 81.6, 93.88, 105.34, 119.79, 156.88, 221.75, 293.07, 373.64, 455.66};
    static const double over_us[10] = {57.24, 142.01, 219.44, 298.18, 376.92, 457.74, 538.67, 617.48, 693.07, 769.03};
    const double n = (double)c->nUE;
    const int maxt = prach_max_time(c);
    const double frac = (c->max_steps > 0 && c->max_steps < maxt) ? (double)c->max_steps / (double)maxt : 1.0;
    if (c->variant == PRACH_VARIANT_NOMA_C) return 2.0e-3 * n * frac;
    if (c->uniform) return 1.2e-3 * n * frac;
    /* overloaded: more UEs than the UL grants of the whole trial can serve (nGrantUL - 1 per 5 ms window, Beta.c:112,336) */
    const double capacity = (double)(c->nGrantUL > 1 ? c->nGrantUL - 1 : 0) * (double)maxt / 5.0;
    const double *tab = (c->variant == PRACH_VARIANT_WITHNOMA_C || n > capacity) ? over_us : beta_us;
    double x = n / 10000.0 - 1.0; /* position in the table: 0 at nUE = 10 000 */
    if (x <= 0.0) return tab[0] * n / 10000.0 * frac;
    if (x >= 9.0) return tab[9] * n / 100000.0 * frac;
    const int k = (int)x;
    return (tab[k] + (tab[k + 1] - tab[k]) * (x - (double)k)) * frac;
}

const char *prach_strerror(int s) {
    switch (s) {
    case PRACH_OK: return "ok";
    case PRACH_ERR_ARG: return "invalid argument";
    case PRACH_ERR_UNSUPPORTED: return "parameter outside the supported range (nPreamble<=254, maxRarWindow<=255, maxMsg2TxCount<=255, nUE<=2^24; NOMA_C: nPreamble<=64, Beta arrivals)";
    case PRACH_ERR_DEVICE: return "HIP device/runtime error (an MI355X/gfx950 device is required; there is no CPU fallback)";
    case PRACH_ERR_STREAM: return "glibc draw stream exhausted";
    case PRACH_ERR_INTERNAL: return "device-side consistency check failed";
    case PRACH_ERR_IO: return "I/O error";
    case PRACH_ERR_TIMEOUT: return "a workgroup of a cluster waited too long for a peer (cluster not co-resident)";
    default: return "unknown status";
    }
}

/* Arrival process.  Beta: every accessTime ms, activeCheck += ceil(nUE * pdf(t/maxTime) / (maxTime/accessTime))
 * with the reference's mixed float/double evaluation (Beta.c:127-128, beta_dist Beta.c:516-519:
 * the normaliser is the literal 0.0165, `1 - x` is a float subtraction, pow() runs in double, the
 * product is rounded to float on return).  Uniform: += ceil(n*accessTime/60000), min 1 (Beta.c:95-100).
 * The process uses no random numbers, so it is a table. */
static float beta34_density(float x) {
    const float a = 3, b = 4;
    float v = (1 / 0.0165) * (pow(x, (a - 1))) * (pow((1 - x), (b - 1)));
    return v;
}

int prach_arrival_schedule(const prach_cfg *c, int32_t *out, int cap, int32_t *nAccessUEo) {
    const int nUE = c->nUE, aT = c->accessTime, maxTime = prach_max_time(c);
    int nAccessUE = 0;
    if (c->uniform) {
        nAccessUE = ceil((float)nUE * (float)aT * 1.0 / (float)maxTime);
        if (nAccessUE <= 0) nAccessUE = 1;
    }
    if (nAccessUEo) *nAccessUEo = nAccessUE;
    int activeCheck = 0, s = 0;
    for (int time = 0; time < maxTime; time += aT, s++) {
        if (activeCheck < nUE) {
            if (c->uniform) {
                activeCheck += nAccessUE;
            } else {
                float pdf = beta34_density((float)time / (float)maxTime);
                activeCheck += (int)ceil((float)nUE * pdf / ((float)maxTime / (float)aT));
            }
            if (activeCheck >= nUE) activeCheck = nUE;
        }
        if (s < cap) out[s] = activeCheck;
    }
    return s;
}

/* glibc srand()/rand() (stdlib/random_r.c, TYPE_3: degree 31, separation 3, 310 warm-up draws).
 * The reference links libc's rand(); reproducing its exact stream is what makes per-trial counts
 * bit-exact "under identical seeds".  Checked against libc itself in tests/test_host_logic.py. */
void prach_glibc_stream(uint32_t seed, uint64_t first, uint64_t n, int32_t *out) {
    uint32_t r[31];
    int32_t word = (int32_t)(seed == 0 ? 1u : seed);
    r[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
        int32_t hi = word / 127773, lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (uint32_t)word;
    }
    int f = 3, b = 0;
    const uint64_t skip = 310 + first;
    for (uint64_t k = 0; k < skip + n; k++) {
        r[f] += r[b];
        if (k >= skip) out[k - skip] = (int32_t)(r[f] >> 1);
        if (++f == 31) f = 0;
        if (++b == 31) b = 0;
    }
}

/* ---- NOMA.c activation table ---------------------------------------------------------------- */
static uint32_t philox31(uint64_t seed, uint32_t nUE, uint32_t variant, uint32_t ue, uint32_t k) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t c0 = ue, c1 = k, c2 = nUE, c3 = variant, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0 >> 1;
}

int prach_noma_activation_table(const prach_cfg *c, int32_t *preamble0, int32_t *sector, double *gain, double *lgain,
                                uint32_t *ndraws) {
    if (!c) return PRACH_ERR_ARG;
    return prach_noma_activation_range(c, 0, c->nUE, preamble0, sector, gain, lgain, ndraws);
}

/* UEs [lo, hi) of the table (outputs indexed from lo): every UE's draws are its own Philox counter, so ranges are
 * independent and the engine deals them to all host cores. */
int prach_noma_activation_range(const prach_cfg *c, int lo, int hi, int32_t *preamble0, int32_t *sector, double *gain, double *lgain,
                                uint32_t *ndraws) {
    if (!c || !preamble0 || !sector || !gain || !lgain || !ndraws || lo < 0 || hi > c->nUE || lo > hi) return PRACH_ERR_ARG;
    preamble0 -= lo; sector -= lo; gain -= lo; lgain -= lo; ndraws -= lo;
    const float pi = 3.14; /* NOMA.c:55 */
    const float cellRadius = c->cellRadius;
    for (int i = lo; i < hi; i++) {
        uint32_t k = 0;
#define DRAW() ((int)philox31(c->seed, (uint32_t)c->nUE, PRACH_VARIANT_NOMA_C, (uint32_t)i, k++))
        preamble0[i] = DRAW() % c->nPreamble; /* NOMA.c:133 */
        float angle = (float)DRAW() / (float)(2147483647) * 2 * pi; /* NOMA.c:142 */
        int sec;
        if (angle >= 0 && angle < ((1. / 3.) * pi)) sec = 0; /* NOMA.c:146-163 */
        else if (angle >= ((1. / 3.) * pi) && angle < ((2. / 3.) * pi)) sec = 1;
        else if (angle >= ((2. / 3.) * pi) && angle < 3.14) sec = 2;
        else if (angle >= pi && angle < ((4. / 3.) * pi)) sec = 3;
        else if (angle >= ((4. / 3.) * pi) && angle < ((5. / 3.) * pi)) sec = 4;
        else sec = 5;
        sector[i] = sec;
        float r;
        while (1) { /* NOMA.c:167-172 */
            r = cellRadius * sqrt((float)DRAW() / (float)2147483647);
            if (r > 35.0) break;
        }
        float x = r * cos(angle), y = r * sin(angle), pathloss; /* NOMA.c:176-178 */
        double env = sqrt(x * x + y * y);
        double ch_g = 0, rayleigh;
        while (ch_g < 1e-7) { /* NOMA.c:185-189 */
            pathloss = sqrt(1 + pow(env, 2));
            rayleigh = sqrt(-2 * log((double)DRAW() / (double)2147483647));
            ch_g = pow(rayleigh / pathloss, 2);
        }
#undef DRAW
        gain[i] = ch_g;
        lgain[i] = log(ch_g);
        ndraws[i] = k;
    }
    return PRACH_OK;
}

/* activeUE (NOMA.c:131-192) for ONE UE in the reference's own rand() stream: the draws are stream[*pos], stream[*pos + 1], ... in the
 * order the reference consumes them (preamble, angle, the radius rejection loop, the Rayleigh-gain rejection loop); *pos advances by the
 * number consumed.  Used by the glibc-mode NOMA path (prach_noma_glibc.hip), where the host activates the arrivals of an access slot between
 * two device steps, with the same libm the reference links.  PRACH_ERR_STREAM: the window is exhausted (the engine retries with a larger one). */
int prach_noma_activation_stream(const prach_cfg *c, const int32_t *stream, uint64_t *pos, uint64_t avail, int32_t *preamble0, int32_t *sector,
                                 double *gain, double *lgain) {
    if (!c || !stream || !pos || !preamble0 || !sector || !gain || !lgain) return PRACH_ERR_ARG;
    const float pi = 3.14; /* NOMA.c:55 */
    const float cellRadius = c->cellRadius;
    uint64_t k = *pos;
#define DRAW() (k < avail ? (int)stream[k++] : (k++, -1))
    *preamble0 = DRAW() % c->nPreamble; /* NOMA.c:133 */
    float angle = (float)DRAW() / (float)(2147483647) * 2 * pi; /* NOMA.c:142 */
    int sec;
    if (angle >= 0 && angle < ((1. / 3.) * pi)) sec = 0; /* NOMA.c:146-163 */
    else if (angle >= ((1. / 3.) * pi) && angle < ((2. / 3.) * pi)) sec = 1;
    else if (angle >= ((2. / 3.) * pi) && angle < 3.14) sec = 2;
    else if (angle >= pi && angle < ((4. / 3.) * pi)) sec = 3;
    else if (angle >= ((4. / 3.) * pi) && angle < ((5. / 3.) * pi)) sec = 4;
    else sec = 5;
    *sector = sec;
    float r;
    while (1) { /* NOMA.c:167-172 */
        if (k >= avail) return PRACH_ERR_STREAM;
        r = cellRadius * sqrt((float)DRAW() / (float)2147483647);
        if (r > 35.0) break;
    }
    float x = r * cos(angle), y = r * sin(angle), pathloss; /* NOMA.c:176-178 */
    double env = sqrt(x * x + y * y);
    double ch_g = 0, rayleigh;
    while (ch_g < 1e-7) { /* NOMA.c:185-189 */
        if (k >= avail) return PRACH_ERR_STREAM;
        pathloss = sqrt(1 + pow(env, 2));
        rayleigh = sqrt(-2 * log((double)DRAW() / (double)2147483647));
        ch_g = pow(rayleigh / pathloss, 2);
    }
#undef DRAW
    if (k > avail) return PRACH_ERR_STREAM;
    *gain = ch_g;
    *lgain = log(ch_g);
    *pos = k;
    return PRACH_OK;
}

size_t prach_format_noma_line(const prach_cfg *c, const prach_result *r, char *buf, size_t cap) {
    char tmp[256];
    int n = snprintf(tmp, sizeof tmp, "%d %d %lf %lf %lf\n", c->nUE, r->nSuccessUE, ((float)r->nSuccessUE / (float)c->nUE) * 100.0,
                     ((float)r->preambleTxCount / (float)r->nSuccessUE), ((float)(int)r->sumTimer / (float)r->nSuccessUE));
    if (buf && (size_t)n < cap) memcpy(buf, tmp, (size_t)n + 1);
    return (size_t)n;
}

/* ---- glibc rand() stream: jump-ahead (for the device-side generator) ------------------------------
 * Linear form of the TYPE_3 generator: r[0..30] from the seed, r[31..33] = r[0..2],
 * r[i] = r[i-31] + r[i-3] (mod 2^32) for i >= 34, and the k-th rand() = r[344+k] >> 1.  A window
 * W_n = (r[n-30] .. r[n]) advances by the 31x31 companion matrix A over Z/2^32, so W_{n+D} = A^D W_n:
 * powers A^(2^j) are cached, a jump costs <= 40 mat-vecs.  prach_internal_glibc_seeds returns, for each
 * chunk c of `chunk` outputs starting at output `first`, the window that precedes its first value. */
typedef struct { uint32_t m[31][31]; } lfg_mat;
static lfg_mat lfg_pow2[48];
static pthread_once_t lfg_pow2_once = PTHREAD_ONCE_INIT; /* engines of several host threads may need it at once */

static void lfg_matmul(const lfg_mat *a, const lfg_mat *b, lfg_mat *out) {
    for (int i = 0; i < 31; i++)
        for (int j = 0; j < 31; j++) {
            uint32_t acc = 0;
            for (int k = 0; k < 31; k++) acc += a->m[i][k] * b->m[k][j];
            out->m[i][j] = acc;
        }
}
static void lfg_matvec(const lfg_mat *a, const uint32_t *w, uint32_t *out) {
    for (int i = 0; i < 31; i++) {
        uint32_t acc = 0;
        for (int k = 0; k < 31; k++) acc += a->m[i][k] * w[k];
        out[i] = acc;
    }
}
static void lfg_build_pows(void) {
    memset(&lfg_pow2[0], 0, sizeof(lfg_mat));
    for (int j = 0; j < 30; j++) lfg_pow2[0].m[j][j + 1] = 1; /* w'[j] = w[j+1] */
    lfg_pow2[0].m[30][0] = 1; lfg_pow2[0].m[30][28] = 1;      /* w'[30] = r[n+1] = r[n-30] + r[n-2] = w[0] + w[28] */
    for (int j = 1; j < 48; j++) lfg_matmul(&lfg_pow2[j - 1], &lfg_pow2[j - 1], &lfg_pow2[j]);
}
static void lfg_init_pows(void) { pthread_once(&lfg_pow2_once, lfg_build_pows); }
static void lfg_jump(uint32_t *w, uint64_t d) {
    uint32_t t[31];
    for (int j = 0; d; j++, d >>= 1)
        if (d & 1) { lfg_matvec(&lfg_pow2[j], w, t); memcpy(w, t, sizeof t); }
}

void prach_internal_glibc_seeds(uint32_t seed, uint64_t first, uint64_t nchunks, uint64_t chunk, uint32_t *out /* [nchunks][31] */) {
    lfg_init_pows();
    uint32_t r[34];
    int32_t word = (int32_t)(seed == 0 ? 1u : seed);
    r[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
        int32_t hi = word / 127773, lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (uint32_t)word;
    }
    r[31] = r[0]; r[32] = r[1]; r[33] = r[2];
    uint32_t w[31];
    for (int j = 0; j < 31; j++) w[j] = r[3 + j];         /* W_33 */
    lfg_jump(w, 310ull + first);                           /* W_{343+first}: the next value is r[344+first] */
    lfg_mat step;
    {   /* A^chunk */
        lfg_mat acc, tmp;
        memset(&acc, 0, sizeof acc);
        for (int i = 0; i < 31; i++) acc.m[i][i] = 1;
        uint64_t d = chunk;
        for (int j = 0; d; j++, d >>= 1)
            if (d & 1) { lfg_matmul(&lfg_pow2[j], &acc, &tmp); acc = tmp; }
        step = acc;
    }
    for (uint64_t c = 0; c < nchunks; c++) {
        memcpy(out + c * 31, w, sizeof w);
        uint32_t t[31];
        lfg_matvec(&step, w, t);
        memcpy(w, t, sizeof t);
    }
}

/* ---- text surfaces ---------------------------------------------------------------------------- */

/* The per-UE log is 24 MB of text per 100k-UE trial (Beta.c:501-508): once the trial itself takes 85 ms the formatter is
 * what the caller waits for, so the line is assembled by hand (fixed labels + decimal integers) instead of snprintf. */
static char *put_int(char *p, int32_t v) {
    uint32_t u = (uint32_t)v;
    if (v < 0) { *p++ = '-'; u = 0u - u; }
    char tmp[10];
    int n = 0;
    do { tmp[n++] = (char)('0' + u % 10u); u /= 10u; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}
#define PUT_LIT(s) do { memcpy(p, s, sizeof(s) - 1); p += sizeof(s) - 1; } while (0)
#define PRACH_LOG_LINE_MAX 384 /* 213 bytes of labels and newline + 15 integers of at most 11 characters = 378 */

static size_t format_log_line(const prach_ue_log *u, char *line) {
    char *p = line;
    PUT_LIT("Idx: "); p = put_int(p, u->idx);
    PUT_LIT(" | Timer: "); p = put_int(p, u->timer);
    PUT_LIT(" | Active: "); p = put_int(p, u->active);
    PUT_LIT(" | txTime: "); p = put_int(p, u->txTime);
    PUT_LIT(" | FirstTxTime: "); p = put_int(p, u->firstTxTime);
    PUT_LIT(" | SecondTxTime: "); p = put_int(p, u->secondTxTime);
    PUT_LIT(" | NowBackoff: "); p = put_int(p, u->nowBackoff);
    PUT_LIT(" | Preamble: "); p = put_int(p, u->preamble);
    PUT_LIT(" | Preamble change: "); p = put_int(p, u->preambleChange);
    PUT_LIT(" | RAR window: "); p = put_int(p, u->rarWindow);
    PUT_LIT(" | Max RAR: "); p = put_int(p, u->maxRarCounter);
    PUT_LIT(" | Preamble reTx: "); p = put_int(p, u->preambleTxCounter);
    PUT_LIT(" | MSG 2 Flag: "); p = put_int(p, u->msg2Flag);
    PUT_LIT(" | ConnectRequest: "); p = put_int(p, u->connectionRequest);
    PUT_LIT(" | MSG 4 Flag: "); p = put_int(p, u->msg4Flag);
    *p++ = '\n';
    return (size_t)(p - line);
}

size_t prach_format_logs(const prach_ue_log *ue, int nUE, char *buf, size_t cap) {
    size_t off = 0;
    char line[PRACH_LOG_LINE_MAX];
    for (int i = 0; i < nUE; i++) {
        if (buf && off + PRACH_LOG_LINE_MAX <= cap) { /* room for any line: format in place */
            off += format_log_line(ue + i, buf + off);
            continue;
        }
        const size_t n = format_log_line(ue + i, line);
        if (buf && off + n <= cap) memcpy(buf + off, line, n);
        off += n;
    }
    return off;
}

typedef struct { float ratioSuccess, nCollisionPreambles, averagePreambleTx, averageDelay; } derived_t;

/* the float arithmetic of saveSimulationLog (Beta.c:434-438) */
static derived_t derive(const prach_cfg *c, const prach_result *r) {
    derived_t d;
    d.ratioSuccess = (float)r->nSuccessUE / (float)c->nUE * 100.0;
    d.nCollisionPreambles = (float)r->collisionPreambles / ((float)c->nUE * (float)c->nPreamble);
    d.averagePreambleTx = (float)r->preambleTxCount / (float)r->nSuccessUE;
    d.averageDelay = r->totalDelay / (float)r->nSuccessUE;
    return d;
}

size_t prach_format_results(const prach_cfg *c, const prach_result *r, double latency_s, char *buf, size_t cap) {
    derived_t d = derive(c, r);
    char tmp[1024];
    int n;
    if (c->variant == PRACH_VARIANT_WITHNOMA_C)
        n = snprintf(tmp, sizeof tmp,
                     "%d\n%.2lf\n%d\n%.2lf\n%.2lf\nNumber of total preamble tx: %d\nFinally Falied: %d\nFinally Success: %lf\n",
                     c->nUE, d.ratioSuccess, r->nSuccessUE, d.averagePreambleTx, d.averageDelay, r->preambleTxCount,
                     r->continueFaliedUEs,
                     (float)r->finalSuccessUEs / (float)(r->continueFaliedUEs + r->finalSuccessUEs));
    else
        n = snprintf(tmp, sizeof tmp, "%d\n%.2lf\n%d\n%.2lf\n%.2lf\n%lf", c->nUE, d.ratioSuccess, r->nSuccessUE,
                     d.averagePreambleTx, d.averageDelay, latency_s);
    if (buf && (size_t)n < cap) memcpy(buf, tmp, (size_t)n + 1);
    return (size_t)n;
}

size_t prach_format_stdout(const prach_cfg *c, const prach_result *r, double latency_s, char *buf, size_t cap) {
    derived_t d = derive(c, r);
    char tmp[2048];
    int n = snprintf(tmp, sizeof tmp, "-------- %05d Result ---------\n", r->activeCheck);
    if (c->uniform) n += snprintf(tmp + n, sizeof tmp - n, "Number of RA try UEs per Subframe: %d\n", r->nAccessUE);
    if (c->variant == PRACH_VARIANT_WITHNOMA_C) n += snprintf(tmp + n, sizeof tmp - n, "Fail Counts: %d\n", r->failCounts);
    else n += snprintf(tmp + n, sizeof tmp - n, "Latency: %lf\n", latency_s);
    n += snprintf(tmp + n, sizeof tmp - n,
                  "Number of UEs: %d\nTotal simulation time: %dms\nSuccess ratio: %.2lf\nNumber of succeed UEs: %d\n", c->nUE,
                  r->time_exit, d.ratioSuccess, r->nSuccessUE);
    if (c->variant == PRACH_VARIANT_WITHNOMA_C)
        n += snprintf(tmp + n, sizeof tmp - n, "Number of falied UEs: %d\n", r->continueFaliedUEs);
    n += snprintf(tmp + n, sizeof tmp - n,
                  "Number of collision preambles: %.6lf\nAverage preamble tx count: %.2lf\nAverage delay: %.2lf\n",
                  d.nCollisionPreambles, d.averagePreambleTx, d.averageDelay);
    if (buf && (size_t)n < cap) memcpy(buf, tmp, (size_t)n + 1);
    return (size_t)n;
}

/* Output file names: Beta.c:452-456,490-494 and WithNOMA:754-758,801-805 (note Beta.c's Uniform log
 * name has no seed in it). */
int prach_result_file_name(const prach_cfg *c, int is_log, char *buf, size_t cap) {
    const int seed = (int)c->seed;
    int n;
    if (c->variant == PRACH_VARIANT_WITHNOMA_C) {
        const char *dir = c->uniform ? "NomaUniformResults" : "NomaBetaResults";
        n = is_log ? snprintf(buf, cap, "%s/%d_%d_UE%05d_Logs.txt", dir, seed, c->nPreamble, c->nUE)
                   : snprintf(buf, cap, "%s/%d_%d_%d_Results.txt", dir, seed, c->nPreamble, c->nUE);
    } else {
        const char *dir = c->uniform ? "BasicUniformSimulationResults" : "BasicBetaSimulationResults";
        if (!is_log) n = snprintf(buf, cap, "%s/%d_%d_%d_Results.txt", dir, seed, c->nPreamble, c->nUE);
        else if (c->uniform) n = snprintf(buf, cap, "%s/%d_Exclude_msg2_failures_UE%05d_Logs.txt", dir, c->nPreamble, c->nUE);
        else n = snprintf(buf, cap, "%s/%d_%d_UE%05d_Logs.txt", dir, seed, c->nPreamble, c->nUE);
    }
    return (n > 0 && (size_t)n < cap) ? PRACH_OK : PRACH_ERR_ARG;
}

static int write_all(const char *path, const char *data, size_t n) {
    FILE *fp = fopen(path, "w+");
    if (!fp) return PRACH_ERR_IO;
    size_t w = fwrite(data, 1, n, fp);
    if (fclose(fp) != 0 || w != n) return PRACH_ERR_IO;
    return PRACH_OK;
}

int prach_write_trial_files(const prach_cfg *c, const prach_result *r, const prach_ue_log *ue, double latency_s,
                            const char *root_dir) {
    char rel[512], path[1024], dirp[1024];
    const char *root = (root_dir && *root_dir) ? root_dir : ".";
    int rc = prach_result_file_name(c, 0, rel, sizeof rel);
    if (rc) return rc;
    /* the reference mkdir()s its output directories (WithNOMA:67-68); Beta.c expects them to exist */
    snprintf(dirp, sizeof dirp, "%s/%.*s", root, (int)(strchr(rel, '/') - rel), rel);
    if (mkdir(dirp, 0755) != 0 && errno != EEXIST) return PRACH_ERR_IO;
    char text[1024];
    size_t n = prach_format_results(c, r, latency_s, text, sizeof text);
    snprintf(path, sizeof path, "%s/%s", root, rel);
    if ((rc = write_all(path, text, n)) != 0) return rc;
    if (ue) {
        const size_t cap = (size_t)c->nUE * PRACH_LOG_LINE_MAX + 1;
        char *buf = (char *)malloc(cap);
        if (!buf) return PRACH_ERR_IO;
        const size_t need = prach_format_logs(ue, c->nUE, buf, cap);
        prach_result_file_name(c, 1, rel, sizeof rel);
        snprintf(path, sizeof path, "%s/%s", root, rel);
        rc = write_all(path, buf, need);
        free(buf);
    }
    return rc;
}

/* ---- results.csv (AveragePerformance.py) ------------------------------------------------------ */

int prach_results_csv_accumulate(double acc[6], const char *txt) {
    if (!acc || !txt) return PRACH_ERR_ARG;
    const char *p = txt;
    for (int i = 0; i < 6; i++) { /* data.append(float(line.strip())) for the six lines (AveragePerformance.py:14-19) */
        char *end;
        double v = strtod(p, &end);
        if (end == p) return PRACH_ERR_ARG;
        acc[i] += v;
        p = end;
        while (*p == '\n' || *p == '\r' || *p == ' ') p++;
    }
    return PRACH_OK;
}

/* Python's repr(float): shortest digit string that round-trips, fixed notation for 1e-4 <= |x| < 1e16 */
static int py_float_repr(double x, char *out, size_t cap) {
    if (x == 0) return snprintf(out, cap, signbit(x) ? "-0.0" : "0.0");
    if (!isfinite(x)) return snprintf(out, cap, isnan(x) ? "nan" : (x > 0 ? "inf" : "-inf"));
    char e[40];
    int prec;
    for (prec = 1; prec <= 17; prec++) {
        snprintf(e, sizeof e, "%.*e", prec - 1, x);
        if (strtod(e, NULL) == x) break;
    }
    /* e = [-]d.ddddde[+-]XX */
    const char *m = e;
    int neg = 0;
    if (*m == '-') { neg = 1; m++; }
    char digits[24];
    int nd = 0;
    for (; *m && *m != 'e'; m++)
        if (*m >= '0' && *m <= '9') digits[nd++] = *m;
    while (nd > 1 && digits[nd - 1] == '0') nd--; /* shortest */
    int ex = atoi(m + 1);
    char body[64];
    int n = 0;
    if (ex < -4 || ex >= 16) {
        body[n++] = digits[0];
        if (nd > 1) { body[n++] = '.'; for (int i = 1; i < nd; i++) body[n++] = digits[i]; }
        n += snprintf(body + n, sizeof body - n, "e%c%02d", ex < 0 ? '-' : '+', ex < 0 ? -ex : ex);
    } else if (ex >= 0) {
        for (int i = 0; i <= ex; i++) body[n++] = i < nd ? digits[i] : '0';
        body[n++] = '.';
        if (nd > ex + 1) for (int i = ex + 1; i < nd; i++) body[n++] = digits[i];
        else body[n++] = '0';
    } else {
        body[n++] = '0'; body[n++] = '.';
        for (int i = 0; i < -ex - 1; i++) body[n++] = '0';
        for (int i = 0; i < nd; i++) body[n++] = digits[i];
    }
    body[n] = 0;
    return snprintf(out, cap, "%s%s", neg ? "-" : "", body);
}

size_t prach_results_csv_row(const double acc[6], int nseeds, char *buf, size_t cap) {
    char tmp[512];
    int n = 0;
    for (int i = 0; i < 6; i++) {
        double v = acc[i] / (double)nseeds;       /* lists/len(seedNumber) */
        v = rint(v * 1000.0) / 1000.0;             /* np.around(., 3): multiply, rint (half to even), divide */
        char f[80];
        py_float_repr(v, f, sizeof f);
        n += snprintf(tmp + n, sizeof tmp - n, "%s%s", i ? "," : "", f);
    }
    n += snprintf(tmp + n, sizeof tmp - n, "\r\n"); /* csv.writer default line terminator */
    if (buf && (size_t)n < cap) memcpy(buf, tmp, (size_t)n + 1);
    return (size_t)n;
}
