/* oracle/prach_oracle.c — TEST INFRASTRUCTURE (oracle).  Not part of the shipped library.
 *
 * CPU restatement of the reference's hot path: the per-subframe loop of
 * RandomAccessSimulatorBeta.c:111-197 ("BETA_C") and RandomAccessWithNOMA.c:267-351
 * ("WITHNOMA_C"), UE by UE in index order with immediate side effects, exactly as the
 * reference executes it.  Two collision-count back ends:
 *   ORACLE_SCAN_LITERAL : the reference's own linear scan over all UEs per preambleCollision
 *                         call (Beta.c:321-330) — O(N^2) per subframe, used on small cases and as
 *                         the "port" CPU baseline;
 *   ORACLE_SCAN_SETS    : per-preamble matched sets maintained incrementally —
 *                         O(N) per subframe, same results (tests compare the two).
 * Parity is PINNED against the compiled reference (tests/test_oracle_golden.py).
 */
#define _GNU_SOURCE
#include "prach_oracle.h"
#include "glibc_rand.h"
#include "philox.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define betaF 0.0165 /* Beta.c:8 (note: not 1/B(3,4); kept as is) */

#include "oracle_internal.h"

oracle_rng *oracle_rng_new(int mode, uint64_t seed) {
    oracle_rng *r = (oracle_rng *)calloc(1, sizeof(*r));
    if (!r) return NULL;
    r->mode = mode;
    r->seed = seed;
    glibc_srand(&r->g, (unsigned int)seed); /* Beta.c:69 */
    return r;
}
void oracle_rng_free(oracle_rng *r) { free(r); }
uint64_t oracle_rng_consumed(const oracle_rng *r) { return r->consumed; }
int oracle_rng_next_glibc(oracle_rng *r) { r->consumed++; return glibc_rand(&r->g); }
int oracle_philox_draw31(uint64_t seed, uint32_t nUE, uint32_t variant, uint32_t ue, uint32_t k) {
    return philox_draw31(seed, nUE, variant, ue, k);
}
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr, key, out);
}

/* ---- per-trial context -------------------------------------------------------------------- */

typedef struct {
    oracle_ue u;
    int32_t raFailed;   /* Beta.c:24, never set to -1 (assignment commented out, Beta.c:252) */
    int32_t sector;     /* WithNOMA:398-410 (only read by the dormant per-sector grant test, WithNOMA:626-637) */
    uint32_t ndraw;     /* philox: per-UE draw index */
} ue_t;

typedef struct {
    const oracle_cfg *cfg;
    oracle_rng *rng;
    ue_t *UE;
    int nUE;
    /* matched sets: doubly linked list per preamble of UEs with active==1 && txTime==time */
    int *head, *cnt, *next, *prev, *inb;
    int collisionPreambles, totalPreambleTxop; /* Beta.c:41-42 */
    int continueFaliedUEs, finalSuccessUEs;    /* WithNOMA:84-85 */
    uint64_t collisionCalls;
} ctx_t;

/* Optional per-subframe census for design studies (tests/test_phase_model.py, oracle/README of the gate in profiles/r04_lcluster_phases.md): how many UEs (re)draw
 * their preamble in a subframe (Beta.c:231,256 / :398 — what would have to MIGRATE if workgroups owned preamble buckets instead of UE ranges), how many
 * preambleCollision calls are singletons (Beta.c:334 — the callers an index-ordered grant ranking has to bring together), how many calls there are. */
static int32_t *census_redraws, *census_singles, *census_calls;
static int census_n, census_t;
void oracle_set_census(int32_t *redraws, int32_t *singles, int32_t *calls, int n) { census_redraws = redraws; census_singles = singles; census_calls = calls; census_n = n; census_t = 0; }
#define CENSUS(arr) do { if (arr && census_t < census_n) arr[census_t]++; } while (0)

static int draw(ctx_t *c, ue_t *u) {
    c->rng->consumed++;
    if (c->rng->mode == ORACLE_RNG_GLIBC) return glibc_rand(&c->rng->g);
    return philox_draw31(c->rng->seed, (uint32_t)c->nUE, (uint32_t)c->cfg->variant, (uint32_t)u->u.idx,
                         u->ndraw++);
}

/* Beta.c:516-519 — note `1 - x` is evaluated in float before the promotion to double. */
static float beta_dist(float a, float b, float x) {
    float betaValue = (1 / betaF) * (pow(x, (a - 1))) * (pow((1 - x), (b - 1)));
    return betaValue;
}

static int max_time(const oracle_cfg *cfg) { return cfg->uniform ? 60000 : 10000; } /* Beta.c:92,103 */

int oracle_arrival_schedule(const oracle_cfg *cfg, int32_t *out, int cap, int32_t *nAccessUEo) {
    int n = cfg->nUE, nUE = cfg->nUE, accessTime = cfg->accessTime, maxTime = max_time(cfg);
    int nAccessUE = 0;
    if (cfg->uniform) { /* Beta.c:95-100 */
        nAccessUE = ceil((float)n * (float)accessTime * 1.0 / (float)maxTime);
        if (nAccessUE <= 0) nAccessUE = 1;
    }
    if (nAccessUEo) *nAccessUEo = nAccessUE;
    int activeCheck = 0, s = 0;
    for (int time = 0; time < maxTime; time += accessTime, s++) {
        if (activeCheck >= nUE) activeCheck = nUE;
        if (activeCheck != nUE) { /* Beta.c:121-134 */
            if (cfg->uniform) {
                activeCheck += nAccessUE;
            } else {
                float betaDist = beta_dist(3, 4, (float)time / (float)maxTime);
                int accessUEs = (int)ceil((float)nUE * betaDist / ((float)maxTime / (float)accessTime));
                activeCheck += accessUEs;
            }
            if (activeCheck >= nUE) activeCheck = nUE;
        }
        if (s < cap) out[s] = activeCheck;
    }
    return s;
}

/* ---- matched sets -------------------------------------------------------------------------- */
static inline int is_member(const ue_t *u, int time) { return u->u.active == 1 && u->u.txTime == time; }

static void set_add(ctx_t *c, int i) {
    int p = c->UE[i].u.preamble;
    c->inb[i] = p;
    c->prev[i] = -1;
    c->next[i] = c->head[p];
    if (c->head[p] >= 0) c->prev[c->head[p]] = i;
    c->head[p] = i;
    c->cnt[p]++;
}
static void set_del(ctx_t *c, int i) {
    int p = c->inb[i];
    if (p < 0) return;
    if (c->prev[i] >= 0) c->next[c->prev[i]] = c->next[i]; else c->head[p] = c->next[i];
    if (c->next[i] >= 0) c->prev[c->next[i]] = c->prev[i];
    c->inb[i] = -1;
    c->cnt[p]--;
}

/* ---- state-machine ops (one function per reference function) ------------------------------- */

/* slot alignment, Beta.c:268-277 / 294-303 / 390-399 */
static inline int slot_align(int subTime, int accessTime) {
    int m = subTime % accessTime;
    if (m == 0) return subTime + 1;
    if (m == 1) return subTime;
    return subTime + (accessTime - m + 1);
}

/* Beta.c:229-312, WithNOMA:475-562 */
static void selectPreamble(ctx_t *c, ue_t *user, int time) {
    const oracle_cfg *k = c->cfg;
    oracle_ue *u = &user->u;
    census_t = time;
    if (u->preamble == -1) {
        CENSUS(census_redraws);
        u->preamble = draw(c, user) % k->nPreamble;
        u->rarWindow = 0;
        u->maxRarCounter = 0;
        u->preambleChange = 1;
        u->preambleTxCounter = 1;
        u->nowBackoff = 0;
        if (k->variant == ORACLE_VARIANT_WITHNOMA_C) u->failCount = 0;
    } else if (u->nowBackoff <= 0) {
        u->rarWindow++;
        if (u->rarWindow >= k->maxRarWindow) {
            if (u->maxRarCounter >= k->maxMsg2TxCount) {
                if (k->variant == ORACLE_VARIANT_WITHNOMA_C) c->continueFaliedUEs++; /* WithNOMA:499 */
                CENSUS(census_redraws);
                u->preamble = draw(c, user) % k->nPreamble;
                u->rarWindow = 0;
                u->maxRarCounter = 0;
                u->preambleChange = 1;
                u->preambleTxCounter = 1;
                u->nowBackoff = 0;
                u->timer = 0;
                u->firstTxTime = time + 1;
                if (k->variant == ORACLE_VARIANT_WITHNOMA_C) u->failCount += 1; /* WithNOMA:512 */
                int tmp = draw(c, user) % k->backoff;
                int subTime = u->txTime + tmp; /* stale txTime, Beta.c:266 */
                u->txTime = slot_align(subTime, k->accessTime);
                u->nowBackoff = u->txTime - time;
            } else {
                u->rarWindow = 0;
                u->maxRarCounter++;
                u->preambleTxCounter++;
                int tmp = draw(c, user) % k->backoff;
                int subTime = time + tmp;
                u->txTime = slot_align(subTime, k->accessTime);
                u->nowBackoff = u->txTime - time;
                u->secondTxTime = u->txTime;
            }
        }
    }
}

/* Beta.c:315-369, WithNOMA:607-665.  `self` has been taken out of the matched sets by the caller. */
static void preambleCollision(ctx_t *c, int self, int time, int *grantCheck) {
    const oracle_cfg *k = c->cfg;
    ue_t *user = &c->UE[self];
    int p = user->u.preamble;
    int check;
    c->collisionCalls++;
    if (k->scan_mode == ORACLE_SCAN_LITERAL) {
        check = 0;
        for (int i = 0; i < c->nUE; i++)
            if (c->UE[i].u.active == 1 && c->UE[i].u.txTime == time && c->UE[i].u.preamble == p) check++;
    } else {
        check = c->cnt[p] + 1;
    }
    census_t = time;
    CENSUS(census_calls);
    if (check == 1) {
        CENSUS(census_singles);
        c->totalPreambleTxop++;
        *grantCheck = *grantCheck + 1;
        if (*grantCheck < k->nGrantUL) {
            user->u.active = 2;
            user->u.txTime = time + 11;
            user->u.connectionRequest = 0;
            user->u.msg2Flag = 1;
        } else {
            user->u.txTime++;
        }
    } else {
        if (k->variant == ORACLE_VARIANT_WITHNOMA_C) { /* WithNOMA:650-652 */
            c->collisionPreambles += check;
            c->totalPreambleTxop += check;
        } else { /* Beta.c:349-351 */
            c->collisionPreambles++;
            c->totalPreambleTxop++;
        }
        if (k->scan_mode == ORACLE_SCAN_LITERAL) {
            for (int i = 0; i < c->nUE; i++)
                if (c->UE[i].u.active == 1 && c->UE[i].u.txTime == time && c->UE[i].u.preamble == p)
                    c->UE[i].u.txTime++;
        } else {
            user->u.txTime++;
            for (int j = c->head[p]; j >= 0;) {
                int nx = c->next[j];
                c->UE[j].u.txTime++;
                c->inb[j] = -1;
                j = nx;
            }
            c->head[p] = -1;
            c->cnt[p] = 0;
        }
    }
}

/* Beta.c:371-411, WithNOMA:667-710 */
static void requestResourceAllocation(ctx_t *c, ue_t *user, int time) {
    const oracle_cfg *k = c->cfg;
    oracle_ue *u = &user->u;
    u->connectionRequest++;
    if (u->connectionRequest < 48) {
        float p = (float)draw(c, user) / (float)2147483647 /* RAND_MAX */;
        if (p > 0.1) {
            u->msg4Flag = 1;
            u->timer = u->timer + 6;
            u->active = 0;
            c->finalSuccessUEs++;
        } else {
            u->connectionRequest = 48;
            u->txTime += 48;
        }
    } else {
        c->continueFaliedUEs++;
        int tmp = draw(c, user) % k->backoff;
        int subTime = u->txTime + tmp;
        u->txTime = slot_align(subTime, 5); /* hard-coded accessTime = 5, Beta.c:389 */
        u->active = 1;
        u->nowBackoff = u->txTime - time;
        census_t = time;
        CENSUS(census_redraws);
        u->preamble = draw(c, user) % k->nPreamble;
        u->timer = 0;
        u->msg2Flag = 0;
        u->rarWindow = 0;
        u->maxRarCounter = 0;
        u->connectionRequest = 0;
        if (k->variant == ORACLE_VARIANT_WITHNOMA_C) u->failCount += 1; /* WithNOMA:708 */
    }
}

int oracle_run_trial(const oracle_cfg *cfg, oracle_rng *rng, oracle_result *res, oracle_ue *ue_out) {
    if (!cfg || !rng || !res || cfg->nUE <= 0 || cfg->nPreamble <= 0 || cfg->backoff <= 0 || cfg->accessTime <= 0)
        return -1;
    const int nUE = cfg->nUE, accessTime = cfg->accessTime;
    const int withnoma = cfg->variant == ORACLE_VARIANT_WITHNOMA_C;
    ctx_t c;
    memset(&c, 0, sizeof(c));
    c.cfg = cfg;
    c.rng = rng;
    c.nUE = nUE;
    c.UE = (ue_t *)calloc((size_t)nUE, sizeof(ue_t)); /* Beta.c:78 */
    c.head = (int *)malloc(sizeof(int) * (size_t)cfg->nPreamble);
    c.cnt = (int *)calloc((size_t)cfg->nPreamble, sizeof(int));
    c.next = (int *)malloc(sizeof(int) * (size_t)nUE);
    c.prev = (int *)malloc(sizeof(int) * (size_t)nUE);
    c.inb = (int *)malloc(sizeof(int) * (size_t)nUE);
    if (!c.UE || !c.head || !c.cnt || !c.next || !c.prev || !c.inb) return -2;
    for (int i = 0; i < nUE; i++) { /* initialUE, Beta.c:220-227 */
        c.UE[i].u.idx = i;
        c.UE[i].u.timer = -1;
        c.UE[i].u.active = -1;
        c.UE[i].u.txTime = -1;
        c.UE[i].u.preamble = -1;
        c.inb[i] = -1;
    }
    for (int p = 0; p < cfg->nPreamble; p++) c.head[p] = -1;

    const uint64_t draws0 = rng->consumed;
    const int maxTime = max_time(cfg);
    int nAccessUE = 0;
    if (cfg->uniform) {
        nAccessUE = ceil((float)nUE * (float)accessTime * 1.0 / (float)maxTime);
        if (nAccessUE <= 0) nAccessUE = 1;
    }
    int nSuccessUE = 0, activeCheck = 0, grantCheck = 0, time;
    int sectorGrants[6] = {0, 0, 0, 0, 0, 0}; /* WithNOMA:260 */
    const int per_sector = withnoma && cfg->sector_grants;
    uint64_t steps = 0;
    const int stop = (cfg->max_steps > 0 && cfg->max_steps < maxTime) ? cfg->max_steps : maxTime;

    for (time = 0; time < stop; time++) {
        steps++;
        if (time % 5 == 0) { /* Beta.c:112 — hard-coded 5; WithNOMA:268-274 */
            grantCheck = 0;
            for (int s = 0; s < 6; s++) sectorGrants[s] = 0;
        }
        if (activeCheck >= nUE) activeCheck = nUE;
        if (time % accessTime == 0 && activeCheck != nUE) { /* Beta.c:121-147 */
            if (cfg->uniform) {
                activeCheck += nAccessUE;
            } else {
                float betaDist = beta_dist(3, 4, (float)time / (float)maxTime);
                int accessUEs = (int)ceil((float)nUE * betaDist / ((float)maxTime / (float)accessTime));
                activeCheck += accessUEs;
            }
            if (activeCheck >= nUE) activeCheck = nUE;
            for (int i = 0; i < activeCheck; i++) {
                oracle_ue *u = &c.UE[i].u;
                if (u->active == -1) {
                    u->active = 1;
                    u->txTime = time + 1;
                    u->timer = 0;
                    u->msg2Flag = 0;
                    u->firstTxTime = time + 1;
                    if (withnoma) { /* activateUEs draws theta and r: WithNOMA:393-394 (r is never read; theta only fixes the sector) */
                        const float pi = 3.14;
                        float theta = (float)draw(&c, &c.UE[i]) / (float)(2147483647) * 2 * pi;
                        (void)draw(&c, &c.UE[i]);
                        int sec; /* WithNOMA:398-410 */
                        if (theta >= 0 && theta < ((1. / 3.) * pi)) sec = 0;
                        else if (theta >= ((1. / 3.) * pi) && theta < ((2. / 3.) * pi)) sec = 1;
                        else if (theta >= ((2. / 3.) * pi) && theta < 3.14) sec = 2;
                        else if (theta >= pi && theta < ((4. / 3.) * pi)) sec = 3;
                        else if (theta >= ((4. / 3.) * pi) && theta < ((5. / 3.) * pi)) sec = 4;
                        else sec = 5;
                        c.UE[i].sector = sec;
                    }
                }
            }
        }
        if (cfg->scan_mode == ORACLE_SCAN_SETS) {
            /* matched sets as of the start of the UE loop */
            for (int p = 0; p < cfg->nPreamble; p++) { c.head[p] = -1; c.cnt[p] = 0; }
            for (int i = 0; i < activeCheck; i++) {
                c.inb[i] = -1;
                if (is_member(&c.UE[i], time)) set_add(&c, i);
            }
        }
        /* Beta.c:150 loops to nUE, WithNOMA:302 to activeCheck: same thing, UEs >= activeCheck are idle */
        for (int i = 0; i < activeCheck; i++) {
            ue_t *user = &c.UE[i];
            oracle_ue *u = &user->u;
            if (u->msg4Flag == 0 && user->raFailed != -1) {
                if (cfg->scan_mode == ORACLE_SCAN_SETS) set_del(&c, i);
                if (withnoma) { /* WithNOMA:307-315 */
                    if (u->active == 1 && u->msg2Flag == 0) {
                        selectPreamble(&c, user, time);
                        /* WithNOMA:312 (the author's commented-out call) passes sectorGrants, WithNOMA:626-637 indexes it by the caller's sector */
                        if (u->txTime == time) preambleCollision(&c, i, time, per_sector ? &sectorGrants[user->sector] : &grantCheck);
                    }
                } else { /* Beta.c:155-164 */
                    if (u->active == 1 && u->msg2Flag == 0) selectPreamble(&c, user, time);
                    if (u->active == 1 && u->msg2Flag == 0 && u->txTime == time && u->nowBackoff <= 0)
                        preambleCollision(&c, i, time, &grantCheck);
                }
                if (u->txTime == time && u->active == 2) requestResourceAllocation(&c, user, time);
                if (u->active > 0) { /* timerIncrease, Beta.c:413-419 */
                    u->timer++;
                    if (u->nowBackoff > 0) u->nowBackoff--;
                }
                if (cfg->scan_mode == ORACLE_SCAN_SETS && is_member(user, time)) set_add(&c, i);
            }
        }
        nSuccessUE = 0; /* successUEs, Beta.c:421-429 */
        for (int i = 0; i < nUE; i++)
            if (c.UE[i].u.msg4Flag == 1) nSuccessUE++;
        if (nSuccessUE == nUE) break;
    }

    float totalDelay = 0; /* Beta.c:185-197 */
    int preambleTxCount = 0, failCounts = 0;
    int64_t sumTimer = 0;
    for (int i = 0; i < nUE; i++) {
        if (c.UE[i].u.msg4Flag == 1) {
            totalDelay += (float)c.UE[i].u.timer;
            sumTimer += c.UE[i].u.timer;
            preambleTxCount += c.UE[i].u.preambleTxCounter;
            failCounts += c.UE[i].u.failCount;
        }
    }
    memset(res, 0, sizeof(*res));
    res->time_exit = time;
    res->maxTime = maxTime;
    res->nSuccessUE = nSuccessUE;
    res->failedUEs = nUE - nSuccessUE;
    res->preambleTxCount = preambleTxCount;
    res->failCounts = failCounts;
    res->collisionPreambles = c.collisionPreambles;
    res->totalPreambleTxop = c.totalPreambleTxop;
    res->activeCheck = activeCheck;
    res->nAccessUE = nAccessUE;
    res->continueFaliedUEs = c.continueFaliedUEs;
    res->finalSuccessUEs = c.finalSuccessUEs;
    res->totalDelay = totalDelay;
    res->sumTimer = sumTimer;
    res->draws = rng->consumed - draws0;
    res->steps = steps;
    res->collisionCalls = c.collisionCalls;
    if (ue_out)
        for (int i = 0; i < nUE; i++) ue_out[i] = c.UE[i].u;
    free(c.UE); free(c.head); free(c.cnt); free(c.next); free(c.prev); free(c.inb);
    return 0;
}

/* ---- text surfaces -------------------------------------------------------------------------- */

size_t oracle_format_logs(const oracle_ue *ue, int nUE, char *buf, size_t cap) {
    size_t off = 0;
    char line[512];
    for (int i = 0; i < nUE; i++) {
        const oracle_ue *u = ue + i;
        /* format string of Beta.c:501 / WithNOMA:812 */
        int n = snprintf(line, sizeof line,
                         "Idx: %d | Timer: %d | Active: %d | txTime: %d | FirstTxTime: %d | SecondTxTime: %d | NowBackoff: %d | Preamble: %d | Preamble change: %d | RAR window: %d | Max RAR: %d | Preamble reTx: %d | MSG 2 Flag: %d | ConnectRequest: %d | MSG 4 Flag: %d\n",
                         u->idx, u->timer, u->active, u->txTime, u->firstTxTime, u->secondTxTime, u->nowBackoff,
                         u->preamble, u->preambleChange, u->rarWindow, u->maxRarCounter, u->preambleTxCounter,
                         u->msg2Flag, u->connectionRequest, u->msg4Flag);
        if (buf && off + (size_t)n <= cap) memcpy(buf + off, line, (size_t)n);
        off += (size_t)n;
    }
    return off;
}

typedef struct { float ratioSuccess, nCollisionPreambles, averagePreambleTx, averageDelay; } derived_t;

static derived_t derive(const oracle_cfg *cfg, const oracle_result *r) { /* Beta.c:434-438 */
    derived_t d;
    d.ratioSuccess = (float)r->nSuccessUE / (float)cfg->nUE * 100.0;
    d.nCollisionPreambles = (float)r->collisionPreambles / ((float)cfg->nUE * (float)cfg->nPreamble);
    d.averagePreambleTx = (float)r->preambleTxCount / (float)r->nSuccessUE;
    d.averageDelay = r->totalDelay / (float)r->nSuccessUE;
    return d;
}

size_t oracle_format_results(const oracle_cfg *cfg, const oracle_result *r, char *buf, size_t cap) {
    derived_t d = derive(cfg, r);
    char tmp[1024];
    int n;
    if (cfg->variant == ORACLE_VARIANT_WITHNOMA_C) /* WithNOMA:762-793 */
        n = snprintf(tmp, sizeof tmp,
                     "%d\n%.2lf\n%d\n%.2lf\n%.2lf\nNumber of total preamble tx: %d\nFinally Falied: %d\nFinally Success: %lf\n",
                     cfg->nUE, d.ratioSuccess, r->nSuccessUE, d.averagePreambleTx, d.averageDelay, r->preambleTxCount,
                     r->continueFaliedUEs,
                     (float)r->finalSuccessUEs / (float)(r->continueFaliedUEs + r->finalSuccessUEs));
    else /* Beta.c:460-479; the 6th line (latency, no newline) is wall-clock and appended by the caller */
        n = snprintf(tmp, sizeof tmp, "%d\n%.2lf\n%d\n%.2lf\n%.2lf\n", cfg->nUE, d.ratioSuccess, r->nSuccessUE,
                     d.averagePreambleTx, d.averageDelay);
    if (buf && (size_t)n < cap) memcpy(buf, tmp, (size_t)n + 1);
    return (size_t)n;
}

size_t oracle_format_stdout(const oracle_cfg *cfg, const oracle_result *r, char *buf, size_t cap) {
    derived_t d = derive(cfg, r);
    char tmp[2048];
    int n = snprintf(tmp, sizeof tmp, "-------- %05d Result ---------\n", r->activeCheck); /* Beta.c:200 */
    if (cfg->uniform) n += snprintf(tmp + n, sizeof tmp - n, "Number of RA try UEs per Subframe: %d\n", r->nAccessUE);
    if (cfg->variant == ORACLE_VARIANT_WITHNOMA_C)
        n += snprintf(tmp + n, sizeof tmp - n, "Fail Counts: %d\n", r->failCounts); /* WithNOMA:361 */
    /* (Beta.c prints "Latency: %lf" here, Beta.c:206 — wall clock, not reproduced by the oracle) */
    n += snprintf(tmp + n, sizeof tmp - n, "Number of UEs: %d\nTotal simulation time: %dms\nSuccess ratio: %.2lf\nNumber of succeed UEs: %d\n",
                  cfg->nUE, r->time_exit, d.ratioSuccess, r->nSuccessUE);
    if (cfg->variant == ORACLE_VARIANT_WITHNOMA_C)
        n += snprintf(tmp + n, sizeof tmp - n, "Number of falied UEs: %d\n", r->continueFaliedUEs); /* WithNOMA:745 */
    n += snprintf(tmp + n, sizeof tmp - n, "Number of collision preambles: %.6lf\nAverage preamble tx count: %.2lf\nAverage delay: %.2lf\n",
                  d.nCollisionPreambles, d.averagePreambleTx, d.averageDelay);
    if (buf && (size_t)n < cap) memcpy(buf, tmp, (size_t)n + 1);
    return (size_t)n;
}
