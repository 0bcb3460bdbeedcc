/* oracle/noma_oracle.c — TEST INFRASTRUCTURE (oracle).  Not part of the shipped library.
 *
 * CPU restatement of NOMA.c's simulation loop (NOMA.c:644-714): per 5 ms access slot, Beta(3,4)
 * arrivals, activeUE (preamble / sector / Rayleigh channel gain with two rejection loops,
 * NOMA.c:131-192), preambleSectorCollisionDetection (per-sector 54-bin histogram, singletons,
 * bubble sort by channel gain, greedy >15 "dB" pairing onto 2 grants per sector, NOMA.c:194-324),
 * msg2Results (NOMA.c:449-498); every ms resourceRequestAllocation (NOMA.c:499-546), timers,
 * success count.  Pinned against the real NOMA program's stdout (tests/golden/noma_c.json) in
 * glibc mode; philox mode (per-UE counters; a pair's two decode draws come from a per-(slot,
 * sector, grant) counter of the pseudo-UE 0xFFFFFFFF, so that every workgroup of a cluster can
 * evaluate them without exchanging state) is the mode the GPU implements, because the rejection
 * loops make the number of rand() calls data dependent.
 */
#define _GNU_SOURCE
#include "prach_oracle.h"
#include "glibc_rand.h"
#include "philox.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define betaF 0.0165

#include "oracle_internal.h"

typedef struct {
    int idx, timer, active, preamble, nTxPreamble, rarWindow, msg1ReTx, msg2, msg3Wait, msg3Faile, txTime, nowBackoff,
        firstTxTime, secondTxTime, RaFailed, RA, sector;
    float angle, xCoordinate, yCoordinate;
    double channelGain;
    uint32_t nd;
} nue_t;

typedef struct {
    const noma_cfg *k;
    oracle_rng *rng;
    nue_t *UE;
} nctx_t;

/* decode draws of a NOMA pair (NOMA.c:284-286): glibc: the global stream; philox: stateless per (slot, sector, grant) */
static int pair_draw(nctx_t *c, int slot, int sector, int grant, int which) {
    c->rng->consumed++;
    if (c->rng->mode == ORACLE_RNG_GLIBC) return glibc_rand(&c->rng->g);
    const uint32_t k = (uint32_t)(((slot * 6 + sector) * c->k->nGrantUL + grant) * 2 + which);
    return philox_draw31(c->rng->seed, (uint32_t)c->k->nUE, 2u, 0xFFFFFFFFu, k);
}

static int ndraw(nctx_t *c, nue_t *u) {
    c->rng->consumed++;
    if (c->rng->mode == ORACLE_RNG_GLIBC) return glibc_rand(&c->rng->g);
    return philox_draw31(c->rng->seed, (uint32_t)c->k->nUE, 2u /* NOMA_C */, (uint32_t)u->idx, u->nd++);
}

static float betaDist(float a, float b, float x) { /* NOMA.c:563-566 */
    float betaValue = (1 / betaF) * (pow(x, (a - 1))) * (pow((1 - x), (b - 1)));
    return betaValue;
}

static int n_align(int subTime, int accessTime) { /* NOMA.c:464-475 */
    if (subTime % accessTime == 0) return subTime + 1;
    if (subTime % accessTime == 1) return subTime;
    return subTime + (accessTime - (subTime % accessTime) + 1);
}

/* NOMA.c:131-192 */
static void activeUE(nctx_t *c, nue_t *user, int time) {
    const noma_cfg *k = c->k;
    const float pi = 3.14; /* NOMA.c:55 */
    const float cellRadius = k->cellRadius;
    user->active = 1;
    user->preamble = ndraw(c, user) % k->nPreamble;
    user->nTxPreamble++;
    user->txTime = time + 1;
    user->timer = 0;
    user->rarWindow = 0;
    user->msg1ReTx = 0;
    user->nowBackoff = 0;
    user->firstTxTime = time + 1;

    float angle = (float)ndraw(c, user) / (float)(2147483647) * 2 * pi;
    user->angle = angle;
    if (user->angle >= 0 && user->angle < ((1. / 3.) * pi)) user->sector = 0;
    else if (user->angle >= ((1. / 3.) * pi) && user->angle < ((2. / 3.) * pi)) user->sector = 1;
    else if (user->angle >= ((2. / 3.) * pi) && user->angle < 3.14) user->sector = 2;
    else if (user->angle >= pi && user->angle < ((4. / 3.) * pi)) user->sector = 3;
    else if (user->angle >= ((4. / 3.) * pi) && user->angle < ((5. / 3.) * pi)) user->sector = 4;
    else user->sector = 5;

    float r;
    while (1) {
        r = cellRadius * sqrt((float)ndraw(c, user) / (float)2147483647);
        if (r > 35.0) break;
    }
    float pathloss;
    user->xCoordinate = r * cos(angle);
    user->yCoordinate = r * sin(angle);
    double env = sqrt(user->xCoordinate * user->xCoordinate + user->yCoordinate * user->yCoordinate);
    double ch_g = 0, rayleigh;
    while (ch_g < 1e-7) {
        pathloss = sqrt(1 + pow(env, 2));
        rayleigh = sqrt(-2 * log((double)ndraw(c, user) / (double)2147483647));
        ch_g = pow(rayleigh / pathloss, 2);
    }
    user->channelGain = ch_g;
}

typedef struct { int idx; double channelGain; } tx_t;

/* NOMA.c:194-324 */
static void preambleSectorCollisionDetection(nctx_t *c, int activeCheck, int time, int *grantCheck, int *tmpIdx, int *cnt, int *who) {
    const noma_cfg *k = c->k;
    nue_t *user = c->UE;
    const int nP = k->nPreamble, nGrantUL = k->nGrantUL;
    tx_t txUEs[256];
    (void)tmpIdx;
    /* k->nonsector: the cell-wide variant preambleCollisionDetection (NOMA.c:325-447; its call at NOMA.c:688 is commented out in
     * the reference): ONE group and ONE grant budget; it differs from the per-sector function in exactly two more places, marked below */
    const int nonsector = k->nonsector != 0, nsect = nonsector ? 1 : 6;
    /* per (sector, preamble): count and the (only relevant) member when count == 1 */
    memset(cnt, 0, sizeof(int) * 6 * (size_t)nP);
    for (int i = 0; i < activeCheck; i++) {
        if (user[i].RA == 0 && user[i].txTime == time + 1 && user[i].msg2 == 0 && user[i].nowBackoff <= 0 && user[i].RaFailed == 0) {
            int b = (nonsector ? 0 : user[i].sector) * nP + user[i].preamble; /* (NOMA.c:332 also asks active == 1: implied by txTime == time + 1) */
            if (cnt[b]++ == 0) who[b] = i;
        }
    }
    for (int s = 0; s < nsect; s++) {
        int count = 0;
        for (int p = 0; p < nP; p++)
            if (cnt[s * nP + p] == 1) { txUEs[count].idx = who[s * nP + p]; txUEs[count].channelGain = user[who[s * nP + p]].channelGain; count++; }
        if (count <= 0) continue;
        if (count <= nGrantUL) {
            for (int i = 0; i < count; i++) {
                if (nonsector) { /* NOMA.c:377-382: msg2 = 1 sits OUTSIDE the budget test */
                    if (grantCheck[s] < nGrantUL) grantCheck[s]++;
                    user[txUEs[i].idx].msg2 = 1;
                } else if (grantCheck[s] < nGrantUL) { grantCheck[s]++; user[txUEs[i].idx].msg2 = 1; } /* NOMA.c:245-250 */
            }
        } else {
            for (int i = 0; i < count; i++) /* sortUE: bubble sort, strict < (stable), NOMA.c:90-103 */
                for (int j = 0; j < count - 1; j++)
                    if (txUEs[j + 1].channelGain < txUEs[j].channelGain) { tx_t t = txUEs[j]; txUEs[j] = txUEs[j + 1]; txUEs[j + 1] = t; }
            int pair = 0;
            for (int i = 0; i < count - 1; i++) {
                for (int j = 0 + 1; j < count; j++) {
                    int rx[2] = {txUEs[i].idx, txUEs[j].idx};
                    double low = txUEs[i].channelGain, high = txUEs[j].channelGain;
                    if (rx[0] != -1 && rx[1] != -1 && 10 * log(high) - 10 * log(low) > 15.) {
                        pair += 2;
                        txUEs[i].idx = -1;
                        txUEs[j].idx = -1;
                        if (grantCheck[s] < nGrantUL) {
                            const int gi = grantCheck[s];
                            grantCheck[s]++;
                            double p = (double)pair_draw(c, time / k->accessTime, s, gi, 0) / (double)2147483647;
                            if (p < 0.3) {
                                if (nonsector) user[rx[0]].msg2 = 1; /* NOMA.c:413-415: always the weaker UE, no second draw */
                                else {
                                    int randomUE = pair_draw(c, time / k->accessTime, s, gi, 1) % 2; /* NOMA.c:287-290 */
                                    user[rx[randomUE]].msg2 = 1;
                                }
                            } else {
                                user[rx[0]].msg2 = 1;
                                user[rx[1]].msg2 = 1;
                            }
                        }
                        break;
                    }
                }
            }
            if (count - pair > 0)
                for (int i = 0; i < count; i++)
                    if (txUEs[i].idx != -1 && grantCheck[s] < nGrantUL) { grantCheck[s]++; user[txUEs[i].idx].msg2 = 1; }
        }
    }
}

/* NOMA.c:449-498 */
static void msg2Results(nctx_t *c, nue_t *user, int time) {
    const noma_cfg *k = c->k;
    if (user->msg2 == 0 && user->active == 1) {
        user->rarWindow = 5;
        user->txTime += 3;
        if (user->rarWindow >= k->maxRarWindow) {
            user->nTxPreamble++;
            user->rarWindow = 0;
            user->msg1ReTx++;
            int tmp = ndraw(c, user) % k->backoff;
            user->txTime = n_align(user->txTime + tmp, k->accessTime);
            user->nowBackoff = user->txTime - time - 1;
            user->secondTxTime = user->txTime;
            if (user->msg1ReTx >= k->maxMsg1ReTx) {
                user->preamble = ndraw(c, user) % k->nPreamble;
                user->RaFailed++;
                user->nTxPreamble = 0;
                user->rarWindow = 0;
                user->msg1ReTx = 0;
                user->timer = 0;
            }
        }
    } else if (user->msg2 == 1) {
        user->active = 2;
        user->txTime += 10;
        user->secondTxTime = user->txTime;
        user->msg3Wait = 0;
    }
}

int noma_oracle_run_trial(const noma_cfg *k, oracle_rng *rng, noma_result *res, noma_ue *ue_out) {
    if (!k || !rng || !res || k->nUE <= 0 || k->nPreamble <= 0 || k->nPreamble > 256) return -1;
    const int nUE = k->nUE, accessTime = k->accessTime, maxTime = 10000;
    nctx_t c;
    c.k = k; c.rng = rng;
    c.UE = (nue_t *)calloc((size_t)nUE, sizeof(nue_t)); /* NOMA.c:651-655: everything 0, sector -1 */
    int *cnt = (int *)malloc(sizeof(int) * 6 * (size_t)k->nPreamble), *who = (int *)malloc(sizeof(int) * 6 * (size_t)k->nPreamble);
    if (!c.UE || !cnt || !who) return -2;
    for (int i = 0; i < nUE; i++) { c.UE[i].idx = i; c.UE[i].sector = -1; }
    const uint64_t draws0 = rng->consumed;
    int activeCheck = 0, sectorGrants[6], time, nSuccessUE = 0;
    uint64_t steps = 0;
    const int stop = (k->max_steps > 0 && k->max_steps < maxTime) ? k->max_steps : maxTime;
    for (time = 0; time < stop; time++) {
        steps++;
        nSuccessUE = 0;
        if (time % accessTime == 0) {
            for (int s = 0; s < 6; s++) sectorGrants[s] = 0;
            float numBetaDist = betaDist(3, 4, (float)time / (float)maxTime);
            int accessUEs = (int)ceil((float)nUE * numBetaDist / ((float)maxTime / (float)accessTime));
            activeCheck += accessUEs;
            if (activeCheck >= nUE) activeCheck = nUE;
            for (int i = 0; i < activeCheck; i++)
                if (c.UE[i].RA == 0 && c.UE[i].active == 0 && c.UE[i].RaFailed == 0) activeUE(&c, &c.UE[i], time);
            preambleSectorCollisionDetection(&c, activeCheck, time, sectorGrants, NULL, cnt, who);
            for (int i = 0; i < activeCheck; i++) {
                nue_t *u = &c.UE[i];
                if (u->nowBackoff <= 0 && u->txTime == time + 1 && u->active == 1 && u->RA == 0 && u->RaFailed == 0)
                    msg2Results(&c, u, time + 1);
            }
        }
        for (int i = 0; i < activeCheck; i++) { /* resourceRequestAllocation, NOMA.c:499-546 */
            nue_t *u = &c.UE[i];
            if (u->txTime == time && u->msg2 == 1 && u->active == 2 && u->RaFailed == 0) {
                if (u->msg3Wait <= 48) {
                    float p = (float)ndraw(&c, u) / (float)2147483647;
                    if (p > 0.1) { u->active = 0; u->RA = 1; u->timer = u->timer + 6; }
                    else { u->txTime += 49; u->msg3Wait = 49; }
                } else {
                    u->RA = 0; u->msg3Faile++; u->active = 1; u->msg2 = 0;
                    u->preamble = ndraw(&c, u) % k->nPreamble;
                    int tmp = ndraw(&c, u) % k->backoff;
                    u->txTime = n_align(u->txTime + tmp, accessTime);
                    u->secondTxTime = u->txTime;
                    u->nowBackoff = u->txTime - time - 1;
                    u->rarWindow = 0; u->nTxPreamble = 0; u->msg1ReTx = 0; u->timer = 0;
                }
            }
        }
        for (int i = 0; i < activeCheck; i++) { /* NOMA.c:702-706 */
            nue_t *u = &c.UE[i];
            if (u->active > 0 && u->RA == 0 && u->RaFailed == 0) { u->timer++; if (u->nowBackoff > 0) u->nowBackoff--; }
        }
        for (int i = 0; i < nUE; i++) if (c.UE[i].RA == 1) nSuccessUE++;
        if (nSuccessUE == nUE) break;
    }
    int delay = 0, nTxP = 0, failed = 0; /* saveResult, NOMA.c:618-625 */
    for (int i = 0; i < nUE; i++) {
        if (c.UE[i].RA == 1) { delay += c.UE[i].timer; nTxP += c.UE[i].nTxPreamble; }
        if (c.UE[i].RaFailed) failed++;
    }
    memset(res, 0, sizeof(*res));
    res->nSuccessUE = nSuccessUE; res->delay = delay; res->nTxP = nTxP; res->activeCheck = activeCheck;
    res->time_exit = time; res->raFailedUEs = failed; res->draws = rng->consumed - draws0; res->steps = steps;
    if (ue_out)
        for (int i = 0; i < nUE; i++) {
            const nue_t *u = &c.UE[i];
            noma_ue *o = &ue_out[i];
            o->idx = u->idx; o->timer = u->timer; o->active = u->active; o->txTime = u->txTime; o->firstTxTime = u->firstTxTime;
            o->secondTxTime = u->secondTxTime; o->nowBackoff = u->nowBackoff; o->preamble = u->preamble; o->sector = u->sector;
            o->rarWindow = u->rarWindow; o->msg1ReTx = u->msg1ReTx; o->nTxPreamble = u->nTxPreamble; o->msg2 = u->msg2;
            o->msg3Wait = u->msg3Wait; o->RA = u->RA; o->RaFailed = u->RaFailed | (u->msg3Faile << 16);
            o->channelGain = u->channelGain;
        }
    free(c.UE); free(cnt); free(who);
    return 0;
}

/* the line NOMA.c prints and appends to TestResults/Sector_{nUE}_Result.txt (NOMA.c:606-632) */
size_t noma_format_result_line(const noma_cfg *k, const noma_result *r, char *buf, size_t cap) {
    char tmp[256];
    int n = snprintf(tmp, sizeof tmp, "%d %d %lf %lf %lf\n", k->nUE, r->nSuccessUE, ((float)r->nSuccessUE / (float)k->nUE) * 100.0,
                     ((float)r->nTxP / (float)r->nSuccessUE), ((float)r->delay / (float)r->nSuccessUE));
    if (buf && (size_t)n < cap) memcpy(buf, tmp, (size_t)n + 1);
    return (size_t)n;
}
