/* oracle/phase_model.c — TEST INFRASTRUCTURE.  Not part of the shipped library.
 *
 * CPU model of the PARALLEL-EXACT decomposition of one subframe that the HIP kernels implement
 * (DESIGN.md §3): per-UE own-state transitions in any order, a per-preamble-bucket resolution of
 * the reference's index-ordered side effects (RandomAccessSimulatorBeta.c:150-177, the
 * preambleCollision scan at :315-369), and a deferred per-UE "apply".  It exists to prove on the
 * CPU — against the sequential restatement in prach_oracle.c, which is pinned to the compiled
 * reference — that the decomposition is exact, including the rare paths (late joiners, reset
 * cycles that re-join, passive members, Msg3-timeout re-entries).  The UE ranges are processed in
 * REVERSE order on purpose: nothing in the per-UE phases may depend on processing order.
 *
 * Shares no code with the product; the product's kernels (csrc/prach_kernels.hip) are written to
 * the same decomposition and are checked on the GPU against prach_oracle.c directly.
 */
#define _GNU_SOURCE
#include "prach_oracle.h"
#include "glibc_rand.h"
#include "philox.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { PEND_NONE = 0, PEND_STAY, PEND_CALLER, PEND_RESET, PEND_PASSIVE, PEND_RJOIN };
enum { EV_CALLER = 1, EV_RESETCAND, EV_PASSIVE, EV_RJOIN };

typedef struct {
    int active, txTime, tb, bo, conn, preamble, rar, mrc, pend;
    int ptc, ftt, stt, fc;
    uint32_t nd;
} mue_t;

typedef struct { int idx, type, p, q, le, is_pre; } ev_t;

typedef struct {
    const oracle_cfg *cfg;
    int rng_mode;
    uint64_t seed;
    const int32_t *stream; /* glibc mode: pre-generated draw stream */
    uint64_t stream_len, base; /* base = draws consumed so far */
    int overflow;
} mrng_t;

static int m_align(int subTime, int aT) {
    int m = subTime % aT;
    if (m == 0) return subTime + 1;
    if (m == 1) return subTime;
    return subTime + (aT - m + 1);
}
static int m_enc_bo(int X, int t) { return X > 0 ? t + X : X; }
static int m_now_bo(int bo, int t) { return bo > 0 ? (bo - t > 0 ? bo - t : 0) : bo; }

/* number of draws UE makes in the UE loop of step t, from its pre-select state */
static int m_draw_count(const oracle_cfg *k, const mue_t *u, int t) {
    if (u->active == 1) {
        if (u->preamble == -1) return 1;
        if (m_now_bo(u->bo, t) > 0) return 0;
        if (u->rar + 1 >= k->maxRarWindow) return u->mrc >= k->maxMsg2TxCount ? 2 : 1;
        return 0;
    }
    if (u->active == 2 && u->txTime == t) return u->conn == 0 ? 1 : 2;
    return 0;
}

static int m_draw(mrng_t *r, mue_t *u, int idx, uint64_t *off) {
    if (r->rng_mode == ORACLE_RNG_GLIBC) {
        uint64_t o = (*off)++;
        if (o >= r->stream_len) { r->overflow = 1; return 0; }
        return r->stream[o];
    }
    return philox_draw31(r->seed, (uint32_t)r->cfg->nUE, (uint32_t)r->cfg->variant, (uint32_t)idx, u->nd++);
}

int model_run_trial(const oracle_cfg *cfg, int rng_mode, uint64_t seed, const int32_t *stream, uint64_t stream_len,
                    uint64_t stream_off, int nranges, oracle_result *res, oracle_ue *ue_out) {
    const int nUE = cfg->nUE, nP = cfg->nPreamble, aT = cfg->accessTime;
    const int withnoma = cfg->variant == ORACLE_VARIANT_WITHNOMA_C;
    const int maxTime = cfg->uniform ? 60000 : 10000;
    if (nranges < 1) nranges = 1;
    mue_t *U = (mue_t *)calloc((size_t)nUE, sizeof(mue_t));
    ev_t *EV = (ev_t *)malloc(sizeof(ev_t) * (size_t)nUE);
    int *Eidx = (int *)malloc(sizeof(int) * (size_t)nUE), *Ep = (int *)malloc(sizeof(int) * (size_t)nUE), nE = 0; /* 'early leaver' pre-members (cluster-kernel rank formulation) */
    int formula_mismatch = 0;
    int *evcnt = (int *)calloc((size_t)nranges, sizeof(int));
    int *evstart = (int *)calloc((size_t)nranges + 1, sizeof(int));
    int *rhist = (int *)calloc((size_t)nranges * nP, sizeof(int));
    int *rsm_idx = (int *)malloc(sizeof(int) * (size_t)nranges * nP);
    int *rsm_le = (int *)malloc(sizeof(int) * (size_t)nranges * nP);
    uint64_t *rdraws = (uint64_t *)calloc((size_t)nranges + 1, sizeof(uint64_t));
    int *total = (int *)malloc(sizeof(int) * nP), *fcall = (int *)malloc(sizeof(int) * nP),
        *lcall = (int *)malloc(sizeof(int) * nP), *sm_idx = (int *)malloc(sizeof(int) * nP),
        *sm_le = (int *)malloc(sizeof(int) * nP);
    int *granted = (int *)malloc(sizeof(int) * (size_t)(cfg->nGrantUL + 1));
    int ngranted = 0;
    int32_t *sched = (int32_t *)malloc(sizeof(int32_t) * (size_t)(maxTime / aT + 2));
    int32_t nAccessUE = 0;
    oracle_arrival_schedule(cfg, sched, maxTime / aT + 2, &nAccessUE);
    for (int i = 0; i < nUE; i++) { U[i].active = -1; U[i].txTime = -1; U[i].preamble = -1; U[i].tb = 0; }
    for (int p = 0; p < nP; p++) { fcall[p] = INT_MAX; lcall[p] = -1; }

    mrng_t R = {cfg, rng_mode, seed, stream, stream_len, stream_off, 0};
    int activeCheck = 0, grantCheck = 0, nSuccess = 0, time, tlast = -1;
    int collisionPreambles = 0, totalPreambleTxop = 0, continueFailed = 0, finalSuccess = 0;
    uint64_t steps = 0, calls = 0, philox_draws = 0;
    const int stop = (cfg->max_steps > 0 && cfg->max_steps < maxTime) ? cfg->max_steps : maxTime;

    for (time = 0; time < stop; time++) {
        const int t = time;
        steps++;
        tlast = t;
        if (t % 5 == 0) grantCheck = 0;
        int prevAC = activeCheck;
        if (t % aT == 0 && activeCheck != nUE) activeCheck = sched[t / aT];
        const int nAct = activeCheck - prevAC;
        /* ranges of [0, activeCheck): contiguous, multiples of 64 like the kernel's wave ranges */
        int groups = (activeCheck + 63) / 64, gper = (groups + nranges - 1) / nranges;
        memset(rhist, 0, sizeof(int) * (size_t)nranges * nP);
        nE = 0;
        for (int k = 0; k < nranges * nP; k++) { rsm_idx[k] = INT_MAX; rsm_le[k] = 0; }

        /* ---- pass 1 (glibc only): apply(t-1) + activation + per-range draw counts ---- */
        /* ---- pass 2: select/msg3 + classification.  Fused in philox mode; here always split, ranges reversed */
        for (int pass = 0; pass < 2; pass++) {
            for (int rr = nranges - 1; rr >= 0; rr--) {
                int lo = rr * gper * 64, hi = (rr + 1) * gper * 64;
                if (hi > activeCheck) hi = activeCheck;
                if (pass == 0) rdraws[rr] = 0;
                uint64_t off = 0;
                if (pass == 1) { off = R.base + (withnoma ? 2ull * (uint64_t)nAct : 0); for (int k = 0; k < rr; k++) off += rdraws[k]; evcnt[rr] = 0; evstart[rr] = lo < activeCheck ? lo : activeCheck; }
                int *run = rhist + (size_t)rr * nP; /* running per-bucket count of pre-members in this range */
                for (int i = lo; i < hi; i++) {
                    mue_t *u = &U[i];
                    if (pass == 0) {
                        /* apply(t-1) */
                        const int tp = t - 1;
                        if (u->pend != PEND_NONE) {
                            int isg = 0;
                            for (int g = 0; g < ngranted; g++) isg |= granted[g] == i;
                            switch (u->pend) {
                            case PEND_STAY: case PEND_CALLER:
                                if (isg) { u->active = 2; u->txTime = tp + 11; u->conn = 0; } else u->txTime = tp + 1;
                                break;
                            case PEND_RESET: {
                                int q = u->bo, tmp = u->txTime, bumped = fcall[q] < i;
                                int tx = m_align(tp + bumped + tmp, aT);
                                if (tx == tp) {
                                    u->bo = 0;
                                    if (isg) { u->active = 2; u->txTime = tp + 11; u->conn = 0; } else u->txTime = tp + 1;
                                } else { u->txTime = tx; u->bo = m_enc_bo(tx - tp, tp); }
                            } break;
                            case PEND_PASSIVE: if (fcall[u->preamble] != INT_MAX) u->txTime = tp + 1; break;
                            case PEND_RJOIN: if (lcall[u->preamble] > i) u->txTime = tp + 1; break;
                            }
                            u->pend = PEND_NONE;
                        }
                        if (i >= prevAC) { /* activation, Beta.c:136-146 */
                            u->active = 1; u->txTime = t + 1; u->tb = t; u->ftt = t + 1;
                            if (withnoma) { u->nd += 2; philox_draws += 2; }
                        }
                        rdraws[rr] += (uint64_t)m_draw_count(cfg, u, t);
                        continue;
                    }
                    /* pass 1: own-state transition of step t */
                    int evtype = 0, evp = -1, evq = -1;
                    const int oldp = u->preamble;
                    const int member_pre = u->active == 1 && u->txTime == t && oldp >= 0;
                    uint64_t off0 = off;
                    const int dc_pred = m_draw_count(cfg, u, t);
                    if (u->active == 1) {
                        int nb = m_now_bo(u->bo, t);
                        if (u->preamble == -1) {
                            u->preamble = m_draw(&R, u, i, &off) % nP;
                            u->rar = 0; u->mrc = 0; u->ptc = 1; u->bo = 0;
                            if (withnoma) u->fc = 0;
                            if (u->txTime == t) { u->pend = PEND_CALLER; evtype = EV_CALLER; evp = u->preamble; }
                        } else if (nb > 0) {
                            if (member_pre) {
                                if (withnoma) u->pend = PEND_STAY;
                                else { u->pend = PEND_PASSIVE; evtype = EV_PASSIVE; evp = oldp; }
                            }
                        } else {
                            u->rar++;
                            if (u->rar >= cfg->maxRarWindow) {
                                if (u->mrc >= cfg->maxMsg2TxCount) {
                                    if (withnoma) { continueFailed++; u->fc++; }
                                    int newp = m_draw(&R, u, i, &off) % nP;
                                    u->rar = 0; u->mrc = 0; u->ptc = 1; u->tb = t; u->ftt = t + 1;
                                    int tmp = m_draw(&R, u, i, &off) % cfg->backoff;
                                    u->preamble = newp;
                                    if (member_pre) {
                                        u->pend = PEND_RESET; u->txTime = tmp; u->bo = oldp;
                                        if (tmp == 0 && aT > 1 && t % aT == 1) { evtype = EV_RESETCAND; evp = newp; evq = oldp; }
                                    } else {
                                        u->txTime = m_align(u->txTime + tmp, aT);
                                        u->bo = m_enc_bo(u->txTime - t, t);
                                        if (u->txTime == t) { u->pend = PEND_CALLER; evtype = EV_CALLER; evp = newp; }
                                    }
                                } else {
                                    u->rar = 0; u->mrc++; u->ptc++;
                                    int tmp = m_draw(&R, u, i, &off) % cfg->backoff;
                                    u->txTime = m_align(t + tmp, aT);
                                    u->bo = m_enc_bo(u->txTime - t, t);
                                    u->stt = u->txTime;
                                    if (u->txTime == t) { u->pend = PEND_CALLER; evtype = EV_CALLER; evp = oldp; }
                                }
                            } else if (member_pre) u->pend = PEND_STAY;
                        }
                    } else if (u->active == 2 && u->txTime == t) {
                        if (u->conn == 0) {
                            u->conn = 1;
                            float pf = (float)m_draw(&R, u, i, &off) / (float)2147483647;
                            if (pf > 0.1) { u->active = 0; u->tb = (t - u->tb) + 6; /* final timer */ nSuccess++; finalSuccess++; }
                            else { u->conn = 2; u->txTime += 48; }
                        } else {
                            continueFailed++;
                            int tmp = m_draw(&R, u, i, &off) % cfg->backoff;
                            u->txTime = m_align(u->txTime + tmp, 5);
                            u->active = 1;
                            u->bo = m_enc_bo(u->txTime - t, t);
                            u->preamble = m_draw(&R, u, i, &off) % nP;
                            u->tb = t; u->rar = 0; u->mrc = 0; u->conn = 0;
                            if (withnoma) u->fc++;
                            if (u->txTime == t) { u->pend = PEND_RJOIN; evtype = EV_RJOIN; evp = u->preamble; }
                        }
                    }
                    if (R.rng_mode == ORACLE_RNG_GLIBC && (int)(off - off0) != dc_pred) R.overflow = 2; /* draw-count prediction must hold */
                    /* E class: a pre-member that leaves its bucket at its own turn without calling on it
                     * (retransmit rescheduled elsewhere, or any reset cycle) */
                    if (member_pre && ((u->pend == PEND_NONE && u->txTime != t) || u->pend == PEND_RESET)) { Eidx[nE] = i; Ep[nE] = oldp; nE++; }
                    if (member_pre) run[oldp]++;
                    /* STAYMIN candidate: first PEND_STAY of its bucket in this range */
                    if (u->pend == PEND_STAY && rsm_idx[(size_t)rr * nP + oldp] == INT_MAX) {
                        rsm_idx[(size_t)rr * nP + oldp] = i;
                        rsm_le[(size_t)rr * nP + oldp] = run[oldp];
                    }
                    if (evtype) {
                        ev_t *e = &EV[evstart[rr] + evcnt[rr]++];
                        e->idx = i; e->type = evtype; e->p = evp; e->q = evq; e->is_pre = member_pre && oldp == evp;
                        e->le = run[evp]; /* local #pre-members of bucket evp with idx <= i (this range) */
                    }
                }
            }
            if (pass == 0) { /* nothing */ }
        }
        { uint64_t tot = withnoma ? 2ull * (uint64_t)nAct : 0; for (int k = 0; k < nranges; k++) tot += rdraws[k]; R.base += tot; }

        /* ---- resolve (parallel-style: every loop below is over independent items unless noted) ---- */
        for (int p = 0; p < nP; p++) { total[p] = 0; sm_idx[p] = INT_MAX; sm_le[p] = 0; fcall[p] = INT_MAX; lcall[p] = -1; }
        /* prefix over ranges: make every local `le` global, pick global STAYMIN (first range that has one) */
        for (int p = 0; p < nP; p++) {
            int acc = 0;
            for (int rr = 0; rr < nranges; rr++) {
                if (sm_idx[p] == INT_MAX && rsm_idx[(size_t)rr * nP + p] != INT_MAX) {
                    sm_idx[p] = rsm_idx[(size_t)rr * nP + p];
                    sm_le[p] = acc + rsm_le[(size_t)rr * nP + p];
                }
                int h = rhist[(size_t)rr * nP + p];
                rhist[(size_t)rr * nP + p] = acc; /* exclusive prefix */
                acc += h;
            }
            total[p] = acc;
        }
        int nev = 0;
        /* gather events of all ranges in range order == index order */
        static ev_t *L = NULL; static int Lcap = 0;
        { int need = 0; for (int rr = 0; rr < nranges; rr++) need += evcnt[rr]; if (need > Lcap) { Lcap = need * 2 + 64; L = (ev_t *)realloc(L, sizeof(ev_t) * (size_t)Lcap); } }
        for (int rr = 0; rr < nranges; rr++)
            for (int k = 0; k < evcnt[rr]; k++) { ev_t e = EV[evstart[rr] + k]; e.le += rhist[(size_t)rr * nP + e.p]; L[nev++] = e; }
        /* R1: first definite caller per bucket */
        for (int p = 0; p < nP; p++) fcall[p] = sm_idx[p];
        for (int k = 0; k < nev; k++) if (L[k].type == EV_CALLER && L[k].idx < fcall[L[k].p]) fcall[L[k].p] = L[k].idx;
        /* R1b: reset-cycle candidates, SEQUENTIAL in index order (rare) */
        for (int k = 0; k < nev; k++) if (L[k].type == EV_RESETCAND) {
            if (fcall[L[k].q] < L[k].idx) L[k].type = 0; /* bumped before its turn: does not join */
            else if (L[k].idx < fcall[L[k].p]) fcall[L[k].p] = L[k].idx;
        }
        /* R3: every call's check */
        ngranted = 0;
        int nsingle_total = 0;
        static int *single_idx = NULL; static int scap = 0;
        if (nev + nP > scap) { scap = (nev + nP) * 2 + 64; single_idx = (int *)realloc(single_idx, sizeof(int) * (size_t)scap); }
        int ns = 0;
        for (int k = 0; k < nev + nP; k++) {
            int idx, p, le, is_pre;
            if (k < nev) { if (L[k].type != EV_CALLER && L[k].type != EV_RESETCAND) continue; idx = L[k].idx; p = L[k].p; le = L[k].le; is_pre = L[k].is_pre; }
            else { p = k - nev; if (sm_idx[p] == INT_MAX || sm_idx[p] != fcall[p]) continue; idx = sm_idx[p]; le = sm_le[p]; is_pre = 1; }
            int first = idx == fcall[p];
            int prev = -1; /* previous caller on the same bucket */
            int post = 0;
            /* post-turn members: only when such events exist (rare); O(nev) scans */
            for (int j = 0; j < nev; j++) {
                if ((L[j].type == EV_CALLER || L[j].type == EV_RESETCAND) && L[j].p == p && L[j].idx < idx && L[j].idx > prev) prev = L[j].idx;
            }
            if (!first && sm_idx[p] == fcall[p] && sm_idx[p] < idx && sm_idx[p] > prev) prev = sm_idx[p];
            for (int j = 0; j < nev; j++) {
                if (L[j].p != p || L[j].idx >= idx) continue;
                if (L[j].type == EV_RJOIN && L[j].idx > prev) post++;
                if (L[j].type == EV_PASSIVE && first) post++; /* idx < fcall: still a member at the first call */
            }
            int check = 1 + (first ? total[p] - le : 0) + post;
            { /* the cluster kernel's set-based formulation of the same count: no index-ordered prefix needed */
                int nlv = 0, rj = 0;
                for (int j = 0; j < nE; j++) nlv += Ep[j] == p && Eidx[j] < idx;
                for (int j = 0; j < nev; j++) rj += L[j].type == EV_RJOIN && L[j].p == p && L[j].idx < idx && L[j].idx > prev;
                int check2 = 1 + (first ? total[p] - is_pre - nlv : 0) + rj;
                if (check2 != check) formula_mismatch = 1;
            }
            calls++;
            if (idx > lcall[p]) lcall[p] = idx;
            if (check == 1) { totalPreambleTxop++; single_idx[ns++] = idx; }
            else if (withnoma) { collisionPreambles += check; totalPreambleTxop += check; }
            else { collisionPreambles++; totalPreambleTxop++; }
        }
        nsingle_total = ns;
        /* R4: grants in index order of the singleton calls (Beta.c:336-347) */
        {
            int G = cfg->nGrantUL - 1 - grantCheck;
            if (G < 0) G = 0;
            for (int a = 0; a < ns; a++) { /* rank by counting */
                int rank = 0;
                for (int b = 0; b < ns; b++) rank += single_idx[b] < single_idx[a];
                if (rank < G) granted[ngranted++] = single_idx[a];
            }
            grantCheck += nsingle_total;
        }
        if (nSuccess == nUE) break;
    }
    /* final apply of the last executed step's outcomes */
    if (tlast >= 0) {
        const int tp = tlast;
        for (int i = 0; i < nUE; i++) {
            mue_t *u = &U[i];
            if (u->pend == PEND_NONE) continue;
            int isg = 0;
            for (int g = 0; g < ngranted; g++) isg |= granted[g] == i;
            switch (u->pend) {
            case PEND_STAY: case PEND_CALLER:
                if (isg) { u->active = 2; u->txTime = tp + 11; u->conn = 0; } else u->txTime = tp + 1;
                break;
            case PEND_RESET: {
                int q = u->bo, tmp = u->txTime, bumped = fcall[q] < i;
                int tx = m_align(tp + bumped + tmp, aT);
                if (tx == tp) { u->bo = 0; if (isg) { u->active = 2; u->txTime = tp + 11; u->conn = 0; } else u->txTime = tp + 1; }
                else { u->txTime = tx; u->bo = m_enc_bo(tx - tp, tp); }
            } break;
            case PEND_PASSIVE: if (fcall[u->preamble] != INT_MAX) u->txTime = tp + 1; break;
            case PEND_RJOIN: if (lcall[u->preamble] > i) u->txTime = tp + 1; break;
            }
            u->pend = PEND_NONE;
        }
    }
    /* aggregate + materialise the logged fields */
    float totalDelay = 0; int64_t sumTimer = 0; int ptcSum = 0, fcSum = 0;
    const int tend = tlast + 1; /* number of executed subframes */
    for (int i = 0; i < nUE; i++) {
        mue_t *u = &U[i];
        int timer = u->active == -1 ? -1 : (u->active == 0 ? u->tb : tend - u->tb);
        if (u->active == 0) { totalDelay += (float)timer; sumTimer += timer; ptcSum += u->ptc; fcSum += u->fc; }
        if (ue_out) {
            oracle_ue *o = &ue_out[i];
            o->idx = i; o->timer = timer; o->active = u->active; o->txTime = u->txTime; o->firstTxTime = u->ftt;
            o->secondTxTime = u->stt; o->nowBackoff = m_now_bo(u->bo, tend); o->preamble = u->preamble;
            o->preambleChange = u->preamble != -1; o->rarWindow = u->rar; o->maxRarCounter = u->mrc;
            o->preambleTxCounter = u->ptc; o->msg2Flag = (u->active == 2 || u->active == 0);
            o->connectionRequest = u->conn == 2 ? 48 : u->conn; o->msg4Flag = u->active == 0; o->failCount = u->fc;
        }
        philox_draws += u->nd;
    }
    memset(res, 0, sizeof(*res));
    res->time_exit = time; res->maxTime = maxTime; res->nSuccessUE = nSuccess; res->failedUEs = nUE - nSuccess;
    res->preambleTxCount = ptcSum; res->failCounts = fcSum; res->collisionPreambles = collisionPreambles;
    res->totalPreambleTxop = totalPreambleTxop; res->activeCheck = activeCheck; res->nAccessUE = nAccessUE;
    res->continueFaliedUEs = continueFailed; res->finalSuccessUEs = finalSuccess; res->totalDelay = totalDelay;
    res->sumTimer = sumTimer; res->steps = steps; res->collisionCalls = calls;
    res->draws = rng_mode == ORACLE_RNG_GLIBC ? R.base - stream_off : 0;
    if (rng_mode == ORACLE_RNG_PHILOX) { uint64_t d = 0; for (int i = 0; i < nUE; i++) d += U[i].nd; res->draws = d; }
    int rc = R.overflow ? -2 - R.overflow : (formula_mismatch ? -5 : 0); /* -3 stream exhausted, -4 draw-count misprediction */
    free(Eidx); free(Ep); free(U); free(EV); free(evcnt); free(evstart); free(rhist); free(rsm_idx); free(rsm_le); free(rdraws);
    free(total); free(fcall); free(lcall); free(sm_idx); free(sm_le); free(granted); free(sched);
    return rc;
}

/* glibc draw stream as an array (what the product generates on the device) */
void model_glibc_stream(unsigned int seed, uint64_t n, int32_t *out) {
    glibc_rand_t g;
    glibc_srand(&g, seed);
    for (uint64_t i = 0; i < n; i++) out[i] = glibc_rand(&g);
}
