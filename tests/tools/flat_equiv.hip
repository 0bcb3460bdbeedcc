// tests/tools/flat_equiv.hip — TEST INFRASTRUCTURE (host program, no GPU involved): the batched kernel's branch-free event body
// (prach_ue_body.h: flat_catch_up / flat_plan / flat_select / flat_schedule / flat_event_info) against the branched form every other kernel runs
// (pw_catch_up / ue_plan / ue_select / pw_schedule / ue_event_info), on random UE states, parameters, caller tables and draws — the draws are random
// for EVERY state, also where the UE needs none: neither form may let an unneeded draw reach an output.
// usage: flat_equiv [cases = 2000000] [seed = 1]     exit code 0 = no difference; 1 = a difference (the first one is printed).
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "../../5g-nr-randomaccess_amd/csrc/prach_ue_body.h"

using namespace prach;

static uint64_t rng_s;
static inline uint32_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 16); }
static inline int rint_(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); } // inclusive

struct Case {
    UeState u; ColdRegs cold; int sdur, granted, i, t, tcu, d1, d2;
    UeK K; int fc[64], lc[64];
};

static void gen(Case &c) { // (host only: the generator)
    const int nP = rnd() % 4 ? rint_(1, 64) : (int[]){1, 2, 54, 64}[rnd() % 4];
    const int backoff = rnd() % 4 ? rint_(1, 80) : rint_(1, 2), aT = rnd() % 4 == 0 ? 1 : (rnd() % 3 ? rint_(1, 20) : 5); // (short backoffs / accessTime 1: re-joins in the same subframe)
    c.K.maxRar = rint_(1, 11); c.K.maxMsg2 = (int[]){0, 1, 2, 3, 9, 10, 20}[rnd() % 7]; c.K.aT = aT; c.K.withnoma = rnd() & 1;
    c.K.fmP = make_fastmod(nP); c.K.fmB = make_fastmod(backoff); c.K.fmA = make_fastmod(aT); c.K.fm5 = make_fastmod(5);
    c.t = rint_(20, 60000); c.i = rint_(0, 200000);
    for (int k = 0; k < 64; k++) {
        c.fc[k] = rnd() % 3 == 0 ? INT_MAX : (rnd() % 4 == 0 ? c.i + rint_(-3, 3) : rint_(0, 200000));
        c.lc[k] = rnd() % 3 == 0 ? -1 : (rnd() % 4 == 0 ? c.i + rint_(-3, 3) : rint_(0, 200000));
    }
    const int t = c.t;
    UeState &u = c.u;
    c.sdur = 0; c.granted = 0; c.tcu = t;
    c.cold.ptc = rint_(0, 30); c.cold.ftt = rint_(0, t); c.cold.stt = rint_(0, t); c.cold.fcnt = rint_(0, 30);
    u.tb = rint_(0, t); u.conn = 0; u.pend = PEND_NONE; u.rar = 0; u.mrc = rint_(0, c.K.maxMsg2 + 1); u.pre = rint_(1, nP); u.bo = 0;
    const int kind = rnd() % 10;
    if (kind == 0) { // a lane without a record
        u = UeState{-1, 0, 0, ACT_IDLE, 0, 0, 0, 0, PEND_NONE};
    } else if (kind == 1) { // arrival (as the kernel synthesises it)
        u = UeState{t + 1, t, 0, ACT_M1, 0, 0, 0, 0, PEND_NONE};
        c.cold = ColdRegs{0, t + 1, 0, 0};
    } else if (kind == 2) { // Msg3 phase: scheduled at txTime
        u.act = ACT_M3; u.tx = rnd() % 8 ? t : t + rint_(-2, 50); u.conn = rint_(0, 2); u.rar = rint_(0, c.K.maxRar);
    } else if (kind == 3) { // a deferred outcome, looked at the very next subframe
        u.act = ACT_M1; u.pend = rint_(PEND_RESET, PEND_RJOIN);
        if (u.pend == PEND_RESET) { u.bo = rint_(0, nP - 1); u.tx = rint_(0, backoff - 1); u.rar = 0; u.mrc = 0; }
        else { u.tx = rnd() % 2 ? t - 1 : t + rint_(-2, 30); u.bo = rnd() % 2 ? 0 : u.tx; u.rar = rint_(0, c.K.maxRar); }
    } else { // Msg1 phase: a window that has run out, an unclean schedule, a backoff, a caller / matched UE of the previous subframe
        u.act = ACT_M1;
        c.sdur = rnd() % 3 ? rint_(0, c.K.maxRar > 1 ? c.K.maxRar - 1 : 0) : 0;
        u.rar = rnd() % 2 ? (c.K.maxRar - 1 - c.sdur > 0 ? c.K.maxRar - 1 - c.sdur : 0) : rint_(0, c.K.maxRar);
        u.pend = rnd() % 3 == 0 ? (rnd() & 1 ? PEND_CALLER : PEND_STAY) : PEND_NONE;
        u.tx = c.sdur > 0 ? t - c.sdur : (rnd() % 2 ? t : t + rint_(-3, 40));
        u.bo = rnd() % 3 == 0 ? u.tx : (rnd() % 2 ? 0 : rint_(-5, t + 40));
        if (rnd() % 5 == 0) { c.granted = 1; c.tcu = t - rint_(0, c.sdur); } // granted in one of the subframes since it was scheduled: looked at as if one subframe later
    }
    c.d1 = (int)(rnd() & 0x7fffffffu); c.d2 = (int)(rnd() & 0x7fffffffu);
    if (rnd() % 16 == 0) c.d1 = rnd() % 2 ? 0 : 214748364; // around the 0.1 threshold of Beta.c:374
}

struct Out { UeState u; ColdRegs cold; int need, evtype, evp, evq, member, eclass, info, cs, cc; unsigned word; };

static __host__ __device__ Out run_branched(const Case &c) {
    Out o; o.u = c.u; o.cold = c.cold; o.cs = 0; o.cc = 0;
    const CallTables tab{c.fc, c.lc};
    if (!(c.u.act == ACT_M1 && c.u.pre == 0 && c.u.tx == c.t + 1)) // (an arrival is not caught up: prach_batch.hip)
        pw_catch_up(o.u, pw_make(c.t - c.sdur, c.sdur, 0), c.granted != 0, c.i, c.tcu, c.K.fmA, tab);
    const UePlan pl = ue_plan(o.u, c.t, c.K.maxRar, c.K.maxMsg2);
    o.need = pl.need;
    const UeOut uo = ue_select(o.u, pl, c.d1, c.d2, c.i, c.t, c.t % c.K.aT, c.K, o.cold, o.cs, o.cc);
    o.evtype = uo.evtype; o.evp = uo.evp; o.evq = uo.evq; o.member = uo.member_pre; o.eclass = uo.eclass; o.info = ue_event_info(uo);
    o.word = c.u.act == ACT_IDLE ? PW_IDLE : pw_schedule(o.u, c.t, c.K.maxRar);
    return o;
}
static __host__ __device__ Out run_flat(const Case &c) {
    Out o; o.u = c.u; o.cold = c.cold; o.cs = 0; o.cc = 0;
    const CallTables tab{c.fc, c.lc};
    if (!(c.u.act == ACT_M1 && c.u.pre == 0 && c.u.tx == c.t + 1))
        flat_catch_up(o.u, c.sdur, lm(c.granted != 0), c.i, c.t, c.tcu, c.K.fmA, tab);
    const FlatPlan pl = flat_plan(o.u, c.t, c.K.maxRar, c.K.maxMsg2);
    o.need = pl.need;
    const lmask wn = c.K.withnoma ? -1 : 0, rc_slot = (c.K.aT > 1 && c.t % c.K.aT == 1) ? -1 : 0;
    FlatOut fo = flat_select(o.u, o.cold, pl, c.d1, c.d2, c.t, rc_slot, c.K, wn, o.cs, o.cc);
    o.evtype = fo.evtype; o.evp = fo.evp; o.evq = fo.evq; o.member = fo.member_pre != 0; o.eclass = fo.eclass != 0; o.info = flat_event_info(fo);
    o.word = c.u.act == ACT_IDLE ? PW_IDLE : flat_schedule(o.u, c.t, c.K.maxRar);
    return o;
}
static __host__ __device__ bool same(const Out &a, const Out &b) {
    const bool idle = a.u.act == ACT_IDLE; // (a lane without a record: only that nothing is reported for it)
    return a.need == b.need && a.evtype == b.evtype && a.evp == b.evp && a.evq == b.evq && a.member == b.member && a.eclass == b.eclass && a.cs == b.cs && a.cc == b.cc &&
           (a.evtype == 0 || a.info == b.info) && a.word == b.word &&
           (idle || (a.u.tx == b.u.tx && a.u.tb == b.u.tb && a.u.bo == b.u.bo && a.u.act == b.u.act && a.u.conn == b.u.conn && a.u.pre == b.u.pre && a.u.rar == b.u.rar &&
                     a.u.mrc == b.u.mrc && a.u.pend == b.u.pend && a.cold.ptc == b.cold.ptc && a.cold.ftt == b.cold.ftt && a.cold.stt == b.cold.stt && a.cold.fcnt == b.cold.fcnt));
}
static void show(const char *name, const Out &o) {
    printf("  %-8s need=%d ev=%d evp=%d evq=%d member=%d eclass=%d info=%x cs=%d cc=%d word=%08x | tx=%d tb=%d bo=%d act=%d conn=%d pre=%d rar=%d mrc=%d pend=%d | ptc=%d ftt=%d stt=%d fcnt=%d\n", name,
           o.need, o.evtype, o.evp, o.evq, o.member, o.eclass, o.info, o.cs, o.cc, o.word, o.u.tx, o.u.tb, o.u.bo, o.u.act, o.u.conn, o.u.pre, o.u.rar, o.u.mrc, o.u.pend, o.cold.ptc, o.cold.ftt, o.cold.stt, o.cold.fcnt);
}

#ifndef FLAT_EQUIV_NO_MAIN
int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    rng_s = 0x9E3779B97F4A7C15ull ^ (uint64_t)(argc > 2 ? atol(argv[2]) : 1);
    long needs[3] = {0, 0, 0}, events = 0;
    for (long k = 0; k < n; k++) {
        Case c; gen(c);
        const Out a = run_branched(c), b = run_flat(c);
        needs[a.need]++; events += a.evtype != 0;
        if (!same(a, b)) {
            printf("case %ld DIFFERS: t=%d i=%d sdur=%d granted=%d tcu=%d d1=%d d2=%d | nP=%u backoff=%u aT=%d maxRar=%d maxMsg2=%d withnoma=%d | in: tx=%d tb=%d bo=%d act=%d conn=%d pre=%d rar=%d mrc=%d pend=%d\n",
                   k, c.t, c.i, c.sdur, c.granted, c.tcu, c.d1, c.d2, c.K.fmP.d, c.K.fmB.d, c.K.aT, c.K.maxRar, c.K.maxMsg2, (int)c.K.withnoma,
                   c.u.tx, c.u.tb, c.u.bo, c.u.act, c.u.conn, c.u.pre, c.u.rar, c.u.mrc, c.u.pend);
            show("branched", a); show("flat", b);
            return 1;
        }
    }
    printf("flat_equiv: %ld cases, 0 differences (draws needed: none %ld, one %ld, two %ld; special events %ld)\n", n, needs[0], needs[1], needs[2], events);
    return 0;
}
#endif
