// prach_ue_body.h — the per-UE state machine of one subframe, ONCE, for every trial kernel (gfx950 only).
//
// One subframe of the reference's UE loop (RandomAccessSimulatorBeta.c:150-177 / RandomAccessWithNOMA.c:300-327) for one UE per lane,
// on the UE's own state, in four steps that every kernel runs in this order:
//   ue_apply     the deferred outcome of the previous subframe's preambleCollision (Beta.c:332-366): grant, bump, the stale-txTime
//                reset cycle (Beta.c:266), passive members, Msg3 re-entries;
//   ue_activate  arrival (Beta.c:136-146 / activateUEs WithNOMA:383-394);
//   ue_plan      what the UE does in this subframe and how many rand() draws that takes — from its own pre-step state only
//                (SURVEY 7.4): first preamble, RAR-window expiry -> retransmit / reset cycle, Msg3, Msg3 timeout;
//   ue_select    selectPreamble (Beta.c:229-312) / requestResourceAllocation (Beta.c:371-411) with the draws the caller fetched
//                (Philox in place, Philox computed ahead, or the reference's rand() stream at an index-ordered prefix position).
// What differs between the kernels is WHERE things live, not what happens: the hot record (global 16 B, global 8 + 4 B, LDS, the
// 32-byte event record of prach_batch.hip), the cold logged fields (global arrays updated with stores / atomics, or registers of a
// record the lane already holds) and the bucket bookkeeping / event lists the caller feeds afterwards.  The cold fields are
// reached through a policy object COLD with
//     ptc_set1(i)  ptc_inc(i)  ftt_set(i, v)  stt_set(i, v)  fcnt_zero(i)  fcnt_inc(i)
// (preambleTxCounter, firstTxTime, secondTxTime, failCount of UE i), the previous subframe's first / last caller tables through
// TAB with fcall(q) / lcall(p).
#pragma once
#include "prach_device_fn.h"
#include <limits.h>

namespace prach {

// wave-uniform parameters of the state machine
struct UeK {
    int maxRar, maxMsg2, aT;
    bool withnoma;
    FastMod fmP, fmB, fmA, fm5;
};

// special events handed to the resolver (the cluster kernels' numbering; prach_kernels.hip maps them onto its own list format)
constexpr int UEV_NONE = 0, UEV_CALLER = 1, UEV_RESETCAND = 2, UEV_RJOIN = 3;

// cold logged fields in global arrays (stores and fire-and-forget atomics: the lane holds no copy)
struct ColdGlobal {
    PRACH_G int *ptc, *ftt, *stt, *fcnt;
    __device__ __forceinline__ void ptc_set1(const int i) const { ptc[i] = 1; }
    __device__ __forceinline__ void ptc_inc(const int i) const { gadd(&ptc[i], 1); }
    __device__ __forceinline__ void ftt_set(const int i, const int v) const { ftt[i] = v; }
    __device__ __forceinline__ void stt_set(const int i, const int v) const { stt[i] = v; }
    __device__ __forceinline__ void fcnt_zero(const int i) const { fcnt[i] = 0; }
    __device__ __forceinline__ void fcnt_inc(const int i) const { gadd(&fcnt[i], 1); }
};
// ... in registers, as part of a record the lane has loaded and will store back
struct ColdRegs {
    int ptc, ftt, stt, fcnt;
    PRACH_HD void ptc_set1(const int) { ptc = 1; }
    PRACH_HD void ptc_inc(const int) { ptc++; }
    PRACH_HD void ftt_set(const int, const int v) { ftt = v; }
    PRACH_HD void stt_set(const int, const int v) { stt = v; }
    PRACH_HD void fcnt_zero(const int) { fcnt = 0; }
    PRACH_HD void fcnt_inc(const int) { fcnt++; }
};
// first / last caller per bucket of the previous subframe, as plain tables (LDS)
struct CallTables {
    const int *fc, *lc;
    PRACH_HD int fcall(const int q) const { return fc[q]; }
    PRACH_HD int lcall(const int p) const { return lc[p]; }
};

// ---- deferred outcome of subframe tp = t - 1 (preambleCollision's side effects, Beta.c:332-366) -------------------------------
// granted: the resolver gave this (singleton-calling) UE an UL grant.  Returns true if the state changed.
template <class TAB>
PRACH_HD bool ue_apply(UeState &u, const bool granted, const int i, const int tp, const FastMod fmA, const TAB &tab) {
    if (u.pend == PEND_NONE) return false;
    if (granted) { // Beta.c:338-343
        u.act = ACT_M3; u.tx = tp + 11; u.conn = 0;
        if (u.pend == PEND_RESET) u.bo = 0;
    } else if (u.pend == PEND_STAY || u.pend == PEND_CALLER) {
        u.tx = tp + 1; // bumped, collided, or singleton without a grant (Beta.c:346,358)
    } else if (u.pend == PEND_RESET) {
        const int q = u.bo, tmp = u.tx; // (the record carries the old bucket and the drawn backoff until now)
        const int bumped = tab.fcall(q) < i ? 1 : 0; // stale txTime seen by Beta.c:266
        const int x = slot_align_fm(tp + bumped + tmp, fmA);
        if (x == tp) { u.bo = 0; u.tx = tp + 1; } // re-joined and called, no grant
        else { u.tx = x; u.bo = x; }
    } else if (u.pend == PEND_PASSIVE) {
        if (tab.fcall(u.pre - 1) != INT_MAX) u.tx = tp + 1;
    } else { // PEND_RJOIN
        if (tab.lcall(u.pre - 1) > i) u.tx = tp + 1;
    }
    u.pend = PEND_NONE;
    return true;
}

// ---- activation (Beta.c:136-146; the two draws of activateUEs, WithNOMA:393-394, are the caller's business) -------------------
template <class COLD>
PRACH_HD void ue_activate(UeState &u, const int i, const int t, COLD &cold) {
    u.act = ACT_M1; u.tx = t + 1; u.tb = t;
    cold.ftt_set(i, t + 1);
}

// ---- what the UE does in subframe t, from its own state ----------------------------------------------------------------------
struct UePlan {
    bool isM1, firstsel, backoff, contend, reset, retx, m3first, m3to, busy;
    int need; // rand() calls of this UE in this subframe: 0, 1 or 2
};
PRACH_HD UePlan ue_plan(const UeState &u, const int t, const int maxRar, const int maxMsg2) {
    UePlan p;
    p.isM1 = u.act == ACT_M1;
    const int nb = now_backoff(u.bo, t);
    p.firstsel = p.isM1 && u.pre == 0;
    p.backoff = p.isM1 && u.pre != 0 && nb > 0;
    p.contend = p.isM1 && u.pre != 0 && nb <= 0;
    const bool expire = p.contend && (u.rar + 1 >= maxRar);
    p.reset = expire && u.mrc >= maxMsg2;
    p.retx = expire && !p.reset;
    const bool m3due = u.act == ACT_M3 && u.tx == t;
    p.m3first = m3due && u.conn == 0;
    p.m3to = m3due && u.conn != 0;
    p.need = (p.firstsel || p.retx || p.m3first) ? 1 : ((p.reset || p.m3to) ? 2 : 0);
    p.busy = p.isM1 || m3due;
    return p;
}

// ---- selectPreamble / requestResourceAllocation on own state ------------------------------------------------------------------
struct UeOut {
    int evtype, evp, evq; // special event for the resolver (UEV_*), its bucket, (reset candidate) its old bucket
    int oldp;             // the bucket the UE was in when the subframe began (-1: none)
    bool member_pre;      // matched by a preambleCollision scan right now: active == 1, txTime == t, preamble == oldp
    bool eclass;          // pre-member that leaves its bucket at its own turn without calling on it
    bool passive;         // Beta.c: matched while nowBackoff > 0 (counted and bumped, never calls: Beta.c:161)
    bool dirty;           // the state changed
};
template <class COLD>
PRACH_HD UeOut ue_select(UeState &u, const UePlan &pl, const int d1, const int d2, const int i, const int t, const int tmod, const UeK &K,
                                           COLD &cold, int &c_succ, int &c_contf) { // tmod = t mod accessTime
    UeOut o;
    o.evtype = UEV_NONE; o.evp = 0; o.evq = 0;
    o.oldp = u.pre - 1;
    o.member_pre = pl.isM1 && u.tx == t && u.pre != 0;
    o.eclass = false; o.passive = false; o.dirty = false;
    const int oldp = o.oldp;
    if (pl.firstsel) { // Beta.c:231-239
        u.pre = fastmod(d1, K.fmP) + 1; u.rar = 0; u.mrc = 0; u.bo = 0;
        cold.ptc_set1(i);
        if (K.withnoma) cold.fcnt_zero(i);
        if (u.tx == t) { u.pend = PEND_CALLER; o.evtype = UEV_CALLER; o.evp = u.pre - 1; }
        o.dirty = true;
    } else if (pl.backoff) { // in backoff (Beta.c:243 false)
        if (o.member_pre) { // WithNOMA:310 calls whatever nowBackoff is
            if (K.withnoma) u.pend = PEND_STAY; else { u.pend = PEND_PASSIVE; o.passive = true; }
            o.dirty = true;
        }
    } else if (pl.contend) {
        u.rar++; // Beta.c:245
        o.dirty = true;
        if (pl.reset) { // Beta.c:250-281
            if (K.withnoma) { c_contf++; cold.fcnt_inc(i); }
            const int newp = fastmod(d1, K.fmP);
            const int tmp = fastmod(d2, K.fmB);
            u.rar = 0; u.mrc = 0; u.tb = t;
            cold.ptc_set1(i); cold.ftt_set(i, t + 1);
            u.pre = newp + 1;
            if (o.member_pre) { // txTime depends on whether an earlier caller bumped this UE: defer
                u.pend = PEND_RESET; u.tx = tmp; u.bo = oldp;
                o.eclass = true;
                if (tmp == 0 && K.aT > 1 && tmod == 1) { o.evtype = UEV_RESETCAND; o.evp = newp; o.evq = oldp; }
            } else {
                u.tx = slot_align_fm(u.tx + tmp, K.fmA);
                u.bo = enc_backoff(u.tx - t, t);
                if (u.tx == t) { u.pend = PEND_CALLER; o.evtype = UEV_CALLER; o.evp = newp; }
            }
        } else if (pl.retx) { // Beta.c:282-308
            u.rar = 0; u.mrc++;
            cold.ptc_inc(i);
            const int tmp = fastmod(d1, K.fmB);
            u.tx = slot_align_fm(t + tmp, K.fmA);
            u.bo = enc_backoff(u.tx - t, t);
            cold.stt_set(i, u.tx);
            if (u.tx == t) { u.pend = PEND_CALLER; o.evtype = UEV_CALLER; o.evp = oldp; } // the "late joiner"
            else if (o.member_pre) o.eclass = true;
        } else if (o.member_pre) {
            u.pend = PEND_STAY;
        }
    } else if (pl.m3first) { // Beta.c:372-383
        u.conn = 1;
        const float pf = (float)d1 / (float)2147483647; // (float)RAND_MAX == 2^31
        if ((double)pf > 0.1) { u.act = ACT_DONE; u.tb = (t - u.tb) + 6; c_succ++; }
        else { u.conn = 2; u.tx += 48; }
        o.dirty = true;
    } else if (pl.m3to) { // Msg3 timeout, Beta.c:384-410
        c_contf++;
        const int tmp = fastmod(d1, K.fmB);
        u.tx = slot_align_fm(u.tx + tmp, K.fm5); // hard-coded accessTime = 5, Beta.c:389
        u.act = ACT_M1;
        u.bo = enc_backoff(u.tx - t, t);
        u.pre = fastmod(d2, K.fmP) + 1;
        u.tb = t; u.rar = 0; u.mrc = 0; u.conn = 0;
        if (K.withnoma) cold.fcnt_inc(i);
        if (u.tx == t) { u.pend = PEND_RJOIN; o.evtype = UEV_RJOIN; o.evp = u.pre - 1; }
        o.dirty = true;
    }
    return o;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// SCHEDULE WORDS (prach_batch.hip): what the event body knows of a UE's future when it lets go of it — one 32-bit word (its window length also rides in the UE's record):
//     [15:0]  tj   subframe from which the UE is matched by preambleCollision scans (its txTime)               0xFFFF: never
//     [21:16] dur  number of subframes it then contends with its RAR window open (Beta.c:245): matched in [tj, tj + dur)
//     [29:24] preamble       [30] finished for good
// A UE in steady contention is bumped every subframe (Beta.c:346,358) and counts one RAR-window subframe each time (Beta.c:245), so
// its trajectory until the window closes is known when it is scheduled: it enters its bucket's histogram / lowest-index tables for
// [tj, tj + dur) (the join calendar) and is an EVENT for the body again at tj + dur (window expiry, Msg3, a deferred outcome) — or one
// subframe after an UL grant.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr unsigned PW_IDLE = 0x0000FFFFu;  // not arrived yet
constexpr unsigned PW_DONE = 0x4000FFFFu;  // finished for good
constexpr int PW_MAX_RAR = 64, PW_MAX_PREAMBLES = 64, PW_MAX_SUBFRAMES = 65000;
PRACH_HD unsigned pw_make(const int tj, const int dur, const int pre) {
    return ((unsigned)tj & 0xFFFFu) | ((unsigned)dur << 16) | ((unsigned)pre << 24);
}

// A UE's record as the event body left it at some subframe s <= t - 1, brought to the start of subframe t: the deferred outcome of
// subframe s (ue_apply; a caller or matched UE was recorded with txTime = s + 1 already), then the subframes it was matched in since
// according to the word it was scheduled with (sw) — bumped every time, one RAR-window subframe each.
template <class TAB>
PRACH_HD void pw_catch_up(UeState &u, const unsigned sw, const bool granted, const int i, const int t, const FastMod fmA, const TAB &tab) {
    const int tp = t - 1;
    if (u.pend == PEND_CALLER || u.pend == PEND_STAY) {
        if (granted) { u.act = ACT_M3; u.tx = tp + 11; u.conn = 0; } // (a grant is applied the very next subframe: s == tp)
        u.pend = PEND_NONE;
    } else if (u.pend != PEND_NONE) {
        ue_apply(u, granted, i, tp, fmA, tab); // (PEND_RESET / PASSIVE / RJOIN are always looked at the very next subframe)
    } else if (granted) { // a matched UE that was its bucket's only member and called (Beta.c:332-343)
        u.act = ACT_M3; u.tx = tp + 11; u.conn = 0;
    }
    const int stj = (int)(sw & 0xFFFFu), sdur = (int)((sw >> 16) & 0x3Fu);
    if (sdur > 0 && t > stj) { u.rar += t - stj; if (!granted) u.tx = t; } // (a granted UE counted its window subframes too: Beta.c:245 runs before the call)
}

// the schedule word of a UE after the event body of subframe t
PRACH_HD unsigned pw_schedule(UeState &u, const int t, const int maxRar) {
    if (u.act == ACT_DONE) return PW_DONE;
    if (u.act == ACT_M3) return u.tx > t ? pw_make(u.tx, 0, 0) : PW_IDLE; // Msg3 / Msg4 at txTime (a txTime in the past never comes: Beta.c:167)
    if (u.pend == PEND_RESET || u.pend == PEND_PASSIVE || u.pend == PEND_RJOIN) return pw_make(t + 1, 0, 0); // outcome needs the caller tables of t
    if (u.pend == PEND_CALLER || u.pend == PEND_STAY) u.tx = t + 1; // bumped, collided, or singleton without a grant (a grant: noted by the resolver, applied when the record comes up)
    // The clean cases: asleep until txTime (nowBackoff runs out exactly then: every reschedule sets nowBackoff = txTime - time,
    // Beta.c:279,305,399) or contending from the next subframe on (nowBackoff <= 0); from txTime on the UE is matched and counts its
    // RAR window (Beta.c:245) until the window closes.  Anything else (never seen with the reference's parameters) is simply looked
    // at again in the next subframe by the full body.
    const bool clean = u.tx > t && (u.bo > 0 ? u.bo == u.tx : u.tx == t + 1);
    if (!clean) return pw_make(t + 1, 0, u.pre - 1);
    return pw_make(u.tx, max(0, maxRar - 1 - u.rar), u.pre - 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// THE SAME STATE MACHINE WITHOUT BRANCHES (prach_batch.hip's event body with Philox draws).
// pw_catch_up / ue_plan / ue_select / pw_schedule restated case by case on per-lane masks (all ones or zero in a vector register, opaque to the
// optimiser) and bit selects: every lane computes every case's new field values and selects.  A wavefront of event UEs is a mix of all cases, so
// the branched form runs every block anyway, each behind its own exec-mask bookkeeping (s_and_saveexec / s_cbranch_execz / s_or) with the
// conditions' boolean algebra in scalar registers.  The batched kernel is bound by instruction issue (profiles/r04_grid.md, section 5): this form
// needs a third of the branched one's scalar and branch instructions at the same number of vector instructions (config 3: 4.07 -> 2.54 x 10^11
// instructions per launch, 575 -> 448 ms).  Case by case the lines below cite the branched form above, which stays the form every other kernel
// (and the reference-stream instantiation of the batched one) runs: the two are compared through the oracle by every batched-kernel test
// (tests/test_gpu_parity.py, scripts/gpu_batch_check.sh).
// ---------------------------------------------------------------------------------------------------------------------------------
typedef int lmask;
PRACH_HD lmask lm(const bool c) { // v_cmp + v_cndmask: no scalar instruction
    int m = c ? -1 : 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(m)); // (the value is opaque from here on: the compiler cannot turn the mask algebra back into branches / scalar-register booleans)
#endif
    return m;
}
PRACH_HD int lsel(const lmask m, const int a, const int b) { return (a & m) | (b & ~m); } // v_bfi_b32
// A mask that leaves one of the functions below is made opaque again: the compiler folds trees of &, |, ~ into gfx950's three-input v_bitop3_b32, and ROCm 7.2's
// selected a WRONG one when the plan's and the select's algebra were folded across the function boundary in one kernel shape (the build is bit-exact with the
// instruction switched off, and with these barriers: LABNOTES, round 4).  With the barriers every tree lies inside one function, whatever the caller looks like —
// and tests/tools/gpu_flat_equiv.hip runs exactly these functions on the device against the branched form.  Costs no instruction.
PRACH_HD void lopaque(lmask &m) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(m));
#else
    (void)m;
#endif
}
PRACH_HD int fastmod_flat(const int x, const FastMod f) { // = fastmod
    const unsigned q = mulhi32((unsigned)x, f.M);
    const unsigned r = (unsigned)x - q * f.d;
    const unsigned r2 = r - f.d; // (wraps above r when r < d; d == 1: make_fastmod's M makes r = 1 for x > 0)
    return (int)(r2 < r ? r2 : r); // = v_min_u32 (spelled out: on the host `min` of two unsigned picks the int overload)
}
PRACH_HD int slot_align_flat(const int sub, const FastMod aT) { // = slot_align_fm: m == 0: sub + 1, m == 1: sub, else sub + (aT - m + 1)
    const int m = fastmod_flat(sub, aT);
    return sub + ((int)aT.d + 1 - m) - ((int)aT.d & lm(m < 2));
}

// pw_catch_up: the record as the event body left it (scheduled with a window of sdur subframes ending at t) brought to the start of subframe tcu
// (t, or the subframe after the UL grant noted for it)
template <class TAB>
PRACH_HD void flat_catch_up(UeState &u, const int sdur, const lmask granted, const int i, const int t, const int tcu, const FastMod fmA, const TAB &tab) {
    const int tp = tcu - 1;
    const lmask mreset = lm(u.pend == PEND_RESET), mpass = lm(u.pend == PEND_PASSIVE), mrj = lm(u.pend == PEND_RJOIN);
    const int fc = tab.fcall(lsel(mreset, u.bo, u.pre - 1) & 63), lc = tab.lcall((u.pre - 1) & 63); // (every lane reads: the tables are LDS)
    const int x = slot_align_flat(tp + (1 & lm(fc < i)) + u.tx, fmA);   // PEND_RESET: the stale txTime of Beta.c:266 (ue_apply)
    const lmask xeq = lm(x == tp);
    const lmask bump = (mpass & lm(fc != INT_MAX)) | (mrj & lm(lc > i)); // PEND_PASSIVE / PEND_RJOIN (ue_apply)
    const int tx = lsel(mreset, lsel(xeq, tp + 1, x), lsel(bump, tp + 1, u.tx));
    const int bo = lsel(mreset, x & ~xeq, u.bo);
    u.tx = lsel(granted, tp + 11, tx);                                   // Beta.c:338-343
    u.bo = bo & ~(granted & mreset);
    u.act = lsel(granted, ACT_M3, u.act);
    u.conn &= ~granted;
    u.pend = PEND_NONE;
    const lmask mw = lm(sdur > 0) & lm(tcu > t - sdur);                  // the subframes it was matched in since it was scheduled
    u.rar += (tcu - (t - sdur)) & mw;
    u.tx = lsel(mw & ~granted, tcu, u.tx);
}

struct FlatPlan { lmask isM1, pre0, firstsel, backoff, reset, retx, stay, m3first, m3to; int need; };
PRACH_HD FlatPlan flat_plan(const UeState &u, const int t, const int maxRar, const int maxMsg2) { // = ue_plan
    FlatPlan p;
    p.isM1 = lm(u.act == ACT_M1); p.pre0 = lm(u.pre == 0);
    const lmask inbo = lm(u.bo > t); // now_backoff(bo, t) > 0 (t >= 0)
    const lmask hasp = p.isM1 & ~p.pre0;
    p.firstsel = p.isM1 & p.pre0;
    p.backoff = hasp & inbo;
    const lmask contend = hasp & ~inbo;
    const lmask expire = contend & lm(u.rar + 1 >= maxRar);
    p.reset = expire & lm(u.mrc >= maxMsg2);
    p.retx = expire & ~p.reset;
    p.stay = contend & ~expire;
    const lmask m3due = lm(u.act == ACT_M3) & lm(u.tx == t), c0 = lm(u.conn == 0);
    p.m3first = m3due & c0; p.m3to = m3due & ~c0;
    p.need = (1 & (p.firstsel | p.retx | p.m3first)) | (2 & (p.reset | p.m3to));
    lopaque(p.isM1); lopaque(p.pre0); lopaque(p.firstsel); lopaque(p.backoff); lopaque(p.reset); lopaque(p.retx); lopaque(p.stay); lopaque(p.m3first); lopaque(p.m3to);
    return p;
}

// what finish() of prach_batch.hip needs of a UE's subframe
struct FlatOut { int evtype, evp, evq, oldp; lmask member_pre, eclass; unsigned word; };
// = ue_select (ColdRegs only: the cold fields ride in the record)
PRACH_HD FlatOut flat_select(UeState &u, ColdRegs &cold, const FlatPlan &p, const int d1, const int d2, const int t, const lmask rc_slot /* aT > 1 && t mod aT == 1 */,
                                               const UeK &K, const lmask withnoma, int &c_succ, int &c_contf) {
    FlatOut o;
    const int oldp = u.pre - 1;
    const lmask txnow = lm(u.tx == t);
    const lmask member = p.isM1 & txnow & ~p.pre0;
    // a UE reduces each draw by ONE modulus: the first by the preamble count at a first selection / reset cycle (Beta.c:231,251), by the backoff window at a
    // retransmission / Msg3 timeout (Beta.c:296,388); the second by the backoff window in a reset cycle (Beta.c:252), by the preamble count at a Msg3 timeout (Beta.c:400)
    const lmask pfirst = p.firstsel | p.reset;
    FastMod f1, f2;
    f1.d = (unsigned)lsel(pfirst, (int)K.fmP.d, (int)K.fmB.d); f1.M = (unsigned)lsel(pfirst, (int)K.fmP.M, (int)K.fmB.M);
    f2.d = (unsigned)lsel(p.reset, (int)K.fmB.d, (int)K.fmP.d); f2.M = (unsigned)lsel(p.reset, (int)K.fmB.M, (int)K.fmP.M);
    const int r1 = fastmod_flat(d1, f1), r2 = fastmod_flat(d2, f2);
    const int fP1 = r1, fB1 = r1, fB2 = r2, fP2 = r2; // (each name is only used under the cases it is valid for)
    // the one new txTime a subframe can bring: reset cycle outside the bucket (Beta.c:266-279), retransmission (Beta.c:296-305), Msg3 timeout (Beta.c:389-399)
    const lmask rsm = p.reset & member, rsn = p.reset & ~member;
    const lmask S = rsn | p.retx | p.m3to;
    FastMod fm;
    fm.d = (unsigned)lsel(p.m3to, (int)K.fm5.d, (int)K.fmA.d); fm.M = (unsigned)lsel(p.m3to, (int)K.fm5.M, (int)K.fmA.M);
    const int sa = slot_align_flat(lsel(p.retx, t, u.tx) + lsel(p.reset, fB2, fB1), fm);
    const lmask sanow = S & lm(sa == t);
    const float pf = (float)d1 / (float)2147483647;                      // Beta.c:374 ((float)RAND_MAX == 2^31)
    const lmask ok = p.m3first & lm((double)pf > 0.1), nok = p.m3first & ~ok;
    const lmask fs = p.firstsel;
    const lmask callF = fs & txnow, callS = (rsn | p.retx) & sanow, rj = p.m3to & sanow;
    const lmask rc = rsm & lm(fB2 == 0) & rc_slot;                       // reset cycle that may re-join its slot (ue_select: UEV_RESETCAND)
    o.evtype = (UEV_CALLER & (callF | callS)) | (UEV_RESETCAND & rc) | (UEV_RJOIN & rj);
    o.evp = (fP1 & (callF | (rsn & sanow) | rc)) | (oldp & (p.retx & sanow)) | (fP2 & rj);
    o.evq = oldp & rc;
    o.oldp = oldp; o.member_pre = member;
    o.eclass = rsm | (p.retx & ~sanow & member);
    u.pend = (PEND_CALLER & (callF | callS)) | (PEND_RESET & rsm) | (PEND_RJOIN & rj) | (lsel(withnoma, PEND_STAY, PEND_PASSIVE) & (p.backoff & member)) | (PEND_STAY & (p.stay & member));
    const lmask z = fs | p.reset | p.m3to;
    u.rar = (u.rar + (1 & p.stay)) & ~(z | p.retx);                      // Beta.c:245, then the resets
    u.mrc = (u.mrc + (1 & p.retx)) & ~z;
    u.pre = lsel(fs | p.reset, fP1 + 1, lsel(p.m3to, fP2 + 1, u.pre));
    const int enc = lsel(lm(sa > t), sa, sa - t);                        // enc_backoff(sa - t, t)
    u.bo = lsel(rsm, oldp, lsel(S, enc, u.bo & ~fs));
    u.tx = lsel(rsm, fB2, lsel(S, sa, u.tx + (48 & nok)));
    u.tb = lsel(p.reset | p.m3to, t, lsel(ok, (t - u.tb) + 6, u.tb));
    u.conn = lsel(p.m3first, lsel(ok, 1, 2), u.conn & ~p.m3to);
    u.act = lsel(ok, ACT_DONE, lsel(p.m3to, ACT_M1, u.act));
    cold.ptc = lsel(fs | p.reset, 1, cold.ptc + (1 & p.retx));
    cold.ftt = lsel(p.reset, t + 1, cold.ftt);
    cold.stt = lsel(p.retx, sa, cold.stt);
    cold.fcnt = (cold.fcnt + (1 & withnoma & (p.reset | p.m3to))) & ~(withnoma & fs);
    c_contf += 1 & ((withnoma & p.reset) | p.m3to);
    c_succ += 1 & ok;
    lopaque(o.member_pre); lopaque(o.eclass);
    return o;
}

// = pw_schedule
PRACH_HD unsigned flat_schedule(UeState &u, const int t, const int maxRar) {
    const lmask done = lm(u.act == ACT_DONE), m3 = lm(u.act == ACT_M3), m1 = ~(done | m3);
    const lmask px = lm((unsigned)(u.pend - PEND_RESET) < 3u), pcs = lm((unsigned)(u.pend - PEND_STAY) < 2u);
    u.tx = lsel(m1 & pcs, t + 1, u.tx);
    const lmask later = lm(u.tx > t);
    const lmask clean = later & lsel(lm(u.bo > 0), lm(u.bo == u.tx), lm(u.tx == t + 1));
    const lmask win = m1 & ~px & clean;
    const int tj = lsel(done, 0xFFFF, lsel(m3, lsel(later, u.tx, 0xFFFF), lsel(win, u.tx, t + 1)));
    const int dur = max(0, maxRar - 1 - u.rar) & win;
    const int pre = (u.pre - 1) & (m1 & ~px);
    return ((unsigned)tj & 0xFFFFu) | ((unsigned)dur << 16) | ((unsigned)pre << 24) | (0x40000000u & (unsigned)done);
}
// = ue_event_info
PRACH_HD int flat_event_info(const FlatOut &o) {
    const int ispre = 1 & ((lm(o.evtype == UEV_CALLER) & o.member_pre & lm(o.oldp == o.evp)) | (lm(o.evtype == UEV_RESETCAND) & lm(o.evp == o.evq)));
    return o.evtype | (ispre << 3) | (o.evp << 4) | (o.evq << 12);
}

// resolver-side info word of a special event (20 bits): type[2:0] ispre[3] bucket p[11:4] old bucket q[19:12]
PRACH_HD int ue_event_info(const UeOut &o) {
    const int ispre = (o.evtype == UEV_CALLER) ? (o.member_pre && o.oldp == o.evp) : (o.evtype == UEV_RESETCAND ? (o.evp == o.evq) : 0);
    return o.evtype | (ispre << 3) | (o.evp << 4) | (o.evq << 12);
}

} // namespace prach
