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
    __device__ __forceinline__ void ptc_set1(const int) { ptc = 1; }
    __device__ __forceinline__ void ptc_inc(const int) { ptc++; }
    __device__ __forceinline__ void ftt_set(const int, const int v) { ftt = v; }
    __device__ __forceinline__ void stt_set(const int, const int v) { stt = v; }
    __device__ __forceinline__ void fcnt_zero(const int) { fcnt = 0; }
    __device__ __forceinline__ void fcnt_inc(const int) { fcnt++; }
};
// first / last caller per bucket of the previous subframe, as plain tables (LDS)
struct CallTables {
    const int *fc, *lc;
    __device__ __forceinline__ int fcall(const int q) const { return fc[q]; }
    __device__ __forceinline__ int lcall(const int p) const { return lc[p]; }
};

// ---- deferred outcome of subframe tp = t - 1 (preambleCollision's side effects, Beta.c:332-366) -------------------------------
// granted: the resolver gave this (singleton-calling) UE an UL grant.  Returns true if the state changed.
template <class TAB>
__device__ __forceinline__ bool ue_apply(UeState &u, const bool granted, const int i, const int tp, const FastMod fmA, const TAB &tab) {
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
__device__ __forceinline__ void ue_activate(UeState &u, const int i, const int t, COLD &cold) {
    u.act = ACT_M1; u.tx = t + 1; u.tb = t;
    cold.ftt_set(i, t + 1);
}

// ---- what the UE does in subframe t, from its own state ----------------------------------------------------------------------
struct UePlan {
    bool isM1, firstsel, backoff, contend, reset, retx, m3first, m3to, busy;
    int need; // rand() calls of this UE in this subframe: 0, 1 or 2
};
__device__ __forceinline__ UePlan ue_plan(const UeState &u, const int t, const int maxRar, const int maxMsg2) {
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
__device__ __forceinline__ UeOut ue_select(UeState &u, const UePlan &pl, const int d1, const int d2, const int i, const int t, const int tmod, const UeK &K,
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
__device__ __forceinline__ unsigned pw_make(const int tj, const int dur, const int pre) {
    return ((unsigned)tj & 0xFFFFu) | ((unsigned)dur << 16) | ((unsigned)pre << 24);
}

// A UE's record as the event body left it at some subframe s <= t - 1, brought to the start of subframe t: the deferred outcome of
// subframe s (ue_apply; a caller or matched UE was recorded with txTime = s + 1 already), then the subframes it was matched in since
// according to the word it was scheduled with (sw) — bumped every time, one RAR-window subframe each.
template <class TAB>
__device__ __forceinline__ void pw_catch_up(UeState &u, const unsigned sw, const bool granted, const int i, const int t, const FastMod fmA, const TAB &tab) {
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
__device__ __forceinline__ unsigned pw_schedule(UeState &u, const int t, const int maxRar) {
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

// resolver-side info word of a special event (20 bits): type[2:0] ispre[3] bucket p[11:4] old bucket q[19:12]
__device__ __forceinline__ int ue_event_info(const UeOut &o) {
    const int ispre = (o.evtype == UEV_CALLER) ? (o.member_pre && o.oldp == o.evp) : (o.evtype == UEV_RESETCAND ? (o.evp == o.evq) : 0);
    return o.evtype | (ispre << 3) | (o.evp << 4) | (o.evq << 12);
}

} // namespace prach
