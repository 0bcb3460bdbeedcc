// prach_noma_act.h — activeUE (NOMA.c:131-192) for ONE UE on the device, its draws supplied by the caller (the UE's own Philox counter:
// noma_activation_kernel in prach_noma.hip; the reference's rand() stream at the UE's position: prach_noma_glibc.hip).
//
// Every +, x, /, sqrt and float <-> double conversion of the reference's expression is an IEEE operation with one correctly rounded result
// (no contraction: the library is built with -ffp-contract=off and the including files carry the pragma) and therefore the reference's;
// cos, sin and log are the device library's and may differ from the host libm's in the last bits, and pow(v, 2) is v x v (the libm's pow is
// within 1 ulp of that).  What the simulation reads is the preamble, the sector (float arithmetic only), the number of draws and the
// ORDER / log-difference of gains.  So the result is FLAGGED — the caller then takes the host's libm instead — whenever a last-bits
// difference could change more than the last bits of the gain:
//   * r cos(angle), r sin(angle) or sqrt(1 + env^2) lies within ACT_BAND double ulps of a float rounding boundary (the three values
//     the reference rounds to float: NOMA.c:176-178,186) — 3 x 2 ACT_BAND / 2^29 = 7e-7 of the UEs;
//   * a gain lies within 1e-12 (relative) of the rejection threshold 1e-7 (NOMA.c:185), is not finite, or a loop ran 4096 times.
// The gains of unflagged UEs are within a few ulp of the reference's (measured: <= 6 on 8e6 UEs); every comparison of gains downstream
// carries a band for that.
#pragma once
#include <hip/hip_runtime.h>

namespace prach {

constexpr long long ACT_BAND = 64;
// Gains downstream.  tests/test_noma.py asserts that the gain of every UNFLAGGED UE is within ACT_GAIN_ULP_ASSERTED ulps of the host libm's
// (measured worst case: 6).  Two gains may each be off by that much in opposite directions, i.e. their difference by 2 x 32 ulps = at most
// 64 x 2^-52 = 1.42e-14 relative to the larger one (an ulp is 2^-52 of a mantissa of 1.0: the worst case).  The resolver treats the ORDER of
// two sorted neighbours as undecided (NOMA_AMBIGUOUS: the trial is rerun with the host-built table) when gb - ga <= ACT_GAIN_ORDER_BAND x gb;
// the band must cover that and does with a margin of 2.8: 4e-14 > 2 x ACT_GAIN_ULP_ASSERTED x 2^-52.  ln g is then within 32 ulp(g) / g + one
// libm ulp, i.e. |error of 10 ln g1 - 10 ln g2| < 1e-12 — the pairing test's own band around 15.0 (1e-9) is a thousand times that.
constexpr int ACT_GAIN_ULP_ASSERTED = 32;
constexpr double ACT_GAIN_ORDER_BAND = 4e-14;
static_assert(ACT_GAIN_ORDER_BAND > 2.0 * ACT_GAIN_ULP_ASSERTED * 2.220446049250313e-16, "the order band covers two gains off by the asserted bound in opposite directions");

__device__ __forceinline__ bool near_float_boundary(const double p) { // could (float)p differ from (float)p' for |p' - p| <= ACT_BAND ulps?
    const unsigned long long bits = (unsigned long long)__double_as_longlong(p);
    long long d = (long long)(bits & ((1ull << 29) - 1ull)) - (1ll << 28); // the 29 bits a float drops; the rounding boundary is their midpoint
    if (d < 0) d = -d;
    return d <= ACT_BAND || !(fabs(p) > 1e-30 && fabs(p) < 1e30); // (outside the float's normal range the dropped bits are others: never happens here)
}

struct ActUe { int pre, sec; double gain, lgain; bool flag; };

// draw(): the next rand() value of this UE, in the reference's order (preamble, angle, the radius rejection loop, the Rayleigh-gain rejection loop)
template <class Draw> __device__ __forceinline__ ActUe noma_active_ue(Draw &&draw, const int nP, const float cell_radius) {
    ActUe o;
    bool flag = false;
    const float pi = 3.14f; // NOMA.c:55
    o.pre = draw() % nP;    // NOMA.c:133
    // (float)rand() / (float)(2147483647) * 2 * pi (NOMA.c:142): (float)2147483647 is 2^31, the division and the doubling are exact scalings
    const float angle = __fmul_rn(__fmul_rn(__fmul_rn((float)draw(), 0x1p-31f), 2.0f), pi);
    const double a = (double)angle, dpi = (double)pi;
    int sec; // NOMA.c:146-163 (the constants are double products of the float pi; `angle >= pi` is the float comparison it is there)
    if (a >= 0 && a < (1. / 3.) * dpi) sec = 0;
    else if (a >= (1. / 3.) * dpi && a < (2. / 3.) * dpi) sec = 1;
    else if (a >= (2. / 3.) * dpi && a < 3.14) sec = 2;
    else if (angle >= pi && a < (4. / 3.) * dpi) sec = 3;
    else if (a >= (4. / 3.) * dpi && a < (5. / 3.) * dpi) sec = 4;
    else sec = 5;
    o.sec = sec;
    float r;
    for (int it = 0;; it++) { // NOMA.c:167-172
        const float u = __fmul_rn((float)draw(), 0x1p-31f);
        r = (float)__dmul_rn((double)cell_radius, __dsqrt_rn((double)u));
        if ((double)r > 35.0) break;
        if (it >= 4096) { flag = true; break; }
    }
    const double px = __dmul_rn((double)r, cos(a)), py = __dmul_rn((double)r, sin(a)); // NOMA.c:176-177
    flag = flag || near_float_boundary(px) || near_float_boundary(py);
    const float x = (float)px, y = (float)py;
    const double env = __dsqrt_rn((double)__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y))); // NOMA.c:178 (float products, float sum)
    const double pld = __dsqrt_rn(__dadd_rn(1.0, __dmul_rn(env, env)));                 // NOMA.c:186
    flag = flag || near_float_boundary(pld);
    const float pathloss = (float)pld;
    double ch_g = 0;
    for (int it = 0; ch_g < 1e-7; it++) { // NOMA.c:185-189
        const double u = __ddiv_rn((double)draw(), 2147483647.0);
        const double rayleigh = __dsqrt_rn(__dmul_rn(-2.0, log(u)));
        const double q = __ddiv_rn(rayleigh, (double)pathloss);
        ch_g = __dmul_rn(q, q);
        if (fabs(__dsub_rn(ch_g, 1e-7)) <= 1e-19 || !(ch_g < 1e300)) flag = true;
        if (it >= 4096) { flag = true; break; }
    }
    o.gain = ch_g;
    o.lgain = log(ch_g);
    o.flag = flag;
    return o;
}

} // namespace prach
