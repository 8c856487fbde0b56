// Double-precision trigonometry of the orientation step, written for the arguments this path can produce.
//
// The reference calls libm: atan2 on the two first-order moments, then cos / sin of the float angle converted
// back to radians (src/orb.cpp:127-132, :172-174).  A general-purpose device libm spends most of its
// instructions on what cannot happen here (huge arguments, double-double argument reduction, denormals): about
// 450 instructions per wave for the three calls.  The functions below need about 130.
//
//  * atan2_deg_moments: the moments are integers below 2^22, so the quotient the arc tangent is taken of can be
//    reduced with exact integer arithmetic to |t| <= 1/8 and a 9-term series; the relative error of the result
//    is a few units in the last place.  The caller narrows the angle to float: `unsure` reports that the double
//    lies within 2^-43 (relative) of the midpoint of two neighbouring floats, the only case in which an error of
//    that size, or the (smaller) error of any libm, the reference's included, can decide the narrowing - about
//    one keypoint in 260 000.  The kernels use the series there too (a library call is no closer to the
//    reference's libm than the series is); the flag is for audits and tests.
//  * sincos_turn: sine and cosine of an angle in [0, 2 pi]: Cody-Waite reduction by pi/2 (exact first step,
//    the second carries its rounding error along as a tail) and the classic degree-13 / degree-14 minimax
//    kernels (Sun fdlibm's published coefficients) on [-pi/4, pi/4].  Error below one unit in the last place.
//
// Plain C++ with __builtin_fma, so that the host can run the same arithmetic (tests/test_trig_kernels.py builds
// this header with g++ -mfma and compares it with mpmath and with glibc).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ORB_TRIG_FN __host__ __device__ __forceinline__
#else
#define ORB_TRIG_FN static inline
#endif

namespace orbtrig {

constexpr double PI = 3.1415926535897932384626433832795;       // the literal of the reference's CV_PI
constexpr double PIO2 = 1.5707963267948966192313216916398;
constexpr double PIO2_LO = 6.123233995736766035868820147292e-17;     // pi/2 - fl64(pi/2)

// All coefficients in one table, read through a pointer the caller passes: a device caller hands in a pointer
// the optimiser cannot see through, so that the table is fetched (scalar loads) where it is used instead of
// being kept in registers across the whole kernel.
enum {
    K_ATAN_QUARTER = 0,     // atan(k / 4), k = 0 .. 4
    K_ATAN_SERIES = 5,      // 1/17, -1/15, 1/13, -1/11, 1/9, -1/7, 1/5, -1/3
    K_SIN = 13,             // S6 .. S1
    K_COS = 19,             // C6 .. C1
    K_COUNT = 25
};
#if defined(__HIP_DEVICE_COMPILE__)
__constant__
#else
static const
#endif
double COEF[K_COUNT] = {
    0.0, 0.24497866312686415417208248121128, 0.46364760900080611621425623146121,
    0.64350110879328438680280922871732, 0.78539816339744830961566084581988,
    1.0 / 17.0, -1.0 / 15.0, 1.0 / 13.0, -1.0 / 11.0, 1.0 / 9.0, -1.0 / 7.0, 1.0 / 5.0, -1.0 / 3.0,
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06,
    -1.98412698298579493134e-04, 8.33333333332248946124e-03, -1.66666666666666324348e-01,
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07,
    2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02};

// atan2(m01, m10) * (180 / pi) for integers |m| < 2^22, as a double before the narrowing to float.
// Straight-line code: every lane of a wave takes the same path.
// Q[k] = atan(k / 4), indexed per lane (a device caller keeps these five in LDS).
template <class Table, class QuarterTable>
ORB_TRIG_FN double atan2_deg_moments(int m01, int m10, bool& unsure, Table K, QuarterTable Q)
{
    const int ax = m10 < 0 ? -m10 : m10, ay = m01 < 0 ? -m01 : m01;
    const bool swap = ay > ax;
    const int mn = swap ? ax : ay;
    int mx = swap ? ay : ax;
    mx = mx > 0 ? mx : 1;                      // both moments zero: 0 / 1, angle 0 like atan2(0, 0)
    // nearest quarter k/4 to mn / mx, k = 0 .. 4 (integer compares: the same k on every machine)
    const int e = 8 * mn;
    const int k = (int)(e >= mx) + (int)(e >= 3 * mx) + (int)(e >= 5 * mx) + (int)(e >= 7 * mx);
    // atan(mn / mx) = atan(k / 4) + atan(t),  t = (4 mn - k mx) / (4 mx + k mn): integers, exact
    const double num = (double)(4 * mn - k * mx), den = (double)(4 * mx + k * mn);
    const double t = num / den;
    const double z = t * t;
    double p = K[K_ATAN_SERIES];
#pragma unroll
    for (int i = 1; i < 8; ++i) p = __builtin_fma(p, z, K[K_ATAN_SERIES + i]);
    const double a = Q[k] + __builtin_fma(t, z * p, t);
    // a -> pi/2 - a (moments swapped), then -> pi - a (m10 < 0), each with the constant's own rounding error
    // added back
    const double a1 = swap ? (PIO2 - a) + PIO2_LO : a;
    const double a2 = m10 < 0 ? (PI - a1) + 2.0 * PIO2_LO : a1;
    const double a3 = m01 < 0 ? -a2 : a2;
    const double deg = a3 * (180.0 / PI);
    uint64_t bits;
    __builtin_memcpy(&bits, &deg, 8);
    // the 29 bits the narrowing drops, within 1024 of the midpoint pattern 0x10000000
    unsure = ((((uint32_t)bits + 1024u) - 0x10000000u) & 0x1FFFFFFFu) <= 2048u;
    return deg;
}

template <class Table>
ORB_TRIG_FN void sincos_turn(double x, double& s, double& c, Table K)
{
    const double kd = __builtin_rint(x * 0.63661977236758134307553505349006);    // 2 / pi
    const double r0 = __builtin_fma(-kd, PIO2, x);              // exact: both are multiples of 2^-52 and |r0| < 1
    const double r = __builtin_fma(-kd, PIO2_LO, r0);
    const double rt = __builtin_fma(-kd, PIO2_LO, r0 - r);      // what the rounding of r lost
    const int k = (int)kd;
    const double z = r * r;
    double ps = K[K_SIN], pc = K[K_COS];
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        ps = __builtin_fma(ps, z, K[K_SIN + i]);
        pc = __builtin_fma(pc, z, K[K_COS + i]);
    }
    // cos(r + rt) = 1 - z/2 + z^2 pc - r rt; the rounding error of 1 - z/2 is recovered exactly
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + __builtin_fma(z * z, pc, -r * rt));
    // sin(r + rt) = r + r z ps + rt (1 - z/2)
    const double sr = r + __builtin_fma(r * z, ps, rt * w);
    const double a = (k & 1) ? cr : sr, b = (k & 1) ? sr : cr;
    s = (k & 2) ? -a : a;
    c = ((k + 1) & 2) ? -b : b;
}

}  // namespace orbtrig
