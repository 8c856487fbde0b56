// Host build of csrc/orb_trig.h for tests/test_trig_kernels.py and scripts/trig_exhaustive.py (test
// infrastructure: the product only ever runs these functions on the device).
//   g++ -O2 -mfma -ffp-contract=off -shared -fPIC -o build/libtrig_host.so tests/trig_host.cpp -lm
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../visual_slam_amd/csrc/orb_trig.h"

extern "C" {

void trig_atan2_deg(const int32_t* m01, const int32_t* m10, long n, double* deg, uint8_t* unsure)
{
    for (long i = 0; i < n; ++i) {
        bool u;
        deg[i] = orbtrig::atan2_deg_moments(m01[i], m10[i], u, orbtrig::COEF, orbtrig::COEF + orbtrig::K_ATAN_QUARTER);
        unsure[i] = u;
    }
}

void trig_sincos(const double* x, long n, double* s, double* c)
{
    for (long i = 0; i < n; ++i) orbtrig::sincos_turn(x[i], s[i], c[i], orbtrig::COEF);
}

// Every float angle in [lo_bits, hi_bits) (bit patterns of non-negative floats), converted to radians as the
// reference does: how often, and by how much, do sine / cosine differ from this host's libm?
// out[0] = angles, out[1] = sines that differ, out[2] = cosines that differ, out[3] = largest difference in
// units of the last place (either function).
void trig_sincos_vs_libm(uint32_t lo_bits, uint32_t hi_bits, long* out)
{
    long n = 0, ds = 0, dc = 0, worst = 0;
    for (uint32_t b = lo_bits; b < hi_bits; ++b) {
        float a;
        memcpy(&a, &b, 4);
        const double rad = (double)a * (orbtrig::PI / 180.0);
        double s, c;
        orbtrig::sincos_turn(rad, s, c, orbtrig::COEF);
        const double ls = sin(rad), lc = cos(rad);
        int64_t is, ils, ic, ilc;
        memcpy(&is, &s, 8); memcpy(&ils, &ls, 8); memcpy(&ic, &c, 8); memcpy(&ilc, &lc, 8);
        long es = (long)llabs(is - ils), ec = (long)llabs(ic - ilc);
        if ((s < 0) != (ls < 0)) es = 1L << 40;
        if ((c < 0) != (lc < 0)) ec = 1L << 40;
        ++n; ds += es != 0; dc += ec != 0;
        if (es > worst) worst = es;
        if (ec > worst) worst = ec;
    }
    out[0] = n; out[1] = ds; out[2] = dc; out[3] = worst;
}

// The float angle the orientation step stores, from this header's arc tangent and from this host's libm: both as
// float bit patterns, with the header's `unsure` flag.
void trig_angle_floats(const int32_t* m01, const int32_t* m10, long n, uint32_t* mine, uint32_t* libm, uint8_t* unsure)
{
    for (long i = 0; i < n; ++i) {
        bool u;
        double deg = orbtrig::atan2_deg_moments(m01[i], m10[i], u, orbtrig::COEF, orbtrig::COEF + orbtrig::K_ATAN_QUARTER);
        const double ref = atan2((double)m01[i], (double)m10[i]) * (180.0 / orbtrig::PI);
        float a = (float)deg, r = (float)ref;
        if (a < 0) a = (float)((double)a + 360.0);
        if (r < 0) r = (float)((double)r + 360.0);
        memcpy(&mine[i], &a, 4); memcpy(&libm[i], &r, 4);
        unsure[i] = u;
    }
}

}
