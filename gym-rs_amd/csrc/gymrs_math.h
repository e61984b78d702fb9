// gymrs_math.h — deterministic f32 sin/cos shared by the HIP kernels and the host.
//
// Why not sinf/cosf: the device (OCML) and host (glibc) libm differ in their last bit, and the
// parity bar for this path is bit-exact integer step/done counts between the GPU kernel and its
// CPU f32 twin (BASELINE.json north_star).  This header uses only IEEE-exact operations
// (+, *, fma, rint, int<->float conversions, integer arithmetic) so that the SAME bits come out of
// gfx950 and x86.  Compile both sides with -ffp-contract=off: every fma below is explicit.
//
// Replaces the f64 libm calls of the reference hot path:
//   theta.cos()/theta.sin()   /root/reference/src/envs/classical_control/cartpole.rs:420-421
//   (3*position).cos()        /root/reference/src/envs/classical_control/mountain_car.rs:412
//
// Accuracy (tests/test_twin_vs_oracle.py against f64 libm): <= 1.6 ulp for |x| <= pi/4 and for |x| > 200,
// <= 1.5e-7 absolute for pi/4 < |x| <= 200 (f32 reduction): far inside the 1e-6 tolerance vs the f64 oracle.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GYMRS_HD __host__ __device__ __forceinline__
#else
#define GYMRS_HD inline
#endif

namespace gymrs {

GYMRS_HD float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
GYMRS_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
GYMRS_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
GYMRS_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
GYMRS_HD float fabsf_(float f) { return u2f(f2u(f) & 0x7fffffffu); }

// |r| <= pi/4 (+2%).  Near-minimax fits (least squares on Chebyshev nodes in z = r*r):
//   sin r = r + r*z*(S1 + z*(S2 + z*S3))          approximation error 1.3e-8 relative
//   cos r = 1 - z/2 + z*z*(C1 + z*(C2 + z*C3))    approximation error 8.4e-9 absolute
GYMRS_HD void sincos_poly(float r, float* s, float* c)
{
    const float S1 = -0.166666641831398f, S2 = 0.008332724682986736f, S3 = -0.0001958291686605662f;
    const float C1 = 0.04166654869914055f, C2 = -0.0013880875194445252f, C3 = 2.360151120228693e-05f;
    float z = r * r;
    float ps = fmaf_(z, S3, S2);
    ps = fmaf_(z, ps, S1);
    float pc = fmaf_(z, C3, C2);
    pc = fmaf_(z, pc, C1);
    float rz = r * z;
    float w = z * z;
    *s = fmaf_(rz, ps, r);
    *c = fmaf_(w, pc, fmaf_(z, -0.5f, 1.0f));
}

// f32 Cody-Waite reduction for |x| <= 200 (|k| <= 128): r = x - k*(P1 + P2), both steps one fma.  The first
// fma forms x - k*P1 exactly and rounds once (|r| <= 0.8: error <= 3e-8 however large x is), the second adds
// the k*P2 correction; the absolute error of r is <= 6e-8, so sin/cos come out within 1.5e-7 ABSOLUTE (not
// within ulps near their zeros) - an order of magnitude inside the 1e-6 budget, at a third of the cost of
// the f64 reduction below.  MountainCar's cos(3*position) and Pendulum's angles live here.
GYMRS_HD int rem_pio2f_short(float x, float* r_out)
{
    const float two_over_pi = 0x1.45f306p-1f, P1 = 0x1.921fb6p+0f, P2 = -0x1.777a5cp-25f;
    const float fn = __builtin_rintf(x * two_over_pi);
    float r = fmaf_(-fn, P1, x);
    r = fmaf_(-fn, P2, r);
    *r_out = r;
    return (int)fn;
}
GYMRS_HD bool in_short_range(float x) { return (__builtin_bit_cast(uint32_t, x) & 0x7fffffffu) <= 0x43480000u; } // |x| <= 200

// Argument reduction x = k*(pi/2) + r, |r| <= pi/4.  Returns k (mod 4 is all that matters).
//   |x| <= pi/4            : k = 0, r = x exactly (so callers may skip the reduction wave-wide)
//   |x| <= 200             : f32 Cody-Waite (rem_pio2f_short)
//   |x| <  2^28 * pi/2     : two-constant Cody-Waite in f64 with fma
//   otherwise (finite)     : Payne-Hanek on a 224-bit table of 2/pi, integer arithmetic
//   inf / NaN              : r = NaN
GYMRS_HD int rem_pio2f(float x, float* r_out)
{
    const uint32_t ux = f2u(x);
    const uint32_t ax = ux & 0x7fffffffu;
    if (ax <= 0x3f490fdbu) { // |x| <= fl32(pi/4)
        *r_out = x;
        return 0;
    }
    if (ax <= 0x43480000u) return rem_pio2f_short(x, r_out); // |x| <= 200
    if (ax < 0x4dc90fdbu) { // |x| < 2^28 * pi/2
        const double invpio2 = 0x1.45f306dc9c883p-1;
        const double pio2_hi = 0x1.921fb54442d18p+0;
        const double pio2_lo = 0x1.1a62633145c07p-54;
        double xd = (double)x;
        double fn = __builtin_rint(xd * invpio2);
        double rd = fma_(-fn, pio2_hi, xd);
        rd = fma_(-fn, pio2_lo, rd);
        *r_out = (float)rd;
        return (int)fn;
    }
    if (ax >= 0x7f800000u) { // inf or NaN
        *r_out = u2f(0x7fc00000u);
        return 0;
    }
    // Payne-Hanek.  |x| = m * 2^e with m a 24-bit integer, e >= 5.
    // 2/pi = 0.a2f9836e 4e441529 fc2757d1 f534ddc0 db629599 3c439041 fe5163ab (hex).  The words are
    // picked with selects, not from a table in memory: a table load would put a vmcnt(0) wait into
    // the kernels' hot loop even though this branch is almost never taken.
    const uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
    const int e = (int)(ax >> 23) - 150;
    // Bits of 2/pi above position s contribute multiples of 4 to x*(2/pi): skip them.
    const int s = e - 2;            // 3 <= s <= 102
    const int i = s >> 5, sh = s & 31; // i in 0..3
    const uint32_t t0 = i == 0 ? 0xa2f9836eu : i == 1 ? 0x4e441529u : i == 2 ? 0xfc2757d1u : 0xf534ddc0u;
    const uint32_t t1 = i == 0 ? 0x4e441529u : i == 1 ? 0xfc2757d1u : i == 2 ? 0xf534ddc0u : 0xdb629599u;
    const uint32_t t2 = i == 0 ? 0xfc2757d1u : i == 1 ? 0xf534ddc0u : i == 2 ? 0xdb629599u : 0x3c439041u;
    const uint32_t t3 = i == 0 ? 0xf534ddc0u : i == 1 ? 0xdb629599u : i == 2 ? 0x3c439041u : 0xfe5163abu;
    uint32_t w[3];
    w[0] = (uint32_t)((((uint64_t)t0 << 32) | t1) >> (32 - sh));
    w[1] = (uint32_t)((((uint64_t)t1 << 32) | t2) >> (32 - sh));
    w[2] = (uint32_t)((((uint64_t)t2 << 32) | t3) >> (32 - sh));
    // P = (m * W) mod 2^96, W = w0:w1:w2 ; x*(2/pi) mod 4 = P / 2^94
    uint64_t p2 = (uint64_t)m * w[2];
    uint64_t p1 = (uint64_t)m * w[1] + (p2 >> 32);
    uint32_t lo = (uint32_t)p2;
    uint32_t mid = (uint32_t)p1;
    uint32_t hi = (uint32_t)((uint64_t)m * w[0] + (p1 >> 32));
    uint32_t q = hi >> 30;
    uint64_t top = ((uint64_t)hi << 34) | ((uint64_t)mid << 2) | (uint64_t)(lo >> 30); // fraction * 2^64
    q += (uint32_t)(top >> 63);                                                         // round to nearest
    double f = (double)(int64_t)top * 0x1p-64;                                          // in [-0.5, 0.5)
    double rd = f * 0x1.921fb54442d18p+0;
    int k = (int)(q & 3u);
    if (ux >> 31) {
        rd = -rd;
        k = -k;
    }
    *r_out = (float)rd;
    return k;
}

GYMRS_HD void sincos_quadrant(int k, float sr, float cr, float* s, float* c)
{
    // k mod 4: 0:(s,c) 1:(c,-s) 2:(-s,-c) 3:(-c,s)
    float a = (k & 1) ? cr : sr;
    float b = (k & 1) ? sr : cr;
    uint32_t sa = (uint32_t)(k & 2) << 30;       // negate sin for k = 2,3
    uint32_t sb = (uint32_t)((k + 1) & 2) << 30; // negate cos for k = 1,2
    *s = u2f(f2u(a) ^ sa);
    *c = u2f(f2u(b) ^ sb);
}

// Branch-free sin & cos for |x| < 2^28 * pi/2 (the Cody-Waite range of rem_pio2f).  Produces the
// same bits as sincosf_ on that range (for |x| <= pi/4: fn = 0, r = x exactly), so a kernel may use
// it for a whole wave whenever every lane is in range, without changing any result.
GYMRS_HD void sincos_medium(float x, float* s, float* c)
{
    const double invpio2 = 0x1.45f306dc9c883p-1;
    const double pio2_hi = 0x1.921fb54442d18p+0;
    const double pio2_lo = 0x1.1a62633145c07p-54;
    const double xd = (double)x;
    const double fn = __builtin_rint(xd * invpio2);
    double rd = fma_(-fn, pio2_hi, xd);
    rd = fma_(-fn, pio2_lo, rd);
    float sr, cr;
    sincos_poly((float)rd, &sr, &cr);
    sincos_quadrant((int)fn, sr, cr, s, c);
}
// Branch-free sin & cos for |x| <= 200; same bits as sincosf_ there (for |x| <= pi/4: fn = 0, r = x exactly).
GYMRS_HD void sincos_short(float x, float* s, float* c)
{
    float r;
    const int k = rem_pio2f_short(x, &r);
    float sr, cr;
    sincos_poly(r, &sr, &cr);
    sincos_quadrant(k, sr, cr, s, c);
}
GYMRS_HD bool in_small_range(float x) { return (f2u(x) & 0x7fffffffu) <= 0x3f490fdbu; }  // |x| <= fl32(pi/4)
GYMRS_HD bool in_medium_range(float x) { return (f2u(x) & 0x7fffffffu) < 0x4dc90fdbu; }  // |x| < 2^28*pi/2

// How a caller wants sin/cos evaluated.  All three give identical bits where their domains overlap.
struct SinCosGeneral { static GYMRS_HD void eval(float x, float* s, float* c); };
struct SinCosSmall { static GYMRS_HD void eval(float x, float* s, float* c) { sincos_poly(x, s, c); } };
struct SinCosMedium { static GYMRS_HD void eval(float x, float* s, float* c) { sincos_medium(x, s, c); } };
struct SinCosShort { static GYMRS_HD void eval(float x, float* s, float* c) { sincos_short(x, s, c); } };

// Full-range sin & cos.
GYMRS_HD void sincosf_(float x, float* s, float* c)
{
    if ((f2u(x) & 0x7fffffffu) <= 0x3f490fdbu) { // |x| <= pi/4: no reduction, no quadrant fix-up
        sincos_poly(x, s, c);                    // (bit-identical to the general path with k = 0)
        return;
    }
    float r;
    int k = rem_pio2f(x, &r);
    float sr, cr;
    sincos_poly(r, &sr, &cr);
    sincos_quadrant(k, sr, cr, s, c);
}

GYMRS_HD void SinCosGeneral::eval(float x, float* s, float* c) { sincosf_(x, s, c); }

GYMRS_HD float cosf_(float x)
{
    float s, c;
    sincosf_(x, &s, &c);
    return c;
}

GYMRS_HD float sinf_(float x)
{
    float s, c;
    sincosf_(x, &s, &c);
    return s;
}

} // namespace gymrs
