/*
 * oracle_math.h — TEST INFRASTRUCTURE (CPU oracle).  Not part of the product; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build or load anything in oracle/.
 *
 * The "math contract": every transcendental the path needs, written as a fixed sequence of
 * IEEE-754 binary32 / binary64 operations (+ - * / sqrt floor and, where written OM_FMA, the fused
 * multiply-add; round-to-nearest-even, no implicit contraction, denormals kept).  The reference calls Rust's f32::{sin,cos,acos,atan2,powf}
 * (libm) at src/rt.rs:522,997-1003 and src/sampler.rs:88; libm is not available on the GPU,
 * so the contract pins one concrete f32 result per input that both this oracle and the HIP
 * kernel (micro_raytracer_amd/csrc/mrt_math.h, written independently against DESIGN.md §4)
 * must reproduce bit for bit.  Accuracy is ~1-2 ulp vs libm (tests/test_oracle_math.py).
 *
 * Compile with -ffp-contract=off and without -ffast-math.  OM_FMA is fmaf: one rounding, a hardware
 * instruction on every x86-64 since 2013 (-mfma) and on the GPU; the polynomial and argument-reduction steps of
 * the transcendentals use it (contract version 2: half the instructions of the mul + add form on the device,
 * and a slightly smaller error).  The reference's own arithmetic (src/lin.rs, src/rt.rs) is never fused.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t om_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float om_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t om_d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double om_u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

#define OM_PI        3.14159274101257324f   /* std::f32::consts::PI */
#define OM_PIO2      1.57079637050628662f
#define OM_PIO4      0.785398185253143311f
#define OM_QNAN      om_u2f(0x7fc00000u)
#define OM_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

/* quadrant reduction constants: pi/2 = P1 + P2 + P3 (+ 2^-60), P1 has 8 significant bits */
#define OM_TWO_OVER_PI 0.636619746685028076f
#define OM_P1 1.5703125f
#define OM_P2 4.83751296997070312e-4f
#define OM_P3 7.54978995489188216e-8f

/* sin and cos of a finite x with |x| <= 65536 (NaN otherwise).  Used on [0, 2*pi]. */
static inline void om_sincosf(float x, float *s, float *c)
{
    if (!(fabsf(x) <= 65536.0f)) { *s = OM_QNAN; *c = OM_QNAN; return; }
    float kf = floorf(OM_FMA(x, OM_TWO_OVER_PI, 0.5f));
    float r = OM_FMA(-kf, OM_P1, x);
    r = OM_FMA(-kf, OM_P2, r);
    r = OM_FMA(-kf, OM_P3, r);
    float z = r * r;
    float sp = OM_FMA(OM_FMA(OM_FMA(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    float cp = OM_FMA(OM_FMA(OM_FMA(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                      OM_FMA(-0.5f, z, 1.0f));
    int q = (int)kf & 3;
    float ss = (q & 1) ? cp : sp;
    float cc = (q & 1) ? sp : cp;
    if (q & 2) ss = -ss;
    if ((q + 1) & 2) cc = -cc;
    *s = ss;
    *c = cc;
}

static inline float om_sinf(float x) { float s, c; om_sincosf(x, &s, &c); return s; }
static inline float om_cosf(float x) { float s, c; om_sincosf(x, &s, &c); return c; }

/* asin on |a| <= 0.5 */
static inline float om_asin_core(float a)
{
    float z = a * a;
    float p = OM_FMA(OM_FMA(OM_FMA(OM_FMA(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z,
                     1.6666752422e-1f);
    return OM_FMA(p * z, a, a);
}

/* acos for x in [-1, 1]; NaN outside (through sqrt of a negative) */
static inline float om_acosf(float x)
{
    if (x > 0.5f) {
        float s = sqrtf(0.5f * (1.0f - x));
        return 2.0f * om_asin_core(s);
    }
    if (x < -0.5f) {
        float s = sqrtf(0.5f * (1.0f + x));
        return OM_PI - 2.0f * om_asin_core(s);
    }
    if (x != x) return OM_QNAN;
    return OM_PIO2 - om_asin_core(x);
}

/* atan for t >= 0 (finite or +inf) */
static inline float om_atan_pos(float t)
{
    float y0, x;
    if (t > 2.414213562373095f) {          /* tan(3pi/8) */
        y0 = OM_PIO2;
        x = -(1.0f / t);
    } else if (t > 0.4142135623730950f) {  /* tan(pi/8) */
        y0 = OM_PIO4;
        x = (t - 1.0f) / (t + 1.0f);
    } else {
        y0 = 0.0f;
        x = t;
    }
    float z = x * x;
    float p = OM_FMA(OM_FMA(OM_FMA(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    return y0 + OM_FMA(p * z, x, x);
}

/* atan2(y, x): result in [-pi, pi]; (0,0) -> 0 (sign of zero ignored: +/-0 are the same input) */
static inline float om_atan2f(float y, float x)
{
    if (x != x || y != y) return OM_QNAN;
    float ay = fabsf(y), ax = fabsf(x);
    float a;
    if (ax == 0.0f) {
        if (ay == 0.0f) return 0.0f;
        a = OM_PIO2;
    } else if (ay == INFINITY && ax == INFINITY) {
        a = OM_PIO4;
    } else {
        a = om_atan_pos(ay / ax);
    }
    if (x < 0.0f) a = OM_PI - a;
    return (y < 0.0f) ? -a : a;
}

/*
 * powf(x, y) for the tone map (src/sampler.rs:88, `v.powf(gamma)`), computed in binary64 and
 * rounded once to binary32 (error << 1 ulp, i.e. correctly rounded except ~1e-6 of inputs).
 *   x < 0 or NaN operand -> NaN ; x == 0 -> (y > 0 ? 0 : y == 0 ? 1 : inf) ; x == inf likewise.
 */
static inline float om_powf(float xf, float yf)
{
    if (yf == 0.0f) return 1.0f;
    if (xf != xf || yf != yf) return OM_QNAN;
    if (xf < 0.0f) return OM_QNAN;
    if (xf == 0.0f) return (yf > 0.0f) ? 0.0f : INFINITY;
    if (xf == INFINITY) return (yf > 0.0f) ? INFINITY : 0.0f;
    if (xf == 1.0f) return 1.0f;
    if (yf == INFINITY) return (xf > 1.0f) ? INFINITY : 0.0f;
    if (yf == -INFINITY) return (xf > 1.0f) ? 0.0f : INFINITY;

    double x = (double)xf;
    uint64_t bits = om_d2u(x);               /* x is a normal double (every f32 incl. denormals is) */
    int64_t e = (int64_t)((bits >> 52) & 0x7ff) - 1023;
    uint64_t mb = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = om_u2d(mb);                   /* [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }   /* [0.7071, 1.4142] */
    double r = (m - 1.0) / (m + 1.0);
    double r2 = r * r;
    /* ln(m) = 2 r (1 + r2/3 + r2^2/5 + ... ), |r| <= 0.1716 */
    double p = 1.0 / 23.0;
    p = p * r2 + 1.0 / 21.0;
    p = p * r2 + 1.0 / 19.0;
    p = p * r2 + 1.0 / 17.0;
    p = p * r2 + 1.0 / 15.0;
    p = p * r2 + 1.0 / 13.0;
    p = p * r2 + 1.0 / 11.0;
    p = p * r2 + 1.0 / 9.0;
    p = p * r2 + 1.0 / 7.0;
    p = p * r2 + 1.0 / 5.0;
    p = p * r2 + 1.0 / 3.0;
    p = p * r2 + 1.0;
    double lnm = 2.0 * r * p;
    double log2x = (double)e + lnm * 1.4426950408889634;    /* 1/ln 2 */
    double t = (double)yf * log2x;
    if (t >= 129.0) return INFINITY;
    if (t <= -151.0) return 0.0f;
    double kf = floor(t + 0.5);
    double f = (t - kf) * 0.6931471805599453;                /* ln 2, |f| <= 0.3466 */
    /* exp(f) Taylor, degree 13 */
    double q = 1.0 / 6227020800.0;
    q = q * f + 1.0 / 479001600.0;
    q = q * f + 1.0 / 39916800.0;
    q = q * f + 1.0 / 3628800.0;
    q = q * f + 1.0 / 362880.0;
    q = q * f + 1.0 / 40320.0;
    q = q * f + 1.0 / 5040.0;
    q = q * f + 1.0 / 720.0;
    q = q * f + 1.0 / 120.0;
    q = q * f + 1.0 / 24.0;
    q = q * f + 1.0 / 6.0;
    q = q * f + 0.5;
    q = q * f + 1.0;
    q = q * f + 1.0;
    int64_t k = (int64_t)kf;                                  /* -151 .. 129 */
    double scale = om_u2d((uint64_t)(k + 1023) << 52);        /* 2^k, normal double */
    return (float)(q * scale);
}

#endif
