// mrt_math.h — the device math contract of the gfx950 path tracer.
//
// The reference calls libm through Rust's f32 methods (sin/cos/acos at src/rt.rs:997-1003,
// atan2 at src/rt.rs:522, powf at src/sampler.rs:88).  A GPU has no libm, and parity with a
// CPU checker at 1e-4 on stochastic paths needs every discrete decision (hit / miss, coin
// flips) to agree, so each transcendental is pinned to one exact sequence of IEEE-754
// binary32 / binary64 operations.  Everything here must be compiled with -ffp-contract=off,
// correctly rounded division and square root, and denormals preserved.
//
// MRT_HD code compiles for gfx950 (hipcc) and, for the CPU-side unit tests of the host
// logic only, for the host.  There is no CPU execution path in the product.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define MRT_HD __host__ __device__ inline __attribute__((always_inline))
#define MRT_HD_NOINLINE __host__ __device__
#else
#define MRT_HD inline __attribute__((always_inline))
#define MRT_HD_NOINLINE
#endif

namespace mrt {

typedef uint32_t u32;
typedef int32_t i32;

MRT_HD u32 f2u(float f) { return __builtin_bit_cast(u32, f); }
MRT_HD float u2f(u32 u) { return __builtin_bit_cast(float, u); }
MRT_HD float fabs_(float x) { return __builtin_fabsf(x); }
MRT_HD float floor_(float x) { return __builtin_floorf(x); }
MRT_HD float trunc_(float x) { return __builtin_truncf(x); }

// ---- correctly rounded sqrt, 1/x and a/b without the compiler's range scaffolding -------------------------------
// The reference's f32 sqrt / recip / division are IEEE correctly rounded, and so is everything here.  hipcc expands
// them (-fhip-fp32-correctly-rounded-divide-sqrt, denormals on) into a core sequence wrapped in scaling and fix-up
// code for denormal, huge, zero, infinite and NaN operands: 16 VALU + 4 s_nop for sqrt, v_div_scale x2 / v_div_fmas /
// v_div_fixup around the division core -- 30 and 25 issue slots (measured, profiles/microbench).  On operands inside
// the exponent window [2^-40, 2^40] that scaffolding is the identity (v_div_scale returns its operand unscaled with
// VCC = 0 -- ISA: exponent difference < 96, denominator / reciprocal / quotient normal, numerator exponent > 23 --,
// v_div_fmas is then a plain fma, v_div_fixup passes a normal quotient through; sqrt scales only below 2^-96), so the
// core sequences alone give THE SAME BITS there: 10 and 9 slots.  A wavefront takes the core when every active lane
// is inside the window and the compiler's full expansion otherwise -- one wave-uniform branch, no divergence.
// mrt_selftest_sweep compares the two on all 2^32 inputs (sqrt, recip) and on 10^10 operand pairs (divide).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MRT_GENERIC_IEEE)
#define MRT_FAST_IEEE 1
MRT_HD bool wave_all(bool ok) { return __builtin_amdgcn_ballot_w64(!ok) == 0ull; }
#else
MRT_HD bool wave_all(bool ok) { return ok; }
#endif
constexpr float kWinLo = 0x1p-40f, kWinHi = 0x1p+40f;
MRT_HD bool in_window(float x) { const float a = fabs_(x); return a >= kWinLo && a <= kWinHi; }       // false for NaN, 0, inf
MRT_HD bool in_window_pos(float x) { return x >= kWinLo && x <= kWinHi; }                               // false for x <= 0 too

#if defined(MRT_FAST_IEEE)
// v_sqrt_f32 is within 1 ulp; the correctly rounded root is s or a neighbour, picked by the sign of the exact
// residuals x - s_down * s and x - s_up * s (fma): the compiler's own selection, minus scaling and class checks.
MRT_HD float sqrt_core_(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = u2f(f2u(s) - 1u), su = u2f(f2u(s) + 1u);
    const float vd = __builtin_fmaf(-sd, s, x), vu = __builtin_fmaf(-su, s, x);
    float r = (0.0f >= vd) ? sd : s;
    r = (0.0f < vu) ? su : r;
    return r;
}
// Newton step on v_rcp_f32, then two residual corrections of the quotient: the compiler's division core
// (fma(-d, r, 1); r += e r; q = n r; q += fma(-d, q, n) r; q = fma(fma(-d, q, n), r, q)).
MRT_HD float rcp_refined_(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
MRT_HD float div_core_(float n, float d, float r)      // r = rcp_refined_(d)
{
    const float q0 = n * r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), r, q1);
}
#endif

MRT_HD float sqrt_(float x)                                     // IEEE correctly rounded
{
#if defined(MRT_FAST_IEEE)
    if (wave_all(in_window_pos(x))) return sqrt_core_(x);
#endif
    return __builtin_sqrtf(x);
}
MRT_HD float recip_(float x)                                    // f32::recip = 1.0 / x, correctly rounded
{
#if defined(MRT_FAST_IEEE)
    if (wave_all(in_window(x))) return div_core_(1.0f, x, rcp_refined_(x));
#endif
    return 1.0f / x;
}
MRT_HD float div_(float a, float b)                             // a / b, correctly rounded
{
#if defined(MRT_FAST_IEEE)
    if (wave_all(in_window(a) && in_window(b))) return div_core_(a, b, rcp_refined_(b));
#endif
    return a / b;
}
// two quotients over one denominator (the two roots of Sphere::intersect, src/rt.rs:350-351): one reciprocal
MRT_HD void div2_(float a0, float a1, float b, float &q0, float &q1)
{
#if defined(MRT_FAST_IEEE)
    if (wave_all(in_window(a0) && in_window(a1) && in_window(b))) {
        const float r = rcp_refined_(b);
        q0 = div_core_(a0, b, r);
        q1 = div_core_(a1, b, r);
        return;
    }
#endif
    q0 = a0 / b;
    q1 = a1 / b;
}
// 1 / sqrt(m) as the reference computes it (mag().recip(), src/lin.rs:60-66): two correctly rounded operations.
// m inside the window puts sqrt(m) inside it as well: one test covers both cores.
MRT_HD float recip_sqrt_(float m)
{
#if defined(MRT_FAST_IEEE)
    if (wave_all(in_window_pos(m))) {
        const float s = sqrt_core_(m);
        return div_core_(1.0f, s, rcp_refined_(s));
    }
#endif
    return 1.0f / __builtin_sqrtf(m);
}
// Approximate reciprocal (1 ulp, one instruction) for values that only steer conservative culling, never a result.
MRT_HD float rcp_fast(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
// Fused multiply-add, one rounding (fmaf): part of the math contract (version 2) in the transcendentals below -- argument
// reduction and polynomial steps -- never in the reference's own arithmetic (lin.rs / rt.rs operations stay mul + add).
MRT_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// a * b + c in one instruction where the hardware has it; like rcp_fast, only for culling arithmetic
MRT_HD float fma_fast(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaf(a, b, c);
#else
    return a * b + c;
#endif
}

constexpr float kPi = 3.14159274101257324f;    // std::f32::consts::PI
constexpr float kPiO2 = 1.57079637050628662f;
constexpr float kPiO4 = 0.785398185253143311f;
constexpr float kInf = __builtin_huge_valf();

MRT_HD float qnan() { return u2f(0x7fc00000u); }

// f32::max / f32::min (maxNum / minNum): a NaN operand yields the other operand; for a (+0, -0) pair
// max returns +0 and min returns -0.  That is exactly v_max_f32 / v_min_f32 on gfx950, which is what the
// builtins lower to in device code; the explicit form is the same function for host-side unit tests.
#if defined(__HIP_DEVICE_COMPILE__)
MRT_HD float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
MRT_HD float fmin_(float a, float b) { return __builtin_fminf(a, b); }
#else
MRT_HD float fmax_(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2u(a) & 0x80000000u) ? b : a;
    return a < b ? b : a;
}
MRT_HD float fmin_(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2u(a) & 0x80000000u) ? a : b;
    return b < a ? b : a;
}
#endif

// sin, cos for |x| <= 65536 (NaN beyond): quadrant reduction with a 3-term split of pi/2
// (8 + 11 + 24 significant bits), then degree-7 / degree-8 polynomials on [-pi/4, pi/4], Horner steps fused.
MRT_HD void sincos_(float x, float &s, float &c)
{
    if (!(fabs_(x) <= 65536.0f)) { s = qnan(); c = qnan(); return; }
    const float kf = floor_(fma_(x, 0.636619746685028076f, 0.5f));
    float r = fma_(-kf, 1.5703125f, x);
    r = fma_(-kf, 4.83751296997070312e-4f, r);
    r = fma_(-kf, 7.54978995489188216e-8f, r);
    const float z = r * r;
    const float sp = fma_(fma_(fma_(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float cp = fma_(fma_(fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                          fma_(-0.5f, z, 1.0f));
    const int q = (int)kf & 3;
    float ss = (q & 1) ? cp : sp;
    float cc = (q & 1) ? sp : cp;
    if (q & 2) ss = -ss;
    if ((q + 1) & 2) cc = -cc;
    s = ss;
    c = cc;
}

MRT_HD float asin_core_(float a)   // |a| <= 0.5
{
    const float z = a * a;
    const float p = fma_(fma_(fma_(fma_(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
    return fma_(p * z, a, a);
}

MRT_HD float acos_(float x)
{
    // The contract's three ranges (x > 0.5: 2 asin(sqrt((1-x)/2)); x < -0.5: pi - 2 asin(sqrt((1+x)/2)); else pi/2 - asin(x))
    // evaluated without branches, so a wavefront pays for one square root and one polynomial instead of all three
    // arms: (1 - |x|) is the same float as (1 - x) resp. (1 + x), so every lane gets the bits of its own arm.
    const float ax = fabs_(x);
    const bool big = ax > 0.5f;
    const float s = sqrt_(0.5f * (1.0f - ax));
    const float p = asin_core_(big ? s : x);
    const float two_p = 2.0f * p;
    const float r_big = (x > 0.0f) ? two_p : kPi - two_p;
    return big ? r_big : kPiO2 - p;           // NaN in, NaN out (the comparisons are false, p is NaN)
}

MRT_HD float atan_pos_(float t)    // t >= 0
{
    float y0, x;
    if (t > 2.414213562373095f) { y0 = kPiO2; x = -recip_(t); }
    else if (t > 0.4142135623730950f) { y0 = kPiO4; x = div_(t - 1.0f, t + 1.0f); }
    else { y0 = 0.0f; x = t; }
    const float z = x * x;
    const float p = fma_(fma_(fma_(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    return y0 + fma_(p * z, x, x);
}

MRT_HD float atan2_(float y, float x)
{
    if (x != x || y != y) return qnan();
    const float ay = fabs_(y), ax = fabs_(x);
    float a;
    if (ax == 0.0f) {
        if (ay == 0.0f) return 0.0f;
        a = kPiO2;
    } else if (ay == kInf && ax == kInf) {
        a = kPiO4;
    } else {
        a = atan_pos_(div_(ay, ax));
    }
    if (x < 0.0f) a = kPi - a;
    return (y < 0.0f) ? -a : a;
}

// powf for the tone map: binary64 log2 / exp2, one final rounding to binary32.
MRT_HD float pow_(float xf, float yf)
{
    if (yf == 0.0f) return 1.0f;
    if (xf != xf || yf != yf) return qnan();
    if (xf < 0.0f) return qnan();
    if (xf == 0.0f) return (yf > 0.0f) ? 0.0f : kInf;
    if (xf == kInf) return (yf > 0.0f) ? kInf : 0.0f;
    if (xf == 1.0f) return 1.0f;
    if (yf == kInf) return (xf > 1.0f) ? kInf : 0.0f;
    if (yf == -kInf) return (xf > 1.0f) ? 0.0f : kInf;

    const double x = (double)xf;
    const uint64_t bits = __builtin_bit_cast(uint64_t, x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __builtin_bit_cast(double, (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double r = (m - 1.0) / (m + 1.0);
    const double r2 = r * r;
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
    const double lnm = 2.0 * r * p;
    const double log2x = (double)e + lnm * 1.4426950408889634;
    const double t = (double)yf * log2x;
    if (t >= 129.0) return kInf;
    if (t <= -151.0) return 0.0f;
    const double kf = __builtin_floor(t + 0.5);
    const double f = (t - kf) * 0.6931471805599453;
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
    const int k = (int)kf;
    const double scale = __builtin_bit_cast(double, (uint64_t)(k + 1023) << 52);
    return (float)(q * scale);
}

// ---- RNG contract: counter hash keyed by (seed, pixel, sample, dimension) ----
MRT_HD u32 mix32(u32 x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
constexpr u32 kGold = 0x9E3779B9u;
MRT_HD u32 path_key(u32 seed_lo, u32 seed_hi, u32 pixel, u32 sample)
{
    return mix32((mix32(pixel + seed_lo) ^ seed_hi) + sample * kGold);
}
MRT_HD u32 draw_u32(u32 pk, u32 dim) { return mix32(pk + (dim + 1u) * kGold); }
MRT_HD float u32_to_unit(u32 u) { return (float)(u >> 9) * 1.1920928955078125e-7f; }   // 23-bit lattice in [0,1)

enum : u32 { DIM_LENS_X = 0, DIM_LENS_Z = 1, DIM_BOUNCE0 = 2, DIMS_PER_BOUNCE = 8 };
enum : u32 { SL_REFL_COIN = 0, SL_REFL_U1, SL_REFL_U2, SL_OPAC_COIN, SL_REFR_COIN, SL_REFR_U1, SL_REFR_U2, SL_EMIT_COIN };

// rand 0.8.5 Bernoulli::sample: p == 1 is always true; otherwise u < p * 2^32 (p in [0,1): exact product)
MRT_HD bool bernoulli(float p, u32 u)
{
    if (p == 1.0f) return true;
    return u < (u32)(p * 4294967296.0f);
}

// f32::total_cmp as a signed-integer key; every NaN orders as -NaN
MRT_HD i32 total_key(float t)
{
    if (t != t) return (i32)0x80000000u;
    i32 i = (i32)f2u(t);
    i ^= (i32)(((u32)(i >> 31)) >> 1);
    return i;
}

}  // namespace mrt
