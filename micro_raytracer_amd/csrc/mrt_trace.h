// mrt_trace.h — the per-lane path tracer of the gfx950 megakernel.
//
// One lane owns one supersampled pixel and walks all samples of a launch with path
// regeneration: as soon as a path ends the lane starts its next sample, so a wavefront only
// idles at the very end of the launch.  The scene is read from the LDS-staged blob
// (mrt_scene.h); the traversal loop over renderers x instances is wave-uniform, so those
// reads are LDS broadcasts and all divergence is in per-lane predicates.
//
// Behaviour follows the reference function by function (file:line in each comment, paths
// relative to the reference checkout).  Geometry (everything that feeds a hit / miss or a
// coin flip) keeps the reference's f32 operation order exactly; radiance is accumulated
// front-to-back instead of the reference's back-to-front fold (src/rt.rs:964-993), which is
// the same affine recurrence evaluated in the other direction.
#pragma once
#include "mrt_scene.h"

namespace mrt {

// Scene features a kernel instantiation is compiled for; the host picks the smallest set that covers the
// scene, so e.g. a Cornell box of planes and spheres carries no octree walker, texture fetch or shadow-ray code.
enum : u32 {
    F_BOX = 1u,       // boxes or meshes present: rays need the patched reciprocal direction (src/rt.rs:303-316)
    F_TRI = 2u,       // triangle / mesh renderers present
    F_MAPS = 4u,      // some material has a texture map
    F_LIGHTS = 8u,    // the scene has lights (shadow rays + direct term)
    F_ALL = 15u,
    F_BVH = 16u,      // many instances: a BVH over them replaces most of the linear scan (only built with F_ALL)
    F_NOSTASH = 32u,  // launch-shape marker, not a scene feature: 1024-thread workgroup whose scene leaves no LDS for the lane stash
    F_COLD = 64u,     // launch-shape marker: texels are read from global memory, not staged in LDS (mesh kernels: with a per-lane walk area)
                      // (Params.lds_words_warm)
    F_DEEP = 128u,    // with F_COLD, meshes beyond the LDS: triangles stay in global memory too (Params.lds_words_hot) and of the
                      // (level-ordered) triangle-BVH table only the first Params.n_tbvh_hot nodes -- the top levels of every
                      // tree -- are staged
    F_IDENT = 256u    // EVERY instance of the scene is untransformed (default `dir`: both matrices the identity as values): the
                      // per-instance identity test of the tag and the transform's address are compiled out of the linear scan --
                      // ~30 of ~220 cycles per instance on the Cornell box (8236 -> 8700 Msamples/s).  Exists for the plain
                      // 256-thread kernels of planes / spheres / boxes with and without lights; rays whose shifted origin has a
                      // zero, infinite or NaN component still take the reference's two mat-vecs (xf_vec)
};
constexpr u32 plain_feat(u32 feat) { return feat & ~(u32)F_IDENT; }
// words of the packed scene a kernel instantiation stages in LDS
MRT_HD u32 staged_words_for(const Params &P, u32 feat) { return (feat & F_DEEP) ? P.lds_words_hot : ((feat & F_COLD) ? P.lds_words_warm : P.lds_words); }
// Mesh kernels that leave the cold tables out of LDS spend it on a per-lane WALK AREA (behind the lane stash), Params.walk_cap
// entries per lane: the leaf queue of the binary walk (kLeafQueue entries), or (F_DEEP) node stack + leaf queue of the 4-wide walk.
constexpr bool has_walk_area(u32 feat) { return (feat & F_TRI) && (feat & F_BOX) && (feat & F_COLD); }
#ifndef MRT_UNIFORM_TAG              // 0: instance tags stay per-lane values in the linear scans (experiment knob)
#define MRT_UNIFORM_TAG 1
#endif
#ifndef MRT_T0_FROM_KEY              // 0: the closest hit's t0 is selected per candidate like its other fields (experiment knob)
#define MRT_T0_FROM_KEY 1
#endif
#ifndef MRT_NZFIN_PRODUCT            // 0: three class checks per shifted origin instead of one on the product (experiment knob)
#define MRT_NZFIN_PRODUCT 1
#endif
#ifndef MRT_SHADOW_QUEUE            // 1: shadow walks of kernels with a walk area postpone every leaf too (experiment: the x86 round
                                    // model says -12 % box steps per wavefront, the GPU 3669 against 3721 Msamples/s: off)
#define MRT_SHADOW_QUEUE 0
#endif
#ifndef MRT_LEAF_QUEUE              // build-time experiment knob (make EXTRA=-D...)
#define MRT_LEAF_QUEUE 8u
#endif
constexpr u32 kLeafQueue = MRT_LEAF_QUEUE;

// Divergence probe: only the x86 build of tests/emu defines MRT_PROBE(phase); in the kernel it is nothing.
#ifndef MRT_PROBE
#define MRT_PROBE(phase)
#endif
// Work counters, likewise only defined by tests/emu/probe.cpp.
#ifndef MRT_COUNT
#define MRT_COUNT(counter)
#endif
#ifndef MRT_PROBE_INST                 // flat instance index about to be tested (tests/emu/probe2.cpp)
#define MRT_PROBE_INST(i)
#endif
// Phase timing (debug builds with -DMRT_PHASE_TIMING only: profiles/README.md): shader-clock ticks a wavefront spends between
// the marks of the main loop, summed per phase; the marks sit where every live lane of the wavefront passes.
#if defined(MRT_PHASE_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define MRT_TICK(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); phase_ticks[slot] += now_ - phase_last; phase_last = now_; } while (0)
#define MRT_TICK_DECL unsigned long long phase_ticks[4] = {0ull, 0ull, 0ull, 0ull}; unsigned long long phase_last = __builtin_amdgcn_s_memtime()
#define MRT_TICK_OUT(dst) do { for (int k_ = 0; k_ < 4; ++k_) (dst)[k_] = phase_ticks[k_]; } while (0)
#else
#define MRT_TICK(slot)
#define MRT_TICK_DECL
#define MRT_TICK_OUT(dst)
#endif
#ifndef MRT_PROBE_FALLBACK             // a mesh query that goes to the reference's walk: the ray (tests/emu/probe.cpp prints it)
#define MRT_PROBE_FALLBACK(ro, rd, dd)
#endif
#ifndef MRT_PROBE_ROUND                // one round of a lane's triangle-BVH walk: box steps taken, triangles tested, membership boxes
#define MRT_PROBE_ROUND(steps, tris, membs)
#endif
enum : u32 { CT_TRACE = 0, CT_LIN_TEST, CT_BVH_NODE, CT_BVH_TEST, CT_MESH_CALL, CT_MESH_ROOT_HIT, CT_TBVH_NODE, CT_TBVH_TRI, CT_TBVH_TRI_HIT, CT_MEMB_BOX, CT_TRACE_ANY, CT_WALK_OVERFLOW, CT_WALK_ROUND, CT_REF_TRI, CT_REF_BOX, CT_COUNT };
enum : u32 { PH_ITER = 0, PH_REGEN, PH_SPHERE_MATH, PH_PLANE_HIT, PH_SHADE, PH_NORMAL_NONPLANE, PH_SCATTER1, PH_SCATTER2, PH_REFRACT, PH_EMIT_END, PH_LIGHTS, PH_COUNT };

constexpr float kE = 0.0001f;                 // src/rt.rs:7
constexpr float kBig = 1.0f / 0.0001f;        // E.recip(), src/rt.rs:307

struct V3 { float x, y, z; };
MRT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
MRT_HD V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }                 // lin.rs:211-221
MRT_HD V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }                 // lin.rs:247-257
MRT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                // lin.rs:259-264
MRT_HD V3 muls(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }                   // lin.rs:266-275
MRT_HD V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }                                      // lin.rs:304-314
MRT_HD V3 hadam(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }               // lin.rs:107-113
MRT_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); } // lin.rs:52-58
MRT_HD float mag(V3 a) { return sqrt_(a.x * a.x + a.y * a.y + a.z * a.z); }               // lin.rs:60-62
MRT_HD V3 norm(V3 a) { return muls(a, recip_sqrt_(a.x * a.x + a.y * a.y + a.z * a.z)); }   // lin.rs:64-66: self * mag().recip()
MRT_HD V3 reflect(V3 d, V3 n) { return sub(d, muls(n, 2.0f * dot(d, n))); }               // lin.rs:68-70
MRT_HD V3 m3mul(const float *m, V3 v)                                                     // lin.rs:344-365
{
    return v3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
MRT_HD V3 ld3(const float *F, u32 i) { return v3(F[i], F[i + 1], F[i + 2]); }
struct alignas(16) F4 { float x, y, z, w; };
MRT_HD F4 ld4(const float *F, u32 i) { return *reinterpret_cast<const F4 *>(F + i); }   // i % 4 == 0 (records are 16-byte aligned)
MRT_HD u32 ldu(const float *F, u32 i) { return f2u(F[i]); }

// true when x is neither zero, infinite nor NaN
MRT_HD bool nzfin(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_classf(x, 0x198);   // -normal | -denormal | +denormal | +normal
#else
    const float a = fabs_(x);
    return a > 0.0f && a < kInf;
#endif
}
MRT_HD bool nzfin3(V3 v) { return nzfin(v.x) && nzfin(v.y) && nzfin(v.z); }
// a SUFFICIENT test with one class check: a finite non-zero product has three finite non-zero factors (a zero factor gives 0 or
// NaN, an infinite or NaN one inf or NaN); products that under- or overflow (|x y z| outside 1e-45 .. 3e38) answer false
MRT_HD bool nzfin3_product(V3 v) { return MRT_NZFIN_PRODUCT ? nzfin((v.x * v.y) * v.z) : nzfin3(v); }
// 24-bit multiply (v_mul_u32_u24: full rate, where v_mul_lo_u32 is not); both factors below 2^24
MRT_HD u32 mul24(u32 a, u32 b)
{
    return (a & 0xffffffu) * (b & 0xffffffu);      // the masks tell the compiler what it needs to pick the 24-bit form
}
// nothing is scheduled across this point (device): used to keep loads that feed the END of a loop body at its start
MRT_HD void sched_fence()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// Values the compiler must take as they are (device): a word it would otherwise trace back to a float load and compare by
// class (a mask constant re-loaded every loop trip), a loop constant it would fold to a literal that VOP3 cannot encode (a
// v_mov per trip).  Empty asm, no instruction.
MRT_HD u32 opaque_v(u32 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}
MRT_HD u32 opaque_s(u32 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(x));
#endif
    return x;
}
// index of the lowest set bit of a non-zero word (v_ffbl_b32)
MRT_HD u32 lowest_bit(u32 x) { return (u32)__builtin_ctz(x | 0x80000000u); }

// rot_y * (look * v), src/rt.rs:730-731, 782, 792, 798.  When both matrices equal the identity as
// values (default instance direction) and no component of v is zero / non-finite the two products
// return v bit for bit, so they are skipped; signed zeros and NaNs take the full route.
MRT_HD V3 xf_full(const float *X, V3 v) { return m3mul(X + XF_R, m3mul(X + XF_L, v)); }
MRT_HD V3 xf_vec(const float *X, bool ident, V3 v)
{
    if (ident && nzfin3_product(v)) return v;      // (false only sends the vector through the reference's two mat-vecs)
    return xf_full(X, v);
}

MRT_HD V3 recip_patched(V3 d)   // Box::intersect's 1/dir with inf -> 1/E, src/rt.rs:303-316
{
    V3 m;
#if defined(MRT_FAST_IEEE)
    if (wave_all(in_window(d.x) && in_window(d.y) && in_window(d.z))) {      // one test for the three reciprocals (mrt_math.h)
        m = v3(div_core_(1.0f, d.x, rcp_refined_(d.x)), div_core_(1.0f, d.y, rcp_refined_(d.y)), div_core_(1.0f, d.z, rcp_refined_(d.z)));
        return m;                                                            // finite: nothing to patch
    }
#endif
    m = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    if (fabs_(m.x) == kInf) m.x = kBig;
    if (fabs_(m.y) == kInf) m.y = kBig;
    if (fabs_(m.z) == kInf) m.z = kBig;
    return m;
}

// Box::intersect, src/rt.rs:299-333, with m = recip_patched(dir) and half = 0.5 * sizes
MRT_HD bool box_isect(V3 half, V3 ro, V3 m, V3 pos, float &t0, float &t1)
{
    const V3 n = hadam(sub(ro, pos), m);
    const V3 k = hadam(half, v3(fabs_(m.x), fabs_(m.y), fabs_(m.z)));
    const V3 a = sub(neg(n), k);
    const V3 b = add(neg(n), k);
    t0 = fmax_(fmax_(a.x, a.y), a.z);
    t1 = fmin_(fmin_(b.x, b.y), b.z);
    return !(t0 > t1 || t1 < 0.0f);
}

// Sphere::intersect, src/rt.rs:335-359; oo = ray.orig - pos, a = dir.dir
MRT_HD bool sphere_isect(float r2, V3 oo, V3 rd, float a, float &t0, float &t1)
{
    const float b = 2.0f * dot(oo, rd);
    const float c = dot(oo, oo) - r2;
    const float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return false;
    // Moving away from the centre (b > 0): t0 = (-b - sqrt(disc)) / (2a) is negative, i.e. the `t0 < 0` miss of
    // src/rt.rs:353, without evaluating the sqrt and the divisions.  The guards keep this exact: disc >= 0 excludes
    // NaN, and with b > 1e-20 and 0 < a < 1e18 the quotient is a non-zero negative number (no underflow to -0,
    // which the reference would accept as a hit at t0 = -0).
    if (b > 1e-20f && a > 0.0f && a < 1e18f && disc >= 0.0f) return false;
    MRT_PROBE(PH_SPHERE_MATH);
    const float sq = sqrt_(disc);
    float q0, q1;
    div2_(-b - sq, -b + sq, 2.0f * a, q0, q1);
    if (q0 < 0.0f) return false;
    t0 = q0;
    t1 = q1;
    return true;
}

// Triangle::intersect, src/rt.rs:361-398; vp = v0 + pos
MRT_HD bool tri_isect(V3 vp, V3 e0, V3 e1, V3 ro, V3 rd, float &t)
{
    const V3 p = cross(rd, e1);
    const float d = dot(e0, p);
    if (d < kE && d > -kE) return false;
    const float inv_d = recip_(d);
    const V3 tv = sub(ro, vp);
    const float u = dot(tv, p) * inv_d;
    if (u < 0.0f || u > 1.0f) return false;
    const V3 q = cross(tv, e0);
    const float v = dot(rd, q) * inv_d;
    if (v < 0.0f || (u + v) > 1.0f) return false;
    const float tt = dot(e1, q) * inv_d;
    if (tt < 0.0f) return false;
    t = tt;
    return true;
}

// Plane::intersect, src/rt.rs:400-412; nn = norm(n), d = (-nn).pos
MRT_HD bool plane_isect(V3 nn, float d, V3 ro, V3 rd, float &t)
{
    const float tt = div_(-(dot(ro, nn) + d), dot(rd, nn));
    if (tt <= 0.0f) return false;
    t = tt;
    return true;
}

MRT_HD bool in_range(float lo, float hi, float x) { return lo <= x && x < hi; }

// Normal for Box, src/rt.rs:414-445 (x / y else-if chain, then an independent z chain that overrides)
MRT_HD V3 box_normal(V3 inv2, V3 hit, V3 pos)
{
    const V3 p = hadam(sub(hit, pos), inv2);
    const float plo = 1.0f - kE, phi = 1.0f + kE, nlo = -1.0f - kE, nhi = -1.0f + kE;
    V3 n = v3(0.0f, 0.0f, 0.0f);
    if (in_range(plo, phi, p.x)) n = v3(1.0f, 0.0f, 0.0f);
    else if (in_range(nlo, nhi, p.x)) n = v3(-1.0f, -0.0f, -0.0f);
    else if (in_range(plo, phi, p.y)) n = v3(0.0f, 1.0f, 0.0f);
    else if (in_range(nlo, nhi, p.y)) n = v3(-0.0f, -1.0f, -0.0f);
    if (in_range(plo, phi, p.z)) n = v3(0.0f, 0.0f, 1.0f);
    else if (in_range(nlo, nhi, p.z)) n = v3(-0.0f, -0.0f, -1.0f);
    return n;
}

struct UV { float x, y; };

// UV for Box (4x3 cross atlas), src/rt.rs:468-516
MRT_HD UV box_uv(V3 inv2, V3 hit, V3 pos)
{
    const V3 p = hadam(sub(hit, pos), inv2);
    const float plo = 1.0f - kE, phi = 1.0f + kE, nlo = -1.0f - kE, nhi = -1.0f + kE;
    UV r;
    if (in_range(plo, phi, p.x)) { r.x = (0.5f + 0.5f * p.y) / 4.0f + 2.0f / 4.0f; r.y = div_(0.5f - 0.5f * p.z, 3.0f) + 1.0f / 3.0f; }
    else if (in_range(nlo, nhi, p.x)) { r.x = (0.5f - 0.5f * p.y) / 4.0f; r.y = div_(0.5f - 0.5f * p.z, 3.0f) + 1.0f / 3.0f; }
    else if (in_range(plo, phi, p.y)) { r.x = (0.5f - 0.5f * p.x) / 4.0f + 3.0f / 4.0f; r.y = div_(0.5f - 0.5f * p.z, 3.0f) + 1.0f / 3.0f; }
    else if (in_range(nlo, nhi, p.y)) { r.x = (0.5f + 0.5f * p.x) / 4.0f + 1.0f / 4.0f; r.y = div_(0.5f - 0.5f * p.z, 3.0f) + 1.0f / 3.0f; }
    else if (in_range(plo, phi, p.z)) { r.x = (0.5f + 0.5f * p.x) / 4.0f + 1.0f / 4.0f; r.y = div_(0.5f - 0.5f * p.y, 3.0f); }
    else if (in_range(nlo, nhi, p.z)) { r.x = (0.5f + 0.5f * p.x) / 4.0f + 1.0f / 4.0f; r.y = div_(0.5f + 0.5f * p.y, 3.0f) + 2.0f / 3.0f; }
    else { r.x = 0.0f; r.y = 0.0f; }
    return r;
}

MRT_HD float fract_(float x) { return x - trunc_(x); }

// `f32 as usize`, saturating, NaN -> 0, capped at 2^31 (so that x + y*w fits 64 bits)
MRT_HD uint64_t to_index(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 2147483648.0f) return 2147483648ull;
    return (uint64_t)(u32)v;
}

#if defined(__HIPCC__) || defined(__HIP__)
typedef __attribute__((address_space(3))) volatile float lds_vfloat;   // keeps ds_read / ds_write addressing
typedef __attribute__((address_space(3))) volatile char lds_char;
#endif

struct Scn {
    const float *F;      // the packed scene (LDS): per-lane (divergent) lookups
    const float *U;      // the same blob for wave-uniform reads of the traversal loop: LDS, or global memory
                         // read through the scalar cache into SGPRs (MRT_UNIFORM_SMEM)
    const float *G;      // the whole blob in global memory (the octree leaf lists are not staged when every mesh has a TBVH)
    const Params *P;
    void *wk;            // this lane's walk area (device: an LDS column, entry e at wk[e * wk_stride]; x86 test build: unused)
    u32 wk_stride;
};

// The walk area of one lane: entries of one word each (device: a per-lane LDS column behind the lane stash, slot-major like
// the stash, so lane i always hits bank i; the x86 test build keeps it in a local array).
struct WalkMem {
#if defined(__HIP_DEVICE_COMPILE__)
    lds_vfloat *b;
    u32 stride;
    MRT_HD explicit WalkMem(const Scn &S) : b((lds_vfloat *)S.wk), stride(S.wk_stride) {}
    MRT_HD void put(u32 e, u32 v) { b[e * stride] = u2f(v); }
    MRT_HD u32 get(u32 e) const { return f2u(b[e * stride]); }
    // entry addressed by its byte offset (entry e is at e * pitch()): a running offset costs an add per step where the entry
    // number costs a shift-add (2 against 4 cycles, profiles/microbench/valu_types.hip)
    MRT_HD u32 pitch() const { return stride * 4u; }
    MRT_HD void put_at(u32 off, u32 v) { *(lds_vfloat *)((lds_char *)b + off) = u2f(v); }
#else
    u32 v[kWalkCapMax];
    MRT_HD explicit WalkMem(const Scn &) {}
    MRT_HD void put(u32 e, u32 x) { v[e] = x; }
    MRT_HD u32 get(u32 e) const { return v[e]; }
    MRT_HD u32 pitch() const { return 4u; }
    MRT_HD void put_at(u32 off, u32 x) { v[off >> 2] = x; }
#endif
};

// Texture::get_color, src/rt.rs:618-628; the flat index is clamped to the last texel where the
// reference would panic (documented divergence).  Returns all three channels.
template <u32 FEAT>
MRT_HD V3 tex_fetch(const Scn &S, i32 id, UV uv)
{
    const float *T = S.F + S.P->off_tex + (u32)id * TEX_WORDS;
    const float *X = (FEAT & F_COLD) ? S.G : S.F;      // the texels: cold
    const u32 w = ldu(T, TEX_W), h = ldu(T, TEX_H), fmt = ldu(T, TEX_FMT), off = ldu(T, TEX_OFF);
    if (fmt == TEXFMT_NONE) return v3(0.0f, 0.0f, 0.0f);
    const uint64_t x = to_index(uv.x * (float)w);
    const uint64_t y = to_index(uv.y * (float)h);
    uint64_t idx = x + y * (uint64_t)w;
    const uint64_t last = (uint64_t)w * h - 1;
    if (idx > last) idx = last;
    const u32 i = (u32)idx;
    if (fmt == TEXFMT_U8) {
        const unsigned char *B = reinterpret_cast<const unsigned char *>(X) + off + i * 3u;
        const float *L = S.F + S.P->off_lut;
        return v3(L[B[0]], L[B[1]], L[B[2]]);
    }
    return ld3(X, off + i * 3u);
}

// ---- one closest-hit candidate ----
struct Hit {
    i32 rend;      // renderer index, -1 = none
    u32 inst;      // flattened instance index
    float t0, t1;
    i32 i0, i1;    // mesh triangle ids of the entry / exit hit
};

struct RayPre {
    V3 o, d;
    V3 m;          // recip_patched(d) (valid when the scene has boxes or meshes)
    float dd;      // d.d
    bool d_ok;     // nzfin3(d)
};

template <u32 FEAT>
MRT_HD RayPre ray_pre(V3 o, V3 d)
{
    RayPre r;
    r.o = o; r.d = d;
    if constexpr (FEAT & F_BOX) r.m = recip_patched(d);
    else r.m = v3(0.0f, 0.0f, 0.0f);
    r.dd = dot(d, d);
    r.d_ok = nzfin3(d);
    return r;
}

// ---- conservative ray / box culling shared by the instance BVH and the triangle BVHs (never decides a result) ----
// The slab test runs on the box grown by a margin mg: t = (c - o) * inv -+ (h + mg) * |inv| per axis, with the
// reciprocal direction clamped to +-1e30 (a zero component then reads as "parallel": no constraint inside the slab,
// a sure miss outside) -- coordinates on this route are bounded by 1e6, so nothing overflows and no NaN can arise.
struct CullRay {
    V3 o, inv, ainv;
};
MRT_HD float clamp_inv(float d) { return fmin_(fmax_(rcp_fast(d), -1e30f), 1e30f); }
MRT_HD CullRay cull_ray(V3 o, V3 d)
{
    CullRay c;
    c.o = o;
    c.inv = v3(clamp_inv(d.x), clamp_inv(d.y), clamp_inv(d.z));
    c.ainv = v3(fabs_(c.inv.x), fabs_(c.inv.y), fabs_(c.inv.z));
    return c;
}
MRT_HD bool cull_ok(V3 o, float dd)     // finite origin within the bounded range, unit-length finite direction
{
    return nzfin(dd) && dd > 0.98f && dd < 1.02f && fabs_(o.x) < 1e6f && fabs_(o.y) < 1e6f && fabs_(o.z) < 1e6f;
}
// Margin = k x (largest coordinate distance from the origin to the far side of the box) + kpos x the coordinate magnitudes
// involved (rounding of positions themselves: T + pos, c - o) + 1e-6, DESIGN.md section 7.
//   instance BVH (Params.inst_k / inst_ksq / inst_kpos, set by pack_scene; ONE margin per ray, from the root box): k = 1e-4 --
//     Box::intersect, Triangle::intersect and the mesh arm's root-box test carry a few eps x distance of rounding, ~500 x less
//     -- plus, when a sphere is among the bounded instances, ksq x distance^2 with ksq = 4e-6 / smallest radius: the sphere
//     test's discriminant cancels (b*b - 4ac; rounding eps' ~ 8 eps of |oo|^2), so a ray passing sqrt(r^2 + 2 eps' D^2) - r
//     <= min(eps' D^2 / r, sqrt(2 eps') D) outside a far, small sphere can answer "hit": ksq x D^2 is 8 x the first bound, and it
//     is capped at 4e-3 x D, 4 x the second (tests/edge_cases.py bvh_far_tiny_spheres has such hits in most of its pixels).  (Rounds 1-3: 4e-3 x distance per NODE, which covered spheres down to 1/8000 of their distance and,
//     on the Minecraft-shaped scene, made the margins of rays returning from far floor hits larger than the boxes.)
//   triangle BVH, k = 5e-5, kpos = 1e-6 (round 4; 5e-4 / 1e-5 before): the Moller-Trumbore test has no such cancellation -- a ray
//     it accepts passes within ~10 eps |tv| = 6e-7 x distance of the triangle at any incidence (near-parallel rays are
//     rejected by |det| < E before they can amplify) -- so 5e-5 is ~80 x the bound.  tests/mesh_probe.py (600 k rays: origins up
//     to 3e5 mesh sizes away, vertex- and edge-aimed, grazing, axis-parallel) finds no difference at these constants, none with
//     both scaled down by 10, 6 rays with both scaled down by 100, 5762 with no margin at all.  The margin grows with the DISTANCE of the origin, and scenes with a floor plane send rays back from
//     hundreds of units away (everything a pixel row below the horizon sees): at 5e-4 those rays' margins reached the size of
//     the triangles and their walks visited most of the tree -- 20 k-triangle bench scene 42 -> 19 ms per launch.
#ifndef MRT_MARGIN_SCALE            // tests/mesh_probe.py builds with 0 to show that the probe sees an unsafe margin
#define MRT_MARGIN_SCALE 1.0f
#endif
constexpr float kMarginTri = 5e-5f * MRT_MARGIN_SCALE, kMarginPosTri = 1e-6f * MRT_MARGIN_SCALE;
MRT_HD float cull_margin(float k, float kpos, V3 r, V3 h, float big)
{
    const float ext = fmax_(fmax_(fabs_(r.x) + h.x, fabs_(r.y) + h.y), fabs_(r.z) + h.z);
    return fma_fast(k, ext, fma_fast(kpos, big, 1e-6f * MRT_MARGIN_SCALE));
}
MRT_HD bool cull_slab(const CullRay &R, V3 r, V3 h, float mg, float &tn)
{
    const V3 p = hadam(r, R.inv);
    const V3 q = hadam(v3(h.x + mg, h.y + mg, h.z + mg), R.ainv);
    tn = fmax_(fmax_(p.x - q.x, p.y - q.y), p.z - q.z);
    const float tf = fmin_(fmin_(p.x + q.x, p.y + q.y), p.z + q.z);
    return !(tn > tf || tf < 0.0f);
}

// The reference's own walk, src/rt.rs:740-772: for rays the triangle BVH must not cull (non-finite, non-unit) and meshes
// without one.  The candidate list is consumed in the reference's order (leaf lists of hit leaves, concatenated, consecutive
// duplicates dropped) without being materialised.  The leaf lists are read from global memory.
template <bool ANY, u32 FEAT>
MRT_HD bool mesh_isect_ref(const Scn &S, u32 mesh, V3 ro, V3 rd, V3 m, V3 pos, float &t0, i32 &i0, float &t1, i32 &i1)
{
    const float *F = S.F;
    const float *CT = (FEAT & F_DEEP) ? S.G : S.F;
    const Params &P = *S.P;
    const float *M = F + P.off_mesh + mesh * MESH_WORDS;
    const u32 tri0 = ldu(M, MESH_TRI0), ntri = ldu(M, MESH_NTRI), root = ldu(M, MESH_ROOT), leaf0 = ldu(M, MESH_LEAF0);
    bool any = false;
    i32 k0 = 0, k1 = 0;
    u32 last_id = 0xffffffffu;
    auto test = [&](u32 id) {
        if (id == last_id) return;          // Vec::dedup (src/rt.rs:756)
        last_id = id;
        MRT_COUNT(CT_REF_TRI);
        const float *T = CT + P.off_tri + (tri0 + id) * TRI_WORDS;
        float t;
        if (!tri_isect(add(ld3(T, 0), pos), ld3(T, 3), ld3(T, 6), ro, rd, t)) return;
        const i32 k = total_key(t);
        if (!any) { any = true; t0 = t1 = t; i0 = i1 = (i32)id; k0 = k1 = k; return; }
        if (k < k0) { k0 = k; t0 = t; i0 = (i32)id; }      // min_by: first minimum, src/rt.rs:764
        if (k >= k1) { k1 = k; t1 = t; i1 = (i32)id; }     // max_by: last maximum, src/rt.rs:765
    };

    if (root == NO_NODE) {
        for (u32 i = 0; i < ntri; ++i) { test(i); if (ANY && any) return true; }
        return any;
    }
    // depth <= 3 (BVH::gen(.., 3), src/parser.rs:816): explicit stack of child ranges
    u32 cur[4], end[4];
    int sp = 0;
    cur[0] = root; end[0] = root + 1;
    while (sp >= 0) {
        if (cur[sp] == end[sp]) { --sp; continue; }
        const u32 node = cur[sp]++;
        const float *N = F + P.off_node + node * NODE_WORDS;
        float a0, a1;
        MRT_COUNT(CT_REF_BOX);
        if (!box_isect(ld3(N, NODE_HALF), ro, m, add(pos, ld3(N, NODE_REL)), a0, a1)) continue;
        const u32 first = ldu(N, NODE_FIRST), cnt = ldu(N, NODE_COUNT);
        if (cnt & 0x80000000u) {
            const u32 n = cnt & 0x7fffffffu;
            for (u32 i = 0; i < n; ++i) { test(ldu(S.G, P.off_leaf + leaf0 + first + i)); if (ANY && any) return true; }
        } else if (sp < 3) {
            ++sp;
            cur[sp] = first; end[sp] = first + cnt;
        }
    }
    return any;
}

// Mesh arm of Renderer::intersect, src/rt.rs:740-772 with the octree walk of intersect_bvh, src/rt.rs:707-723 -- answered from
// the other side (mrt_scene.h): a triangle BVH finds the triangles the ray can hit (conservative culling), the exact triangle
// test runs on those, and a hit triangle counts iff the reference's octree walk would have listed it (membership + parent
// tables, exact box tests).  The candidate set, and with it the answer, does not depend on the order of the tests.

// the conservative slab test of the triangle BVHs: box (c, h) in mesh coordinates against a ray given as inv = 1 / d (clamped),
// ainv = |inv|, oinv = o * inv, qm = margin * |inv|:  t = c * inv - o * inv -+ (h * |inv| + mg * |inv|)
struct TriCull { V3 inv, ainv, oinv, qm; };
MRT_HD bool tri_cull_hit(const TriCull &R, float cx, float cy, float cz, float hx, float hy, float hz)
{
    const float px = fma_fast(cx, R.inv.x, -R.oinv.x), py = fma_fast(cy, R.inv.y, -R.oinv.y), pz = fma_fast(cz, R.inv.z, -R.oinv.z);
    const float qx = fma_fast(hx, R.ainv.x, R.qm.x), qy = fma_fast(hy, R.ainv.y, R.qm.y), qz = fma_fast(hz, R.ainv.z, R.qm.z);
    const float tn = fmax_(fmax_(px - qx, py - qy), pz - qz);
    const float tf = fmin_(fmin_(px + qx, py + qy), pz + qz);
    return !(tn > tf || tf < 0.0f);
}

// Diagnostics: mesh queries answered by the reference's own walk instead of a triangle BVH -- slot 6: the 4-wide walk's area
// was full, slot 7: the ray may not be culled (or the mesh has no tree); slot 5: NaN directions answered by the shortcut.  They are rare and expensive (a wavefront waits for
// the lane that takes one), so they are counted: MRT_DEBUG_FALLBACKS=1 makes mrt_get_stats print the totals.
MRT_HD void count_fallback(const Params &P, u32 slot)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_add(P.segments + slot, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    (void)P; (void)slot;
#endif
}

// Meshes beyond the LDS (F_DEEP): 4-WIDE triangle BVH (mrt_scene.h) -- a visit decides four subtrees with one round of
// independent 16-byte reads and only children whose box the ray hits are visited: a fifth of the dependent round trips of the
// binary walk (5.5 / 7.8 / 8.9 visits per path segment on meshes of 1k / 5k / 20k triangles, where the binary walk makes ~30 /
// ~45 / ~60), which is what counts when most nodes and all triangles come through L2.  It costs more VALU per visit (~125), so
// meshes that fit the LDS keep the binary walk.  "While-while" with a per-lane node stack and a queue of postponed leaves,
// both in the lane's walk area (Params.walk_cap entries; stack from entry 0 up, queue from the last entry down):
//   box phase   visit a node = seven independent 16-byte reads + four slab tests; hit leaves go to the queue; of the hit
//               internal children the first is visited next and the others become ONE stack entry ((index of the node's
//               first child) << 4 | mask of children still to visit: internal children are consecutive nodes) -- so the stack
//               is never deeper than the tree; when nothing internal was hit the next pending child of the top entry is taken,
//               read at the START of the visit (next to the node: its latency is never on the chain).  The step is one basic
//               block; the loop's only branch is its wave-uniform exit.  A visit needs four free entries.
//   exact phase all queued triangles, one per lane per trip.
// A closest-hit query walks until the tree is exhausted or the area is full; a shadow query (ANY) stops the box phase at the
// first leaf: its first candidate ends it.  An area that the stack alone fills (a degenerate tree) is answered by the caller
// through the reference's own walk: exact, only slow.  Returns 0 none, 1 hit, 2 overflow.
template <bool ANY, u32 FEAT>
MRT_HD int mesh_walk4(const Scn &S, u32 tb, const TriCull &R, u32 tri0, u32 root, V3 ro, V3 rd, V3 m, V3 pos, float &t0, i32 &i0, float &t1, i32 &i1)
{
    const Params &P = *S.P;
    const float *F = S.F;
    const float *C = S.G;                                // membership tables and triangles: cold (F_DEEP implies F_COLD)
    const float *CT = S.G;
    const float *N0 = F + P.off_node;
    const float *B0 = S.F + P.off_tbvh;
    bool any = false;
    i32 k0 = 0, k1 = 0;
    u32 s0 = 0, s1 = 0;
    float a0, a1;
    WalkMem W(S);
    const u32 cap = P.walk_cap;
    u32 cur = tb;              // index of the node to visit, NO_NODE: none
    u32 sp = 0u;               // stack entries W[0 .. sp)
    for (;;) {
        u32 nq = 0u;           // queued leaves W[cap - 1 .. cap - nq]
        u32 probe_steps = 0; (void)probe_steps;
        bool walking = cur != NO_NODE && sp + 4u <= cap;
        while (walking) {
            ++probe_steps;
            MRT_COUNT(CT_TBVH_NODE);
            const u32 noff = mul24(cur, B4_WORDS);               // (node indices are below 2^24: pack_scene)
            const float *Nd = B0 + noff;
            if (cur >= P.n_tbvh_hot) Nd = S.G + P.off_tbvh + noff;      // below the staged levels
            const u32 top = W.get(sp ? sp - 1u : 0u);            // the entry a pop would take from (meaningless when sp == 0)
            const F4 cx = ld4(Nd, B4_CX), cy = ld4(Nd, B4_CY), cz = ld4(Nd, B4_CZ), hx = ld4(Nd, B4_HX), hy = ld4(Nd, B4_HY), hz = ld4(Nd, B4_HZ);
            const F4 cw = ld4(Nd, B4_CHILD);
            sched_fence();                                       // all eight reads are issued here, whatever is scheduled below
            const u32 w0 = f2u(cw.x), w1 = f2u(cw.y), w2 = f2u(cw.z), w3 = f2u(cw.w);
            const bool h0 = tri_cull_hit(R, cx.x, cy.x, cz.x, hx.x, hy.x, hz.x) && w0 != 0u, h1 = tri_cull_hit(R, cx.y, cy.y, cz.y, hx.y, hy.y, hz.y) && w1 != 0u;
            const bool h2 = tri_cull_hit(R, cx.z, cy.z, cz.z, hx.z, hy.z, hz.z) && w2 != 0u, h3 = tri_cull_hit(R, cx.w, cy.w, cz.w, hx.w, hy.w, hz.w) && w3 != 0u;
            // hit internal children as a mask (internal children occupy the first slots and are consecutive nodes)
            const u32 hm = ((h0 && (w0 >> 31)) ? 1u : 0u) | ((h1 && (w1 >> 31)) ? 2u : 0u) | ((h2 && (w2 >> 31)) ? 4u : 0u) | ((h3 && (w3 >> 31)) ? 8u : 0u);
            const u32 c0 = w0 & ~B4_INTERNAL;                    // index of child 0 (only used when some internal child was hit)
            const u32 from = hm ? ((c0 << 4) | hm) : top;        // entry the next node comes out of
            const bool have = hm != 0u || sp != 0u;
            const u32 k = lowest_bit(from & 15u);
            const u32 rest = from & (from - 1u);                 // the entry without that child (its mask is non-zero whenever it is used)
            cur = have ? (from >> 4) + k : NO_NODE;
            const bool keep = (rest & 15u) != 0u;
            // hm != 0: push `rest` when children remain; hm == 0: the top entry shrinks in place, or goes when it is spent
            const u32 at = hm ? sp : (sp ? sp - 1u : 0u);
            W.put(at, rest);                                     // (a spent or unused entry is rewritten harmlessly: W[at] is free or dead)
            sp = hm ? sp + (keep ? 1u : 0u) : sp - ((sp != 0u && !keep) ? 1u : 0u);
            // leaves: one store per child to the next free queue entry; it only counts when the pointer moves
            W.put(cap - 1u - nq, w0); nq += (h0 && !(w0 >> 31)) ? 1u : 0u;
            W.put(cap - 1u - nq, w1); nq += (h1 && !(w1 >> 31)) ? 1u : 0u;
            W.put(cap - 1u - nq, w2); nq += (h2 && !(w2 >> 31)) ? 1u : 0u;
            W.put(cap - 1u - nq, w3); nq += (h3 && !(w3 >> 31)) ? 1u : 0u;
            walking = cur != NO_NODE && sp + nq + 4u <= cap && (!ANY || nq == 0u);
        }
        if (nq == 0u) {
            MRT_PROBE_ROUND(probe_steps, 0u, 0u);
            if (cur == NO_NODE) break;
            MRT_COUNT(CT_WALK_OVERFLOW);                         // nodes left, nothing queued, no room
            return 2;
        }
        MRT_COUNT(CT_WALK_ROUND);
        u32 e = 0u, j = 0u, leaf = W.get(cap - 1u), probe_tris = 0; (void)probe_tris;
        while (e < nq) {
            const u32 id = (leaf & 0xffffffu) + j;
            ++j; ++probe_tris;
            if (j == (leaf >> 24)) { ++e; j = 0u; if (e < nq) leaf = W.get(cap - 1u - e); }
            // (the exact half is written out here, `continue` by `continue`, as in the binary walks: the same statements behind
            // a helper with early returns compiled into a slower loop, 3.10 against 3.33 Gsamples/s on the 967-triangle scene)
            const float *T = CT + P.off_tri + (tri0 + id) * TRI_WORDS;
            float t;
            MRT_COUNT(CT_TBVH_TRI);
            if (!tri_isect(add(ld3(T, 0), pos), ld3(T, 3), ld3(T, 6), ro, rd, t)) continue;
            MRT_COUNT(CT_TBVH_TRI_HIT);
            // candidate iff some octree leaf listing the triangle is reached: every box from that leaf up to the root hit
            const u32 head = ldu(C, P.off_memb + tri0 + id);
            const u32 e0 = head & 0xffffffu, ne = head >> 24;
            u32 sl = 0xffffffffu, sh = 0u;
            bool cand = false;
            for (u32 q = 0; q < ne; ++q) {
                const u32 w = ldu(C, P.off_membe + e0 + q);
                u32 n = root + (w >> MEMB_SLOT_BITS);
                bool reached = true;
                while (n != root) {
                    MRT_COUNT(CT_MEMB_BOX);
                    if (!box_isect(ld3(N0, n * NODE_WORDS + NODE_HALF), ro, m, add(pos, ld3(N0, n * NODE_WORDS + NODE_REL)), a0, a1)) { reached = false; break; }
                    n = ldu(F, P.off_parent + n);
                }
                if (!reached) continue;
                if (ANY) return 1;
                const u32 slot = w & MEMB_SLOT_MASK;    // entries are in slot order
                if (!cand) sl = slot;
                sh = slot;
                cand = true;
            }
            if (!cand) continue;
            const i32 k = total_key(t);
            if (!any) { any = true; t0 = t1 = t; i0 = i1 = (i32)id; k0 = k1 = k; s0 = sl; s1 = sh; continue; }
            if (k < k0 || (k == k0 && sl < s0)) { k0 = k; s0 = sl; t0 = t; i0 = (i32)id; }      // min_by: first minimum, src/rt.rs:764
            if (k > k1 || (k == k1 && sh > s1)) { k1 = k; s1 = sh; t1 = t; i1 = (i32)id; }      // max_by: last maximum, src/rt.rs:765
        }
        MRT_PROBE_ROUND(probe_steps, probe_tris, 0u);
        if (cur == NO_NODE) break;
    }
    return any ? 1 : 0;
}

// Meshes in LDS walk a BINARY triangle BVH, threaded (stackless, depth-first, skip links), one 32-byte node per step: 27 VALU
// per step, which is what these walks are bound by -- a wavefront pays the longest walk of its lanes (tests/emu/round_probe.cpp):
//   closest-hit query, kernels with a walk area (F_COLD): EVERY leaf postponed -- the box walk runs until the tree is
//     exhausted or the lane's queue of kLeafQueue leaves is full, then all queued triangles are tested, one per lane per trip
//     (76 -> 46 box steps, 12 -> 11 triangle tests per loop iteration of the bench mesh against two-leaf rounds);
//   otherwise ("while-while" with one postponed leaf): a lane walks boxes until it holds two leaves (or the end), then the
//     wavefront runs the exact tests of both.  Shadow queries keep this form: their first candidate ends the walk.
// The step is one basic block (next node, leaf bookkeeping and the done flag are selects); the loop's only branch is its exit.
template <bool ANY, u32 FEAT>
MRT_HD bool mesh_isect(const Scn &S, u32 mesh, V3 ro, V3 rd, float dd, V3 m, V3 pos, float &t0, i32 &i0, float &t1, i32 &i1)
{
    const float *F = S.F;
    const float *C = (FEAT & F_DEEP) ? S.G : S.F;        // membership tables and triangles: cold only for meshes beyond the LDS
    const float *CT = C;
    const Params &P = *S.P;
    const float *M = F + P.off_mesh + mesh * MESH_WORDS;
    const u32 tri0 = ldu(M, MESH_TRI0), root = ldu(M, MESH_ROOT);
    const u32 tb = ldu(M, MESH_TBVH);
    bool any = false;
    i32 k0 = 0, k1 = 0;
    MRT_COUNT(CT_MESH_CALL);

    // Rays that are NaN in every component of their direction (Vec3f::norm of a zero or non-finite vector, src/lin.rs:64-66) or
    // of their origin (a hit "at t = NaN": Plane::intersect lets 0 / 0 through, src/rt.rs:400-412, and the next ray leaves from
    // that point) fail every comparison of Box::intersect (src/rt.rs:299-333): every box of the octree "is hit" and every
    // listed triangle is tested.  Triangle::intersect (src/rt.rs:361-398) can then only reject by |det| < E -- u, v and t are
    // NaN and pass -- and answers t = NaN, so all keys are equal and the reference returns the FIRST and the LAST accepted
    // triangle of its concatenated leaf lists (min_by keeps the first, max_by the last of equal elements, src/rt.rs:764-765) --
    // after testing every triangle of the mesh, some of them several times.  The same answer from the two ends of the list:
    // such rays are one in ten million, but a path keeps them to its last bounce, and the lane that met one held its wavefront
    // for ~80 ms at the end of a launch of the 20 k-triangle scene (19 ms otherwise).
    {
        const bool d_nan = rd.x != rd.x && rd.y != rd.y && rd.z != rd.z, o_nan = ro.x != ro.x && ro.y != ro.y && ro.z != ro.z;
        const bool d_num = rd.x == rd.x && rd.y == rd.y && rd.z == rd.z;
        if (root != NO_NODE && (d_nan || (o_nan && d_num))) {
            const u32 n_ids = ldu(M, MESH_NIDS), leaf0 = ldu(M, MESH_LEAF0);
            count_fallback(P, 5);
            auto accept = [&](u32 at, float &t) {
                const float *T = CT + P.off_tri + (tri0 + ldu(S.G, P.off_leaf + leaf0 + at)) * TRI_WORDS;
                return tri_isect(add(ld3(T, 0), pos), ld3(T, 3), ld3(T, 6), ro, rd, t);
            };
            u32 a = 0u, b = n_ids;
            float ta = 0.0f, tb_ = 0.0f;
            bool found = false;
            for (; a < n_ids; ++a) if (accept(a, ta)) { found = true; break; }
            if (!found) return false;
            if (ANY) return true;
            while (b-- > a) if (accept(b, tb_)) break;                   // (stops at `a` at the latest)
            t0 = ta; i0 = (i32)ldu(S.G, P.off_leaf + leaf0 + a); t1 = tb_; i1 = (i32)ldu(S.G, P.off_leaf + leaf0 + b);
            return true;
        }
    }

    if constexpr (FEAT & F_DEEP) {
        // ---- meshes beyond the LDS: the 4-wide table ----
        const V3 ol = sub(ro, pos);
        if (tb != NO_NODE && cull_ok(ol, dd) && fabs_(pos.x) < 1e6f && fabs_(pos.y) < 1e6f && fabs_(pos.z) < 1e6f) {
            const float *N0 = F + P.off_node;
            float a0, a1;
            // a ray that misses the octree root has no candidates at all, src/rt.rs:745
            if (!box_isect(ld3(N0, root * NODE_WORDS + NODE_HALF), ro, m, add(pos, ld3(N0, root * NODE_WORDS + NODE_REL)), a0, a1)) return false;
            MRT_COUNT(CT_MESH_ROOT_HIT);
            const CullRay CR = cull_ray(ol, rd);
            TriCull R;
            R.inv = CR.inv; R.ainv = CR.ainv;
            {
                const V3 c = ld3(M, MESH_BC), hh = ld3(M, MESH_BH);
                const float big = fmax_(fmax_(fabs_(pos.x), fabs_(pos.y)), fabs_(pos.z)) + fmax_(fmax_(fabs_(c.x) + hh.x, fabs_(c.y) + hh.y), fabs_(c.z) + hh.z)
                                  + fmax_(fmax_(fabs_(ol.x), fabs_(ol.y)), fabs_(ol.z));
                const float mg = cull_margin(kMarginTri, kMarginPosTri, sub(c, ol), hh, big);
                R.oinv = hadam(ol, R.inv);
                R.qm = muls(R.ainv, mg);
            }
            const int r = mesh_walk4<ANY, FEAT>(S, tb, R, tri0, root, ro, rd, m, pos, t0, i0, t1, i1);
            if (r == 2) { count_fallback(P, 6); return mesh_isect_ref<ANY, FEAT>(S, mesh, ro, rd, m, pos, t0, i0, t1, i1); }
            return r != 0;
        }
        count_fallback(P, 7);
        MRT_PROBE_FALLBACK(ro, rd, dd);
        return mesh_isect_ref<ANY, FEAT>(S, mesh, ro, rd, m, pos, t0, i0, t1, i1);
    }

    // ---- triangle BVH route (mrt_scene.h): the triangles the ray hits, then their candidacy in the reference's octree walk ----
    const V3 ol = sub(ro, pos);
    if (tb != NO_NODE && cull_ok(ol, dd) && fabs_(pos.x) < 1e6f && fabs_(pos.y) < 1e6f && fabs_(pos.z) < 1e6f) {
        const float *N0 = F + P.off_node;
        float a0, a1;
        // a ray that misses the octree root has no candidates at all, src/rt.rs:745
        if (!box_isect(ld3(N0, root * NODE_WORDS + NODE_HALF), ro, m, add(pos, ld3(N0, root * NODE_WORDS + NODE_REL)), a0, a1)) return false;
        MRT_COUNT(CT_MESH_ROOT_HIT);
        const float *B0 = F + P.off_tbvh;
        // one culling margin per ray, from the mesh bounds
        const CullRay R = cull_ray(ol, rd);
        V3 oinv, qm;
        {
            const V3 c = ld3(M, MESH_BC), hh = ld3(M, MESH_BH);
            const float big = fmax_(fmax_(fabs_(pos.x), fabs_(pos.y)), fabs_(pos.z)) + fmax_(fmax_(fabs_(c.x) + hh.x, fabs_(c.y) + hh.y), fabs_(c.z) + hh.z)
                              + fmax_(fmax_(fabs_(ol.x), fabs_(ol.y)), fabs_(ol.z));
            const float mg = cull_margin(kMarginTri, kMarginPosTri, sub(c, ol), hh, big);
            oinv = hadam(ol, R.inv);
            qm = muls(R.ainv, mg);
        }
        u32 s0 = 0, s1 = 0;
        u32 node = tb;
        if constexpr ((!ANY || MRT_SHADOW_QUEUE) && has_walk_area(FEAT)) {
            // Closest-hit walk with EVERY leaf postponed: the box walk runs on until the tree is exhausted (or the lane's queue
            // of kQ = Params.walk_cap leaves is full), then the exact tests of all queued leaves run together, one triangle per lane per trip.
            // A wavefront pays, per round, the longest box walk and the longest triangle list of its lanes: with two leaves
            // per round (below) that is 76 box steps + 12 triangle tests per loop iteration on the 967-triangle bench mesh,
            // with one round per walk 46 + 11 (tests/emu/round_probe.cpp).  Shadow queries keep the two-leaf rounds: their
            // first candidate ends the walk.  The candidate set, and with it the answer, is the same in any order.
            const u32 kQ = P.walk_cap;                             // kLeafQueue entries at least (plan_launch)
            WalkMem q(S);
            for (;;) {
                u32 qo = 0u;                                            // byte offset of the next free queue entry
                const u32 q_pitch = opaque_s(q.pitch()), q_full = kQ * q_pitch;     // (a scalar register, not a literal to re-load per step)
                u32 probe_steps = 0; (void)probe_steps;
                bool walking = node != BVH_END;
                while (walking) {
                    ++probe_steps;
                    const F4 na = ld4(B0, node * BVH_WORDS), nb = ld4(B0, node * BVH_WORDS + 4);
                    MRT_COUNT(CT_TBVH_NODE);
                    const u32 skip = f2u(nb.z);
                    const u32 leaf = opaque_v(f2u(nb.w)), child = node + 1u;        // (an integer to the compiler: `!= 0` stays v_cmp_ne_u32)
                    const float px = fma_fast(na.x, R.inv.x, -oinv.x), py = fma_fast(na.y, R.inv.y, -oinv.y), pz = fma_fast(na.z, R.inv.z, -oinv.z);
                    const float qx = fma_fast(na.w, R.ainv.x, qm.x), qy = fma_fast(nb.x, R.ainv.y, qm.y), qz = fma_fast(nb.y, R.ainv.z, qm.z);
                    const float tn = fmax_(fmax_(px - qx, py - qy), pz - qz);
                    const float tf = fmin_(fmin_(px + qx, py + qy), pz + qz);
                    const bool hit = !(tn > tf || tf < 0.0f);
                    q.put_at(qo, leaf);                                // branch-free: the slot only counts when the leaf was hit
                    qo += (hit && leaf != 0u) ? q_pitch : 0u;
                    // the table is in depth-first order: a leaf's skip link IS the next node (pack_scene puts a never-hit
                    // sentinel behind the last leaf of a tree), so "hit" alone decides -- one compare less per step
                    node = hit ? child : skip;
                    walking = node != BVH_END && qo < q_full;
                }
                const u32 nq = qo / q_pitch;
                if (nq == 0u) { MRT_PROBE_ROUND(probe_steps, 0u, 0u); break; }
                u32 e = 0u, j = 0u, leaf = q.get(0u), probe_tris = 0; (void)probe_tris;
                while (e < nq) {
                    const u32 id = (leaf & 0xffffffu) + j;
                    ++j; ++probe_tris;
                    if (j == (leaf >> 24)) { ++e; j = 0u; if (e < nq) leaf = q.get(e); }
                    const float *T = CT + P.off_tri + (tri0 + id) * TRI_WORDS;
                    float t;
                    MRT_COUNT(CT_TBVH_TRI);
                    if (!tri_isect(add(ld3(T, 0), pos), ld3(T, 3), ld3(T, 6), ro, rd, t)) continue;
                    MRT_COUNT(CT_TBVH_TRI_HIT);
                    const u32 head = ldu(C, P.off_memb + tri0 + id);
                    const u32 e0 = head & 0xffffffu, ne = head >> 24;
                    u32 sl = 0xffffffffu, sh = 0u;
                    bool cand = false;
                    for (u32 k = 0; k < ne; ++k) {
                        const u32 w = ldu(C, P.off_membe + e0 + k);
                        u32 n = root + (w >> MEMB_SLOT_BITS);
                        bool reached = true;
                        while (n != root) {
                            MRT_COUNT(CT_MEMB_BOX);
                            if (!box_isect(ld3(N0, n * NODE_WORDS + NODE_HALF), ro, m, add(pos, ld3(N0, n * NODE_WORDS + NODE_REL)), a0, a1)) { reached = false; break; }
                            n = ldu(F, P.off_parent + n);
                        }
                        if (!reached) continue;
                        if (ANY) return true;
                        const u32 slot = w & MEMB_SLOT_MASK;    // entries are in slot order
                        if (!cand) sl = slot;
                        sh = slot;
                        cand = true;
                    }
                    if (!cand) continue;
                    const i32 k = total_key(t);
                    if (!any) { any = true; t0 = t1 = t; i0 = i1 = (i32)id; k0 = k1 = k; s0 = sl; s1 = sh; continue; }
                    if (k < k0 || (k == k0 && sl < s0)) { k0 = k; s0 = sl; t0 = t; i0 = (i32)id; }      // min_by: first minimum, src/rt.rs:764
                    if (k > k1 || (k == k1 && sh > s1)) { k1 = k; s1 = sh; t1 = t; i1 = (i32)id; }      // max_by: last maximum, src/rt.rs:765
                }
                MRT_PROBE_ROUND(probe_steps, probe_tris, 0u);
                if (node == BVH_END) break;
            }
            return any;
        }
        // "while-while" with one postponed leaf: a lane walks boxes until it has found two leaves (or the end), then the
        // wavefront runs the exact triangle tests of both together -- fewer, fuller rounds than one leaf at a time.
        for (;;) {
            u32 leaf_a = 0u, leaf_b = 0u;
            u32 probe_steps = 0, probe_membs = 0; (void)probe_steps; (void)probe_membs;
            // branch-free step: the only branch of the box walk is its wave-uniform exit; a lane that is done (end of the
            // tree, or two leaves in hand) keeps re-reading the root and changes nothing
            bool walking = node != BVH_END;
            while (walking) {
                ++probe_steps;
                const F4 na = ld4(B0, node * BVH_WORDS), nb = ld4(B0, node * BVH_WORDS + 4);
                MRT_COUNT(CT_TBVH_NODE);
                const u32 skip = f2u(nb.z);
                const u32 leaf = f2u(nb.w), child = node + 1u;
                // t = c * inv - o * inv -+ (h * |inv| + mg * |inv|)
                const float px = fma_fast(na.x, R.inv.x, -oinv.x), py = fma_fast(na.y, R.inv.y, -oinv.y), pz = fma_fast(na.z, R.inv.z, -oinv.z);
                const float qx = fma_fast(na.w, R.ainv.x, qm.x), qy = fma_fast(nb.x, R.ainv.y, qm.y), qz = fma_fast(nb.y, R.ainv.z, qm.z);
                const float tn = fmax_(fmax_(px - qx, py - qy), pz - qz);
                const float tf = fmin_(fmin_(px + qx, py + qy), pz + qz);
                const bool hit = !(tn > tf || tf < 0.0f);
                const u32 cand = hit ? leaf : 0u;                  // a leaf to test, or 0
                const bool have_a = leaf_a != 0u;
                leaf_b |= have_a ? cand : 0u;
                leaf_a |= have_a ? 0u : cand;
                node = hit ? child : skip;                         // (a leaf's skip link is the next node: see above)
                walking = node != BVH_END && leaf_b == 0u;
            }
            if (leaf_a == 0u) { MRT_PROBE_ROUND(probe_steps, 0u, 0u); break; }
            const u32 cnt_a = leaf_a >> 24, first_a = leaf_a & 0xffffffu, cnt_b = leaf_b >> 24, first_b = leaf_b & 0xffffffu;
            MRT_PROBE_ROUND(probe_steps, cnt_a + cnt_b, 0u);
            for (u32 j = 0; j < cnt_a + cnt_b; ++j) {
                const u32 id = j < cnt_a ? first_a + j : first_b + (j - cnt_a);
                const float *T = CT + P.off_tri + (tri0 + id) * TRI_WORDS;
                float t;
                MRT_COUNT(CT_TBVH_TRI);
                if (!tri_isect(add(ld3(T, 0), pos), ld3(T, 3), ld3(T, 6), ro, rd, t)) continue;
                MRT_COUNT(CT_TBVH_TRI_HIT);
                // candidate iff some octree leaf listing the triangle is reached: every box from that leaf up to the root hit
                const u32 head = ldu(C, P.off_memb + tri0 + id);
                const u32 e0 = head & 0xffffffu, ne = head >> 24;
                u32 sl = 0xffffffffu, sh = 0u;
                bool cand = false;
                for (u32 e = 0; e < ne; ++e) {
                    const u32 w = ldu(C, P.off_membe + e0 + e);
                    u32 n = root + (w >> MEMB_SLOT_BITS);
                    bool reached = true;
                    while (n != root) {
                        MRT_COUNT(CT_MEMB_BOX);
                        if (!box_isect(ld3(N0, n * NODE_WORDS + NODE_HALF), ro, m, add(pos, ld3(N0, n * NODE_WORDS + NODE_REL)), a0, a1)) { reached = false; break; }
                        n = ldu(F, P.off_parent + n);
                    }
                    if (!reached) continue;
                    if (ANY) return true;
                    const u32 slot = w & MEMB_SLOT_MASK;    // entries are in slot order
                    if (!cand) sl = slot;
                    sh = slot;
                    cand = true;
                }
                if (!cand) continue;
                const i32 k = total_key(t);
                if (!any) { any = true; t0 = t1 = t; i0 = i1 = (i32)id; k0 = k1 = k; s0 = sl; s1 = sh; continue; }
                if (k < k0 || (k == k0 && sl < s0)) { k0 = k; s0 = sl; t0 = t; i0 = (i32)id; }      // min_by: first minimum, src/rt.rs:764
                if (k > k1 || (k == k1 && sh > s1)) { k1 = k; s1 = sh; t1 = t; i1 = (i32)id; }      // max_by: last maximum, src/rt.rs:765
            }
        }
        return any;
    }

    count_fallback(P, 7);
    MRT_PROBE_FALLBACK(ro, rd, dd);
    return mesh_isect_ref<ANY, FEAT>(S, mesh, ro, rd, m, pos, t0, i0, t1, i1);
}

// Renderer::intersect for flat instance i (record words ia, ib): the exact test of the reference, src/rt.rs:725-774
// UNIFORM: every active lane of the wavefront tests the SAME instance (the linear scans).  The kernels with triangle / mesh
// code then read its tag into a scalar register, so the kind dispatch, the identity test and the transform's address are
// scalar instructions and scalar branches instead of per-lane compares (4 VALU cycles each, profiles/microbench/
// valu_types.hip) of a value all lanes share: mesh scenes +1 ... 2 %.  The plane / sphere kernels lose by it (headline 8268 ->
// 7810 Msamples/s: their scalar unit is as busy as their vector pipes) and keep the per-lane form.
#if defined(__HIP_DEVICE_COMPILE__)
MRT_HD u32 wave_uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
#else
MRT_HD u32 wave_uniform(u32 v) { return v; }
#endif
template <bool ANY, u32 FEAT, bool UNIFORM = false>
MRT_HD bool isect_instance(const Scn &S, const RayPre &ray, u32 i, const F4 &ia, const F4 &ib, float &t0, float &t1, i32 &i0, i32 &i1, const float *X0 = nullptr)
{
    const float *F = S.U;
    const Params &P = *S.P;
    const V3 pos = v3(ia.x, ia.y, ia.z);
    const u32 tag = (UNIFORM && (FEAT & F_TRI) && MRT_UNIFORM_TAG) ? wave_uniform(f2u(ib.x)) : f2u(ib.x);
    // (F_IDENT kernels: the kind is all that is left of the tag in the scan; dispatched on a scalar copy of it, +0.7 %)
    const u32 kind = (UNIFORM && (FEAT & F_IDENT)) ? wave_uniform(tag & TAG_KIND_MASK) : (tag & TAG_KIND_MASK);
    const bool ident = (FEAT & F_IDENT) ? true : (tag & TAG_IDENT) != 0;
    // the tag carries the word offset of the transform; F_IDENT: all instances share ONE de-duplicated identity entry, X0, and
    // trace() has already sent the direction through it where that changes bits (ray.d_ok is then true for every lane)
    const float *X = (FEAT & F_IDENT) ? X0 : F + P.off_xf + (tag >> TAG_XF_SHIFT);
    // n_ray.orig = pos + R*(L*(orig - pos)), n_ray.dir = R*(L*dir), src/rt.rs:729-733
    const V3 ro = add(pos, xf_vec(X, ident, sub(ray.o, pos)));
    const bool fast_d = (FEAT & F_IDENT) ? true : (ident && ray.d_ok);
    V3 rd = ray.d, m = ray.m;
    float dd = ray.dd;
    if (!fast_d) {
        rd = xf_full(X, ray.d);
        if constexpr (FEAT & F_BOX) m = recip_patched(rd);
        dd = dot(rd, rd);
    }
    t0 = 0.0f; t1 = 0.0f; i0 = -1; i1 = -1;
    if (kind == KIND_SPHERE) return sphere_isect(ia.w, sub(ro, pos), rd, dd, t0, t1);
    if (kind == KIND_PLANE) { const bool h = plane_isect(v3(ib.y, ib.z, ib.w), ia.w, ro, rd, t0); t1 = t0; return h; }
    if ((FEAT & F_BOX) && kind == KIND_BOX) return box_isect(v3(ia.w, ib.y, ib.z), ro, m, pos, t0, t1);
    if (FEAT & F_TRI) {
        const float *R = F + P.off_rend + ldu(F, P.off_instx + i * INSTX_WORDS + INSTX_REND) * REND_WORDS;
        if (kind == KIND_TRIANGLE) {
            const bool h = tri_isect(add(ld3(R, REND_GEO), pos), ld3(R, REND_GEO + 3), ld3(R, REND_GEO + 6), ro, rd, t0);
            t1 = t0;
            return h;
        }
        if ((FEAT & F_BOX) && kind == KIND_MESH) return mesh_isect<ANY, FEAT>(S, ldu(R, REND_GEO), ro, rd, dd, m, pos, t0, i0, t1, i1);
    }
    return false;
}

template <bool V> struct BoolTag { static constexpr bool value = V; };

// RayTracer::closest_hit, src/rt.rs:867-898: every renderer x instance in order, first minimum of the entry distance
// under f32::total_cmp, i.e. the lexicographic minimum of (total_cmp key, flat instance index) -- which is how the
// BVH variant, visiting candidates in tree order, returns the very same hit.  ANY = true answers only Some / None
// (the shadow query of src/rt.rs:1036).  The linear loop fetches each record one iteration ahead of its use.
template <bool ANY, u32 FEAT>
MRT_HD bool trace(const Scn &S, const RayPre &ray_, Hit &best)
{
    const float *F = S.U;
    const Params &P = *S.P;
    i32 best_key = 0x7fffffff;
    best.rend = -1; best.inst = 0; best.t0 = 0.0f; best.t1 = 0.0f; best.i0 = -1; best.i1 = -1;
    const float *I = F + P.off_inst;

    if (ANY) { MRT_COUNT(CT_TRACE_ANY); } else { MRT_COUNT(CT_TRACE); }
    // IN_ORDER: candidates arrive in increasing flat index (the linear scan of a scene without an instance BVH), so the
    // first minimum is kept by a strict comparison alone -- an equal key never replaces an earlier candidate, and the
    // initial key 0x7fffffff loses to every real one (a NaN distance maps to 0x80000000, the smallest key).
    // F_IDENT (every instance untransformed): n_ray.dir = R*(L*dir) is the same vector for every instance -- dir itself, or, for a
    // direction with a zero / infinite / NaN component, what the two identity mat-vecs make of it (src/rt.rs:729-733) -- so it is
    // formed once per query instead of once per instance
    const float *X0 = nullptr;
    RayPre ray_i = ray_;
    if constexpr (FEAT & F_IDENT) {
        X0 = F + P.off_xf + (ldu(I, INST_TAG) >> TAG_XF_SHIFT);
        if (!ray_.d_ok) {
            ray_i.d = xf_full(X0, ray_.d);
            if constexpr (FEAT & F_BOX) ray_i.m = recip_patched(ray_i.d);
            ray_i.dd = dot(ray_i.d, ray_i.d);
        }
    }
    const RayPre &ray = ray_i;
    auto consider = [&](u32 i, const F4 &ia, const F4 &ib, auto in_order, auto uniform) -> bool {
        float t0, t1;
        MRT_COUNT(CT_LIN_TEST);
        MRT_PROBE_INST(i);
        i32 i0, i1;
        if (!isect_instance<ANY, FEAT, decltype(uniform)::value>(S, ray, i, ia, ib, t0, t1, i0, i1, X0)) return false;
        if (ANY) return true;
        const i32 key = total_key(t0);
        // any order (BVH scenes): the lexicographic minimum of (key, flat index) as ONE unsigned 64-bit compare -- the key with
        // its sign bit flipped in the high word, the index in the low word; "no candidate yet" is the all-ones pair
        // (best_key = 0x7fffffff, best.inst = 0xffffffff), below which every real candidate lies
        const unsigned long long pair = ((unsigned long long)((u32)key ^ 0x80000000u) << 32) | i;
        const unsigned long long best_pair = ((unsigned long long)((u32)best_key ^ 0x80000000u) << 32) | (best.rend < 0 ? 0xffffffffu : best.inst);
        const bool better = decltype(in_order)::value ? key < best_key : pair < best_pair;
        if (better) {
            best_key = key;
            best.rend = 0; best.inst = i; best.t1 = t1; best.i0 = i0; best.i1 = i1;
            // (kernels without an instance BVH recover t0 from its key after the scan: one select less per candidate, 4 cycles)
            if constexpr ((FEAT & F_BVH) != 0 || !MRT_T0_FROM_KEY) best.t0 = t0;
        }
        return false;
    };
    using InOrder = BoolTag<(FEAT & F_BVH) == 0>;       // BVH scenes keep the general rule for their short linear list too
    using AnyOrder = BoolTag<false>;
    using Shared = BoolTag<true>;                       // one instance for the whole wavefront (linear scans)
    using PerLane = BoolTag<false>;                     // BVH leaves: every lane its own

    // ---- linear scan: every instance, or (BVH scenes) the ones that cannot be bounded: planes, odd transforms ----
    const bool bvh = (FEAT & F_BVH) != 0;
    const u32 n = bvh ? P.n_lin : P.n_inst;
    if (!bvh && !(FEAT & F_TRI) && n) {          // (with triangle / mesh code the test is too large to have twice)
        // two record buffers used in turn, each loaded one instance ahead of its use: the next record is in flight
        // while the current one is tested, and no register copies are needed to keep it (one buffer + a copy per
        // instance cost 8 v_mov each)
        u32 j = 0;
        F4 a0 = ld4(I, 0), a1 = ld4(I, 4), b0 = a0, b1 = a1;
        for (;;) {
            if (j + 1u < n) { b0 = ld4(I, (j + 1u) * INST_WORDS); b1 = ld4(I, (j + 1u) * INST_WORDS + 4); }
            if (consider(j, a0, a1, InOrder(), Shared())) return true;
            if (++j >= n) break;
            if (j + 1u < n) { a0 = ld4(I, (j + 1u) * INST_WORDS); a1 = ld4(I, (j + 1u) * INST_WORDS + 4); }
            if (consider(j, b0, b1, InOrder(), Shared())) return true;
            if (++j >= n) break;
        }
    } else if (n) {
        const float *Lst = F + P.off_lin;
        u32 cur = bvh ? ldu(Lst, 0) : 0u;
        F4 qa = ld4(I, cur * INST_WORDS), qb = ld4(I, cur * INST_WORDS + 4);
        for (u32 j = 0; j < n; ++j) {
            const F4 ia = qa, ib = qb;
            const u32 i = cur;
            if (j + 1 < n) {
                cur = bvh ? ldu(Lst, j + 1) : j + 1;
                qa = ld4(I, cur * INST_WORDS); qb = ld4(I, cur * INST_WORDS + 4);
            }
            if (consider(i, ia, ib, InOrder(), Shared())) return true;
        }
    }

    // ---- BVH over the bounded instances: threaded depth-first walk, one lane = one walk ----
    if constexpr (bvh) {
        // Culling must never drop an instance whose exact test would answer Some.  A node is skipped only when the ray
        // misses its box grown by the ray's margin (below): tens to hundreds of times the rounding error of
        // the exact tests at the ray's distance from the farthest instance.  Rays that are not finite or not unit length
        // are not culled at all.
        const bool cull = cull_ok(ray_.o, ray_.dd);         // (the ray as given: F_IDENT's ray_i only serves the exact tests)
        const CullRay R = cull_ray(ray_.o, ray_.d);
        const float *N0 = F + P.off_bvh;
        u32 node = P.n_bvh_nodes ? 0u : BVH_END;
        // ONE margin per ray, from the root box (node 0): k x its extent seen from the origin + ksq x that extent squared (spheres)
        // + kpos x the coordinate magnitudes, so that a step is the triangle BVH's: t = c * inv - o * inv -+ (h * |inv| + mg * |inv|)
        V3 oinv, qm;
        float mg;
        {
            const F4 ra = ld4(N0, 0), rb = ld4(N0, 4);
            const V3 r = sub(v3(ra.x, ra.y, ra.z), R.o);
            const float ext = fmax_(fmax_(fabs_(r.x) + ra.w, fabs_(r.y) + rb.x), fabs_(r.z) + rb.y);
            const float obig = fmax_(fmax_(fabs_(ray_.o.x), fabs_(ray_.o.y)), fabs_(ray_.o.z));
            // (spheres: sqrt(r^2 + 2 eps' D^2) - r <= min(eps' D^2 / r, sqrt(2 eps') D) -- the square law up to D / r = 1000, 4e-3 D beyond)
            const float ksq = fmin_(P.inst_ksq * ext, 4e-3f) * MRT_MARGIN_SCALE;
            mg = fma_fast(P.inst_k * MRT_MARGIN_SCALE + ksq, ext, fma_fast(P.inst_kpos * MRT_MARGIN_SCALE, obig + obig + ext, 1e-6f * MRT_MARGIN_SCALE));
            oinv = hadam(R.o, R.inv);
            qm = muls(R.ainv, mg);
        }
        // "while-while": every lane first walks boxes until it stands on a leaf (cheap iterations, all lanes busy), then
        // the wavefront runs the expensive exact tests of the leaves together.
        for (;;) {
            u32 leaf_a = 0u, leaf_b = 0u;          // one postponed leaf, as in the triangle BVH
            bool walking = node != BVH_END;        // branch-free step: the loop's only branch is its exit
            // nothing in a node whose near side lies beyond the current closest hit can win; the bound only changes in
            // the exact tests between two box walks
            const float far = (!ANY && best.rend >= 0 && best.t0 >= 0.0f) ? best.t0 + 1e-3f * best.t0 + mg : kInf;
            while (walking) {
                const F4 na = ld4(N0, node * BVH_WORDS), nb = ld4(N0, node * BVH_WORDS + 4);
                MRT_COUNT(CT_BVH_NODE);
                const u32 skip = f2u(nb.z), leaf = f2u(nb.w);
                const float px = fma_fast(na.x, R.inv.x, -oinv.x), py = fma_fast(na.y, R.inv.y, -oinv.y), pz = fma_fast(na.z, R.inv.z, -oinv.z);
                const float qx = fma_fast(na.w, R.ainv.x, qm.x), qy = fma_fast(nb.x, R.ainv.y, qm.y), qz = fma_fast(nb.y, R.ainv.z, qm.z);
                const float tn = fmax_(fmax_(px - qx, py - qy), pz - qz);
                const float tf = fmin_(fmin_(px + qx, py + qy), pz + qz);
                bool hit_node = !(tn > tf || tf < 0.0f);
                if (!ANY) hit_node = hit_node && !(tn > far);
                hit_node = hit_node || !cull;      // rays that must not be culled visit everything
                const u32 cand = hit_node ? leaf : 0u;             // a leaf to test, or 0
                const bool have_a = leaf_a != 0u;
                leaf_b |= have_a ? cand : 0u;
                leaf_a |= have_a ? 0u : cand;
                node = (hit_node && leaf == 0u) ? node + 1u : skip;
                walking = node != BVH_END && leaf_b == 0u;
            }
            if (leaf_a == 0u) break;
            const u32 cnt_a = leaf_a >> 24, first_a = leaf_a & 0xffffffu, cnt_b = leaf_b >> 24, first_b = leaf_b & 0xffffffu;
            for (u32 k = 0; k < cnt_a + cnt_b; ++k) {
                const u32 i = ldu(F, P.off_bvhinst + (k < cnt_a ? first_a + k : first_b + (k - cnt_a)));
                if (consider(i, ld4(I, i * INST_WORDS), ld4(I, i * INST_WORDS + 4), AnyOrder(), PerLane())) return true;
            }
        }
    }
    if (best.rend < 0) return false;
    if constexpr ((FEAT & F_BVH) == 0 && MRT_T0_FROM_KEY) {
        // total_key is its own inverse (the sign bit stays); a NaN distance comes back as a NaN (payloads are not part of the contract)
        const i32 k = best_key;
        best.t0 = u2f((u32)(k ^ (i32)(((u32)(k >> 31)) >> 1)));
    }
    best.rend = (i32)ldu(S.F, P.off_instx + best.inst * INSTX_WORDS + INSTX_REND);
    return true;
}

// Object-space image of a world-space hit point, src/rt.rs:782, 798
struct Obj {
    const float *R, *IX, *X;
    V3 pos;
    bool ident;
    u32 kind;
};
MRT_HD Obj obj_of(const Scn &S, const Hit &h)
{
    Obj o;
    const float *I = S.F + S.P->off_inst + h.inst * INST_WORDS;
    const u32 tag = ldu(I, INST_TAG);
    o.R = S.F + S.P->off_rend + (u32)h.rend * REND_WORDS;
    o.IX = S.F + S.P->off_instx + h.inst * INSTX_WORDS;
    o.X = S.F + S.P->off_xf + (tag >> TAG_XF_SHIFT);
    o.pos = ld3(I, INST_POS);
    o.ident = (tag & TAG_IDENT) != 0;
    o.kind = tag & TAG_KIND_MASK;
    return o;
}
MRT_HD V3 to_object(const Obj &o, V3 hp) { return add(o.pos, xf_vec(o.X, o.ident, sub(hp, o.pos))); }

// Renderer::normal, src/rt.rs:776-793 with the Normal impls, src/rt.rs:414-466
template <u32 FEAT>
MRT_HD V3 hit_normal(const Scn &S, const Obj &o, V3 n_hit, i32 tri_idx)
{
    if (o.kind == KIND_PLANE) return ld3(o.IX, INSTX_PLANE_NW);          // norm(R*(L*n)) is per instance
    V3 n = v3(0.0f, 0.0f, 0.0f);
    if (o.kind == KIND_SPHERE) n = sub(n_hit, o.pos);
    else if ((FEAT & F_BOX) && o.kind == KIND_BOX) n = box_normal(ld3(o.R, REND_GEO + 3), n_hit, o.pos);
    else if ((FEAT & F_TRI) && o.kind == KIND_TRIANGLE) n = cross(ld3(o.R, REND_GEO + 3), ld3(o.R, REND_GEO + 6));
    else if (FEAT & F_TRI) {
        const float *M = S.F + S.P->off_mesh + ldu(o.R, REND_GEO) * MESH_WORDS;
        const float *T = ((FEAT & F_DEEP) ? S.G : S.F) + S.P->off_tri + (ldu(M, MESH_TRI0) + (u32)tri_idx) * TRI_WORDS;
        n = cross(ld3(T, 3), ld3(T, 6));
    }
    return norm(xf_vec(o.X, o.ident, n));
}

// Renderer::to_uv, src/rt.rs:795-809 with the UV impls, src/rt.rs:468-542 (triangle / mesh maps are
// rejected by mrt_create: the reference hits todo!())
MRT_HD UV hit_uv(const Obj &o, V3 n_hit)
{
    UV r;
    if (o.kind == KIND_SPHERE) {
        const V3 v = norm(sub(n_hit, o.pos));
        r.x = 0.5f + div_(0.5f * atan2_(v.x, -v.y), kPi);
        r.y = 0.5f - 0.5f * v.z;
    } else if (o.kind == KIND_PLANE) {
        r.x = fract_(n_hit.x + 0.5f);
        if (r.x < 0.0f) r.x = 1.0f + r.x;
        r.y = fract_(n_hit.y + 0.5f);
        if (r.y < 0.0f) r.y = 1.0f + r.y;
    } else if (o.kind == KIND_BOX) {
        r = box_uv(ld3(o.R, REND_GEO + 3), n_hit, o.pos);
    } else {
        r.x = 0.0f; r.y = 0.0f;
    }
    return r;
}

// RayHit::get_* / Renderer::get_*, src/rt.rs:592-616, 811-863
struct Surf {
    const float *M;
    bool maps;
    UV uv;
};
template <u32 FEAT>
MRT_HD Surf surf_of(const Scn &S, const Hit &h, const Obj &o, V3 n_hit)
{
    Surf s;
    s.M = S.F + S.P->off_mat + (u32)h.rend * MAT_WORDS;
    s.maps = false;
    s.uv.x = 0.0f; s.uv.y = 0.0f;
    if constexpr (FEAT & F_MAPS) {
        s.maps = (ldu(o.R, REND_FLAGS) & RF_HAS_MAPS) != 0;
        if (s.maps) s.uv = hit_uv(o, n_hit);
    }
    return s;
}
template <u32 FEAT>
MRT_HD float surf_scalar(const Scn &S, const Surf &s, u32 slot, u32 field)
{
    if (s.maps) {
        const i32 id = (i32)ldu(s.M, MAT_MAP + slot);
        if (id >= 0) return tex_fetch<FEAT>(S, id, s.uv).x;
    }
    return s.M[field];
}
template <u32 FEAT>
MRT_HD V3 surf_color(const Scn &S, const Surf &s)
{
    const V3 albedo = ld3(s.M, MAT_ALBEDO);
    if (s.maps) {
        const i32 id = (i32)ldu(s.M, MAT_MAP + MAP_TEX);
        if (id >= 0) return hadam(albedo, tex_fetch<FEAT>(S, id, s.uv));
    }
    return albedo;
}

// RayTracer::rand, src/rt.rs:996-1007.  Math contract v3: cos th = 1 - 2 u1 and sin th = sqrt(4 u1 (1 - u1)) directly instead
// of sin / cos of acos(1 - 2 u1) (both factors exact, one rounding, one correctly rounded root: see oracle/mrt_oracle.c
// rt_rand and DESIGN.md section 4) -- an acos and a sincos less per scatter.
MRT_HD V3 rand_normal(V3 n, float r, float u1, float u2)
{
    const float phi = u2 * 2.0f * kPi;
    const float cth = 1.0f - 2.0f * u1;
    const float sth = sqrt_((4.0f * u1) * (1.0f - u1));
    float sphi, cphi;
    sincos_(phi, sphi, cphi);
    const V3 v = v3(sth * cphi, sth * sphi, cth);
    return norm(add(n, muls(v, r)));
}

// Vec3f::refract, src/lin.rs:96-105
MRT_HD bool refract(V3 d, float eta, V3 n, V3 &out)
{
    const float cosv = dot(neg(n), d);
    const float k = 1.0f - (eta * eta) * (1.0f - cosv * cosv);
    if (k < 0.0f) return false;
    out = add(muls(d, eta), muls(n, cosv * eta + sqrt_(k)));
    return true;
}

// RayTracer::iter + the pixel-only half of RayTracer::cast, src/rt.rs:937-947, 900-914
MRT_HD V3 pixel_focus(const Params &P, float cx, float cy)
{
    const float uvx = div_(P.aspect * (cx - 0.5f * P.w), P.w);
    const float uvy = div_(cy - 0.5f * P.h, P.h);
    const V3 dir = norm(v3(uvx, P.inv2tan, -uvy));
    const V3 cam = v3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    const V3 orig = add(cam, muls(dir, kE));           // Ray::cast_default
    return add(orig, muls(dir, P.foc));                // Vec3f::from(&ray) with t = foc
}

// the per-sample half of RayTracer::cast, src/rt.rs:916-931
// lens position of a sample: the two draws of src/rt.rs:916-920
MRT_HD V3 lens_pos(const Params &P, u32 pk)
{
    const float u1 = u32_to_unit(draw_u32(pk, DIM_LENS_X));
    const float u2 = u32_to_unit(draw_u32(pk, DIM_LENS_Z));
    return v3(P.cam_pos[0] + (u1 - 0.5f) * P.aprt, P.cam_pos[1], P.cam_pos[2] + (u2 - 0.5f) * P.aprt);
}

MRT_HD void camera_ray(const Params &P, const float *camF, V3 focus, u32 pk, V3 &o, V3 &d)      // camF: cam_L, cam_R in the blob
{
    const float u1 = u32_to_unit(draw_u32(pk, DIM_LENS_X));
    const float u2 = u32_to_unit(draw_u32(pk, DIM_LENS_Z));
    const V3 pos = v3(P.cam_pos[0] + (u1 - 0.5f) * P.aprt, P.cam_pos[1], P.cam_pos[2] + (u2 - 0.5f) * P.aprt);
    const V3 new_dir = norm(sub(focus, pos));
    // rot_y * (look * new_dir): both matrices are the identity as values for the default camera direction
    if (P.cam_ident && nzfin3(new_dir)) d = new_dir;
    else d = m3mul(camF + 9, m3mul(camF, new_dir));
    o = add(pos, muls(d, kE));
}

// gen_bool(0.80) of src/rt.rs:564,579 takes the f64 literal 0.80: the threshold is floor(0.8 * 2^32), not the f32 0.8
constexpr u32 kThr080 = 3435973836u;

MRT_HD u32 dim_of(u32 bounce, u32 slot) { return DIM_BOUNCE0 + bounce * DIMS_PER_BOUNCE + slot; }

// Bernoulli draw that only touches the generator when the outcome is open (p == 0 is always false, p == 1 always
// true: same results as bernoulli(p, draw), fewer hashes)
MRT_HD bool coin(float p, u32 pk, u32 dim)
{
    if (p == 1.0f) return true;
    if (!(p > 0.0f)) return false;          // p == 0 (or an invalid p, which mrt_create rejects)
    return draw_u32(pk, dim) < (u32)(p * 4294967296.0f);
}

// Per-lane state that is only touched between segments (path radiance, throughput, the pixel's camera focus point
// and chunk bookkeeping).  RegStash keeps it in registers; LdsStash parks it in a per-lane LDS column (slot-major, so
// lane i always hits bank i) to free VGPRs for the traversal loop without the compiler spilling to scratch,
// whose write-backs would show up as HBM traffic.  volatile: the values must really live in LDS across the loop.
enum : u32 { ST_FOCUS = 0, ST_PIXKEY = 3, ST_CHUNK = 4, ST_SEND = 5, ST_WORD = 6, ST_SLOTS = 7, ST_P0Y = 7, ST_P0Z = 8, ST_PK = 9, ST_SLOTS_HIT = 10, ST_T = 10, ST_L = 13, ST_SLOTS_TL = 16 };
// The kernels bound to 6 waves per SIMD (80 VGPRs: instance BVH, no mesh code, warm staging) also park the path's throughput
// and radiance (T, L: touched between segments only) in the stash instead of leaving them to the register allocator's spills.
// (The 8-wave plane / sphere kernel of the 256-thread shape, 64 VGPRs + 24 B of scratch, gains nothing from it: 7949 vs 7987.)
constexpr bool tl_in_stash(u32 feat, u32 /*threads*/) { return (feat & F_COLD) && (feat & F_BVH) && !(feat & F_TRI); }
// Those kernels also park what a shaded hit needs again after its queries: two components of hit.0's point and the path's hash
// key (HBM traffic of the Minecraft-shaped scene 3.6 -> 1.7 GB per 32-spp launch: they were spilled to scratch).  (The warm
// mesh kernels, 4 waves per SIMD at 128 VGPRs, gain nothing from the same three slots: 3421-3427 against 3403-3425 Msamples/s.)
constexpr bool hit_in_stash(u32 feat, u32 threads) { return tl_in_stash(feat, threads); }
constexpr u32 stash_slots_for(u32 feat, u32 threads) { return tl_in_stash(feat, threads) ? (u32)ST_SLOTS_TL : (u32)ST_SLOTS; }
struct RegStash {
    static constexpr bool in_lds = false;
    static constexpr u32 threads = 0;
    float v[ST_SLOTS_TL];
    MRT_HD void put(u32 slot, float x) { v[slot] = x; }
    MRT_HD float get(u32 slot) const { return v[slot]; }
};
#if defined(__HIPCC__) || defined(__HIP__)
template <u32 THREADS>
struct LdsStash {
    static constexpr bool in_lds = true;
    static constexpr u32 threads = THREADS;
    lds_vfloat *base;         // &lds_stash[tid]; slot-major columns of THREADS floats
    MRT_HD void put(u32 slot, float x) { base[slot * THREADS] = x; }
    MRT_HD float get(u32 slot) const { return base[slot * THREADS]; }
};
#endif
template <class St> MRT_HD void st_put3(St &st, u32 slot, V3 v) { st.put(slot, v.x); st.put(slot + 1, v.y); st.put(slot + 2, v.z); }
template <class St> MRT_HD V3 st_get3(const St &st, u32 slot) { return v3(st.get(slot), st.get(slot + 1), st.get(slot + 2)); }

// Samples are accumulated in a canonical order that does not depend on how the work is split: the global sample
// indices are cut into aligned chunks of kChunk; a chunk is summed from 0 in index order, and chunk sums are added to
// the pixel's accumulator in chunk order.  A lane owns every k_split-th chunk of a launch (k_split = 1: all of them,
// added straight to the accumulator; k_split > 1: chunk sums go to `partial` and reduce_chunks adds them in order),
// so small frames can be spread over more wavefronts with bit-identical results.

struct LaneJob {
    u32 k;                 // this lane's chunk phase, 0 <= k < P.k_split
    u32 word;              // 3 * (shard-local pixel index): accum[word..word+2] is this pixel (k_split == 1, read and
                           // written in place); partial[chunk * partial_stride + word ..] receives chunk sums otherwise
};

// All samples of one supersampled pixel: Sampler::execute's per-pixel body, n_samples times
// (src/sampler.rs:45-70 calling RayTracer::iter / reduce_light, src/rt.rs:937-994, whose iterator
// is RaytraceIterator::next, src/rt.rs:1014-1066).
template <u32 FEAT, class Stash>
MRT_HD void render_pixel(const Scn &S, Stash &st, u32 x, u32 y, const LaneJob &job, u32 &segments, unsigned long long *ticks = nullptr)
{
    MRT_TICK_DECL;
    const Params &P = *S.P;
    const u32 pixel = y * P.nw + x;
    const u32 pix_key = mix32(pixel + P.seed_lo) ^ P.seed_hi;      // path_key = mix32(pix_key + sample * kGold)
    const V3 sky_init = v3(P.sky_init[0], P.sky_init[1], P.sky_init[2]);
    const u32 s_base = P.sample_base, s_stop = P.sample_base + P.n_samples;
    const u32 g0 = s_base / kChunk;
    const u32 n_chunks = (s_stop - 1u) / kChunk - g0 + 1u;          // n_samples > 0
    // look-ahead launches of the per-call path (Params.to_planes, one lane per pixel): every sample is a chunk of its own and
    // goes to a plane of its own -- plane j holds sample s_base + j -- and the accumulator is not touched
    const bool planes1 = P.to_planes != 0u;
    const bool direct = P.k_split == 1u && !planes1;
    st_put3(st, ST_FOCUS, pixel_focus(P, (float)x, (float)y));

    // chunk bookkeeping and the pixel's hash key are only needed when a path starts or a chunk ends: stashed too
    st.put(ST_PIXKEY, u2f(pix_key));
    st.put(ST_WORD, u2f(job.word));
    st.put(ST_CHUNK, u2f(job.k));                                   // local chunk index of this lane
    u32 s = (g0 + job.k) * kChunk;                                  // global sample index
    if (s < s_base || planes1) s = s_base;
    {
        u32 e = (g0 + job.k + 1u) * kChunk;
        if (e > s_stop) e = s_stop;
        if (planes1) e = s_base + 1u;
        st.put(ST_SEND, u2f(e));
    }
    bool alive = planes1 ? job.k == 0u : job.k < n_chunks;
    V3 csum = v3(0.0f, 0.0f, 0.0f);
    u32 pk = 0, b = 0;
    V3 o = v3(0, 0, 0), d = v3(0, 1, 0);
    // path throughput T and radiance L: registers, or (tl_in_stash) two stash columns
    constexpr bool kTL = Stash::in_lds && tl_in_stash(FEAT, Stash::threads);
    constexpr bool kHit = Stash::in_lds && hit_in_stash(FEAT, Stash::threads);
    V3 T_ = v3(1, 1, 1), L_ = v3(0, 0, 0);
    auto getT = [&]() { if constexpr (kTL) return st_get3(st, ST_T); else return T_; };
    auto getL = [&]() { if constexpr (kTL) return st_get3(st, ST_L); else return L_; };
    auto setT = [&](V3 v) { if constexpr (kTL) st_put3(st, ST_T, v); else T_ = v; };
    auto setL = [&](V3 v) { if constexpr (kTL) st_put3(st, ST_L, v); else L_ = v; };
    setT(v3(1, 1, 1)); setL(v3(0, 0, 0));
    float pwr = 1.0f;
    u32 seg = 0;
    if (alive) {                                 // first sample of this lane: every lane of the wavefront is here
        pk = mix32(pix_key + s * kGold);
        if constexpr (kHit) st.put(ST_PK, u2f(pk));
        camera_ray(P, S.F + P.off_cam, st_get3(st, ST_FOCUS), pk, o, d);
    }

    // ONE flat loop: an iteration traces one segment; a lane whose path ends draws its next sample's lens position in
    // the same iteration, and the direction of the next ray -- scattered or fresh from the camera -- goes through ONE
    // shared normalisation at the bottom.  No lane ever waits for its neighbours' paths to end, and the divergent
    // regeneration block holds only two hashes and a few adds.
    while (alive) {
        MRT_PROBE(PH_ITER);
        // ---- RaytraceIterator::next ----
        const RayPre ray = ray_pre<FEAT>(o, d);
        Hit h;
        ++seg;
        V3 contrib = v3(0.0f, 0.0f, 0.0f);
        bool ended = false;
        V3 X = v3(0.0f, 1.0f, 0.0f);             // un-normalised direction of the next ray
        V3 base = o;                             // the point it leaves from (hit point, or lens position)
        MRT_TICK(3);                                                   // loop bottom -> here: ray set-up
        const bool hit_any = trace<false, FEAT>(S, ray, h);
        MRT_TICK(0);                                                   // the closest-hit query
        if (!hit_any) {
            // primary miss: raw sky colour (src/rt.rs:957-959); otherwise the fold starts from sky*pwr (:964)
            contrib = (b == 0) ? v3(P.sky[0], P.sky[1], P.sky[2]) : add(getL(), hadam(getT(), sky_init));
            ended = true;
        } else {
            MRT_PROBE(PH_SHADE);
            const Obj ob = obj_of(S, h);
            // (6-wave kernels: the path's hash key lives in the lane stash across the queries and is read once per shaded hit)
            if constexpr (kHit) pk = f2u(st.get(ST_PK));
            const V3 p0_ = add(o, muls(d, h.t0));                      // Vec3f::from(&hit.0.ray)
            // hit.0's point is needed again by the scatter and by every light: the 6-wave kernels park two of its components in
            // the lane stash instead of leaving them to the register allocator's spills (scratch traffic past L2)
            if constexpr (kHit) { st.put(ST_P0Y, p0_.y); st.put(ST_P0Z, p0_.z); }
            auto p0 = [&]() { if constexpr (kHit) return v3(p0_.x, st.get(ST_P0Y), st.get(ST_P0Z)); else return p0_; };
            const V3 nh0 = to_object(ob, p0_);
            const Surf sf0 = surf_of<FEAT>(S, h, ob, nh0);
            const float opacity0 = surf_scalar<FEAT>(S, sf0, MAP_OPACITY, MAT_OPACITY);
            const float metal_c = sf0.M[MAT_METAL];                     // hit.obj.mat.metal, not the map (src/rt.rs:564)

            // Scatter.  The reference always builds the reflected ray at hit0 (src/rt.rs:1049) and replaces it by the
            // refracted ray from the exit hit when the 15 % / opacity coin comes up and refraction is possible
            // (src/rt.rs:1054-1058).  Draws are slot-addressed, so only the ray that survives is computed: lanes that
            // try to refract and lanes that reflect share one pass through the normal / perturbation code; a failed
            // refraction (total internal reflection) takes a second pass as a reflection.
            bool refr = coin(fmin_(1.0f - opacity0, 0.85f), pk, dim_of(b, SL_OPAC_COIN));
            V3 hp, hn;
            Surf sfh;
            u32 pass = 0;
            for (;;) {
                if (pass == 0) { MRT_PROBE(PH_SCATTER1); } else { MRT_PROBE(PH_SCATTER2); }
                ++pass;
                if (refr) { MRT_PROBE(PH_REFRACT); }
                if (ob.kind != KIND_PLANE) { MRT_PROBE(PH_NORMAL_NONPLANE); }
                hp = refr ? add(o, muls(d, h.t1)) : p0();               // recorded hit point: hit.1 or hit.0
                const V3 nhh = refr ? to_object(ob, hp) : nh0;
                hn = hit_normal<FEAT>(S, ob, nhh, refr ? h.i1 : h.i0);
                sfh = sf0;
                if (refr) sfh = surf_of<FEAT>(S, h, ob, nhh);
                float rough = surf_scalar<FEAT>(S, sfh, MAP_ROUGH, MAT_ROUGH);            // Ray::reflect / Ray::refract, src/rt.rs:559-589
                const float opac = refr ? surf_scalar<FEAT>(S, sfh, MAP_OPACITY, MAT_OPACITY) : opacity0;
                const u32 dbase = dim_of(b, refr ? SL_REFR_COIN : SL_REFL_COIN);    // coin, u1, u2 are consecutive slots
                if (metal_c == 0.0f && opac != 0.0f && draw_u32(pk, dbase) < kThr080) rough = 1.0f;
                const V3 nn = rand_normal(hn, rough, u32_to_unit(draw_u32(pk, dbase + 1u)), u32_to_unit(draw_u32(pk, dbase + 2u)));
                if (refr) {
                    const float eta = 1.0f + 0.5f * surf_scalar<FEAT>(S, sfh, MAP_GLASS, MAT_GLASS);
                    if (refract(d, eta, nn, X)) break;                  // .norm() of src/rt.rs:586 happens at the bottom
                    refr = false;                                       // Vec3f::refract returned None
                    continue;
                }
                X = reflect(d, nn);                                     // .norm() of src/rt.rs:569 happens at the bottom
                break;
            }
            base = hp;

            // emit coin of the fold, src/rt.rs:966-970: replaces everything behind this hit
            const V3 color = surf_color<FEAT>(S, sfh);
            const float emit = surf_scalar<FEAT>(S, sfh, MAP_EMIT, MAT_EMIT);
            if (coin(emit, pk, dim_of(b, SL_EMIT_COIN))) {
                MRT_PROBE(PH_EMIT_END);
                contrib = add(getL(), hadam(getT(), color));
                ended = true;
            } else {
                // direct light, visibility from hit0 (src/rt.rs:1027-1046), shading at the recorded hit (:973-987)
                if ((FEAT & F_LIGHTS) && P.n_light) {
                    V3 l_col = v3(0.0f, 0.0f, 0.0f);
                    const float rough_h = surf_scalar<FEAT>(S, sfh, MAP_ROUGH, MAT_ROUGH);
                    const float metal_h = surf_scalar<FEAT>(S, sfh, MAP_METAL, MAT_METAL);
                    for (u32 li = 0; li < P.n_light; ++li) {
                        const float *Lt = S.F + P.off_light + li * LIGHT_WORDS;
                        const bool point = ldu(Lt, LIGHT_KIND) == LK_POINT;
                        const V3 lv = ld3(Lt, LIGHT_V);
                        // The light's term of the fold (src/rt.rs:973-987), evaluated before its shadow ray: when it is exactly zero
                        // (surface facing away and no highlight: diff == 0, spec == 0) the light's visibility cannot change l_col
                        // -- l_col is never -0, so adding +-0 leaves every bit -- and the shadow ray (src/rt.rs:1027-1045, a
                        // whole any-hit traversal) is not traced.  A NaN anywhere in the term compares unequal to zero and takes
                        // the reference's route.
                        const V3 ln = point ? norm(sub(lv, hp)) : lv;             // l.norm() at the recorded hit
                        const float diff = fmax_(dot(ln, hn), 0.0f);
                        const float sp = fmax_(dot(d, reflect(ln, hn)), 0.0f);
                        const float s2 = sp * sp, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8, s32 = s16 * s16;   // powi(32)
                        const float spec = s32 * (1.0f - rough_h);
                        const V3 o_col = muls(color, 1.0f - metal_h);
                        V3 t = hadam(muls(o_col, diff), ld3(Lt, LIGHT_COLOR));
                        t = v3(t.x + spec, t.y + spec, t.z + spec);
                        const V3 term = muls(t, Lt[LIGHT_PWR]);
                        if (term.x == 0.0f && term.y == 0.0f && term.z == 0.0f) continue;
                        const V3 ph0 = p0();
                        const V3 ls = point ? norm(sub(lv, ph0)) : lv;            // l.norm() at hit0
                        const V3 so = add(ph0, muls(ls, kE));                     // Ray::cast_default
                        Hit hs;
                        MRT_TICK(1);                                   // shading up to the shadow query (lanes that ask one)
                        const bool blocked = trace<true, FEAT>(S, ray_pre<FEAT>(so, ls), hs);
                        MRT_TICK(2);                                   // the shadow query
                        if (blocked) continue;
                        l_col = add(l_col, term);
                    }
                    // the fold step (d_col + l_col) * pwr, src/rt.rs:990-992, front to back
                    setL(add(getL(), hadam(getT(), muls(l_col, pwr))));
                }
                setT(hadam(getT(), muls(v3(0.5f + color.x, 0.5f + color.y, 0.5f + color.z), pwr)));
                pwr = pwr * P.q;                                        // Ray::cast, src/rt.rs:571
                ++b;
                if (b > P.bounce) {          // src/rt.rs:1018
                    contrib = add(getL(), hadam(getT(), sky_init));
                    ended = true;
                }
            }
        }
    
        MRT_TICK(1);                                                   // shading (rest)
        bool from_camera = false;
        if (ended) {
            csum = add(csum, contrib);
            ++s;
            if (s == f2u(st.get(ST_SEND))) {                        // chunk complete: flush its sum
                u32 j = f2u(st.get(ST_CHUNK));
                if (direct) {
                    // one lane per pixel: the chunk sum goes straight into the accumulator (chunk order = program order; 24 B of
                    // traffic per pixel per 16 samples instead of three stash slots held for the whole launch)
                    float *q = P.accum + f2u(st.get(ST_WORD));
                    q[0] += csum.x; q[1] += csum.y; q[2] += csum.z;
                } else {
                    float *q = P.partial + ((size_t)j * P.partial_stride + f2u(st.get(ST_WORD)));
                    q[0] = csum.x; q[1] = csum.y; q[2] = csum.z;
                }
                csum = v3(0.0f, 0.0f, 0.0f);
                u32 e;
                if (planes1) {                                      // (wave-uniform) the next sample, the next plane
                    j += 1u;
                    alive = s < s_stop;
                    e = s + 1u;
                } else {
                    j += P.k_split;
                    alive = j < n_chunks;
                    s = (g0 + j) * kChunk;
                    e = s + kChunk;
                    if (e > s_stop) e = s_stop;
                }
                st.put(ST_CHUNK, u2f(j));
                st.put(ST_SEND, u2f(e));
            }
            if (alive) {                                            // next sample: RayTracer::cast, src/rt.rs:916-922
                MRT_PROBE(PH_REGEN);
                pk = mix32(f2u(st.get(ST_PIXKEY)) + s * kGold);
                if constexpr (kHit) st.put(ST_PK, u2f(pk));
                base = lens_pos(P, pk);
                X = sub(st_get3(st, ST_FOCUS), base);               // new_dir before .norm()
                setT(v3(1.0f, 1.0f, 1.0f)); setL(v3(0.0f, 0.0f, 0.0f));
                pwr = 1.0f; b = 0;
                from_camera = true;
            }
        }
        // next ray: dir = X.norm(), orig = base + dir * E  (Ray::cast src/rt.rs:551-553; cast_default :555-557)
        V3 nd = norm(X);
        if (from_camera && !(P.cam_ident && nzfin3(nd))) nd = m3mul(S.F + P.off_cam + 9, m3mul(S.F + P.off_cam, nd));   // rot_y * (look * new_dir), src/rt.rs:930
        o = add(base, muls(nd, kE));
        d = nd;
    }
    segments = seg;
    (void)ticks;
    MRT_TICK_OUT(ticks);
}

}  // namespace mrt
