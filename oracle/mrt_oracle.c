/*
 * mrt_oracle.c — TEST INFRASTRUCTURE: CPU restatement of the reference's path tracer.
 * See mrt_oracle.h for the rules about who may use this and for the parity-pin status.
 *
 * Every function cites the reference lines it restates (paths relative to the reference
 * checkout).  f32 arithmetic follows the reference's operation order (the one exception is D8);
 * compile with -ffp-contract=off.  Deliberate, documented differences from the Rust binary:
 *   D1  RNG: seeded counter RNG instead of rand::thread_rng (unseedable).
 *   D2  sin/cos/acos/atan2/powf: the math contract of oracle_math.h instead of libm.
 *   D3  reduce_light evaluates the path once (the reference traces it twice, src/rt.rs:957-961;
 *       only the second evaluation reaches the result).
 *   D4  Texture index is clamped to the last texel (reference: slice panic, src/rt.rs:624).
 *   D5  NaN ordering keys compare as -NaN (x86 default-NaN sign).
 *   D6  Pixels of the over-covering tiles beyond nw x nh (src/sampler.rs:32-33,45-48) are not
 *       computed: Sampler::img never reads them.
 *   D7  Inputs the reference would panic on are rejected by orc_create.
 *   D8  Math contract v3: RayTracer::rand's polar angle (src/rt.rs:997-1003) is NOT taken as acos, then sin / cos:
 *       rt_rand computes cos th = 1 - 2 u1 (exact on the 2^-23 lattice of u1) and sin th = sqrt(4 u1 (1 - u1)).
 *       Bounded by tests/test_oracle_math.py::test_contract_v3_polar_angle over every lattice value of u1:
 *       <= 3e-7 from the literal om_acosf -> om_sincosf composition, <= 6e-8 from the f64 truth.
 *       RULE: the restatement of reference arithmetic changes only together with a committed
 *       bound of this kind and a re-run of all five doc/out*.png pins.
 */
#define _GNU_SOURCE
#include "mrt_oracle.h"
#include "oracle_math.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define E_ 0.0001f /* src/rt.rs:7 */

/* ------------------------------------------------------------------ lin.rs */
typedef struct { float x, y, z; } v3;
typedef struct { float w, x, y, z; } v4;
typedef struct { float m[9]; } m3;
typedef struct { float x, y; } v2;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }      /* lin.rs:211-221 */
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }      /* lin.rs:247-257 */
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }     /* lin.rs:259-264 */
static inline v3 v3_muls(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }        /* lin.rs:266-275 */
static inline v3 v3_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }                           /* lin.rs:304-314 */
static inline v3 v3_hadam(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }    /* lin.rs:107-113 */
static inline v3 v3_cross(v3 a, v3 b)                                                    /* lin.rs:52-58 */
{
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3_mag(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }    /* lin.rs:60-62 */
static inline float f_recip(float x) { return 1.0f / x; }
static inline v3 v3_norm(v3 a) { return v3_muls(a, f_recip(v3_mag(a))); }                /* lin.rs:64-66 */
static inline v3 v3_reflect(v3 d, v3 n) { return v3_sub(d, v3_muls(n, 2.0f * v3_dot(d, n))); } /* lin.rs:68-70 */
static inline v3 v3_recip(v3 a) { return V3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }       /* lin.rs:72-78 */
static inline v3 v3_abs(v3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }         /* lin.rs:80-86 */
static inline v4 v4_neg(v4 a) { v4 r = {-a.w, -a.x, -a.y, -a.z}; return r; }             /* lin.rs:445-456 */

/* f32::max / f32::min (maxNum / minNum): a NaN operand yields the other one.  IEEE leaves the
 * result for a (+0, -0) pair open; the contract fixes max -> +0, min -> -0. */
static inline float f_max(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (om_f2u(a) & 0x80000000u) ? b : a;
    return a < b ? b : a;
}
static inline float f_min(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (om_f2u(a) & 0x80000000u) ? a : b;
    return b < a ? b : a;
}

/* Vec3f::refract, lin.rs:96-105 */
static inline int v3_refract(v3 d, float eta, v3 n, v3 *out)
{
    float cosv = v3_dot(v3_neg(n), d);
    float k = 1.0f - (eta * eta) * (1.0f - cosv * cosv);
    if (k < 0.0f) return 0;
    *out = v3_add(v3_muls(d, eta), v3_muls(n, cosv * eta + sqrtf(k)));
    return 1;
}

/* Mat3f::rotate_y, lin.rs:175-183 */
static inline m3 m3_rotate_y(v4 dir)
{
    float cw = sqrtf(1.0f - dir.w * dir.w);
    m3 r = {{cw, 0.0f, dir.w, 0.0f, 1.0f, 0.0f, -dir.w, 0.0f, cw}};
    return r;
}

/* Mat4f::lookat (upper-left 3x3 is all that Mat4f * Vec3f uses), lin.rs:197-208, 356-365 */
static inline m3 m3_lookat(v4 dir, v3 up)
{
    v3 fwd = v3_norm(V3(dir.x, dir.y, dir.z));
    v3 right = v3_norm(v3_cross(fwd, up));
    v3 n_up = v3_cross(right, fwd);
    m3 r = {{right.x, -right.y, right.z, -fwd.x, fwd.y, -fwd.z, n_up.x, -n_up.y, n_up.z}};
    return r;
}

/* Mat * Vec3f, lin.rs:344-365 */
static inline v3 m3_mul(const m3 *m, v3 v)
{
    return V3(m->m[0] * v.x + m->m[1] * v.y + m->m[2] * v.z,
              m->m[3] * v.x + m->m[4] * v.y + m->m[5] * v.z,
              m->m[6] * v.x + m->m[7] * v.y + m->m[8] * v.z);
}

/* f32::total_cmp as an integer key, NaN forced to -NaN (D5) */
static inline int32_t total_key(float t)
{
    if (t != t) return INT32_MIN;
    int32_t i = (int32_t)om_f2u(t);
    i ^= (int32_t)(((uint32_t)(i >> 31)) >> 1);
    return i;
}

/* `as usize` (saturating, NaN -> 0), capped so that x + y*w cannot overflow */
static inline uint64_t f_to_index(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 2147483648.0f) return 2147483648ull;
    return (uint64_t)v;
}

/* ------------------------------------------------------------------ RNG contract (D1) */
static inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
#define GOLD 0x9E3779B9u
uint32_t orc_path_key(uint64_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t a = mix32(pixel + (uint32_t)seed) ^ (uint32_t)(seed >> 32);
    return mix32(a + sample * GOLD);
}
uint32_t orc_draw_u32(uint32_t pk, uint32_t dim) { return mix32(pk + (dim + 1u) * GOLD); }
float orc_draw_f32(uint32_t pk, uint32_t dim) { return (float)(orc_draw_u32(pk, dim) >> 9) * 1.1920928955078125e-7f; }

enum { DIM_LENS_X = 0, DIM_LENS_Z = 1, DIM_BOUNCE0 = 2, DIMS_PER_BOUNCE = 8 };
enum { SL_REFL_COIN = 0, SL_REFL_U1, SL_REFL_U2, SL_OPAC_COIN, SL_REFR_COIN, SL_REFR_U1, SL_REFR_U2, SL_EMIT_COIN };
static inline uint32_t dim_of(uint32_t bounce, uint32_t slot) { return DIM_BOUNCE0 + bounce * DIMS_PER_BOUNCE + slot; }

/* gen_bool(0.80) at src/rt.rs:564,579 takes the f64 literal 0.80: threshold floor(0.8 * 2^32), not the f32 0.8 */
#define THR_080 3435973836u

/* rand 0.8.5 Bernoulli: p == 1 always true without a draw; else u < p * 2^N */
static inline int bernoulli(float p, uint32_t u)
{
    if (p == 1.0f) return 1;
    uint32_t thr = (uint32_t)(p * 4294967296.0f);   /* p in [0,1): exact product, truncation */
    return u < thr;
}

/* ------------------------------------------------------------------ scene (src/rt.rs:63-190) */
typedef struct { uint32_t w, h; float *dat; } tex_t;
typedef struct {
    v3 albedo; float rough, metal, glass, opacity, emit;
    int32_t map[6];   /* tex rmap mmap gmap omap emap */
} mat_t;
typedef struct { v3 pos; v4 dir; } inst_t;
typedef struct { v3 a, b, c; } tri_t;
typedef struct bvh {
    v3 aabb, rel_pos;
    uint32_t *content; uint32_t n_content;   /* NULL = None */
    struct bvh **childs; uint32_t n_childs;  /* NULL = None */
} bvh_t;
typedef struct {
    uint32_t kind;
    float r; v3 n; v3 sizes; tri_t tri;
    tri_t *mesh; uint32_t n_mesh; bvh_t *bvh; uint32_t leaf_ids_total;
    mat_t mat;
    inst_t *inst; uint32_t n_inst;
} rend_t;
typedef struct { uint32_t kind; v3 v; float pwr; v3 color; } light_t;

struct orc_ctx {
    uint32_t bounce; float loss;
    uint16_t res_w, res_h; float ssaa; mrt_camera cam;
    rend_t *rend; uint32_t n_rend;
    light_t *light; uint32_t n_light;
    v3 sky_color; float sky_pwr;
    tex_t *tex; uint32_t n_tex;
    uint64_t seed;
    uint32_t nw, nh;
    float *colors;        /* [nh][nw][3] */
    uint32_t last_count;
    uint64_t segments;
    uint32_t max_ids;
};

typedef struct { v3 orig, dir; float t, pwr; uint32_t bounce; } ray_t;          /* src/rt.rs:45-52 */
typedef struct { const rend_t *obj; const inst_t *inst; int32_t idx; ray_t ray; v3 norm; } hit_t; /* src/rt.rs:55-61 */

static __thread char g_err[256];
const char *orc_error(void) { return g_err; }

static inline v3 ray_point(const ray_t *r) { return v3_add(r->orig, v3_muls(r->dir, r->t)); }   /* src/rt.rs:193-197 */

/* Ray::cast / cast_default, src/rt.rs:551-557 */
static inline ray_t ray_cast(v3 orig, v3 dir, float pwr, uint32_t bounce)
{
    ray_t r; r.orig = v3_add(orig, v3_muls(dir, E_)); r.dir = dir; r.pwr = pwr; r.bounce = bounce; r.t = 0.0f;
    return r;
}
static inline ray_t ray_cast_default(v3 orig, v3 dir) { return ray_cast(orig, dir, 1.0f, 0); }

/* ------------------------------------------------------------------ primitives */
/* Box::intersect, src/rt.rs:299-333 */
static int box_intersect(v3 sizes, const ray_t *ray, v3 pos, float *t0, float *t1)
{
    v3 m = v3_recip(ray->dir);
    if (isinf(m.x)) m.x = f_recip(E_);
    if (isinf(m.y)) m.y = f_recip(E_);
    if (isinf(m.z)) m.z = f_recip(E_);
    v3 n = v3_hadam(v3_sub(ray->orig, pos), m);
    v3 k = v3_hadam(v3_muls(sizes, 0.5f), v3_abs(m));
    v3 a = v3_sub(v3_neg(n), k);
    v3 b = v3_add(v3_neg(n), k);
    float a0 = f_max(f_max(a.x, a.y), a.z);
    float b1 = f_min(f_min(b.x, b.y), b.z);
    if (a0 > b1 || b1 < 0.0f) return 0;
    *t0 = a0; *t1 = b1;
    return 1;
}

/* Sphere::intersect, src/rt.rs:335-359 */
static int sphere_intersect(float r, const ray_t *ray, v3 pos, float *t0, float *t1)
{
    v3 o = v3_sub(ray->orig, pos);
    float a = v3_dot(ray->dir, ray->dir);
    float b = 2.0f * v3_dot(o, ray->dir);
    float c = v3_dot(o, o) - r * r;
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return 0;
    float q0 = (-b - sqrtf(disc)) / (2.0f * a);
    float q1 = (-b + sqrtf(disc)) / (2.0f * a);
    if (q0 < 0.0f) return 0;
    *t0 = q0; *t1 = q1;
    return 1;
}

/* Triangle::intersect, src/rt.rs:361-398 */
static int tri_intersect(const tri_t *tr, const ray_t *ray, v3 pos, float *tout)
{
    v3 e0 = v3_sub(tr->b, tr->a);
    v3 e1 = v3_sub(tr->c, tr->a);
    v3 p = v3_cross(ray->dir, e1);
    float d = v3_dot(e0, p);
    if (d < E_ && d > -E_) return 0;
    float inv_d = f_recip(d);
    v3 t = v3_sub(ray->orig, v3_add(tr->a, pos));
    float u = v3_dot(t, p) * inv_d;
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = v3_cross(t, e0);
    float v = v3_dot(ray->dir, q) * inv_d;
    if (v < 0.0f || (u + v) > 1.0f) return 0;
    float tt = v3_dot(e1, q) * inv_d;
    if (tt < 0.0f) return 0;
    *tout = tt;
    return 1;
}

/* Plane::intersect, src/rt.rs:400-412 */
static int plane_intersect(v3 n, const ray_t *ray, v3 pos, float *tout)
{
    v3 nn = v3_norm(n);
    float d = v3_dot(v3_neg(nn), pos);
    float t = -(v3_dot(ray->orig, nn) + d) / v3_dot(ray->dir, nn);
    if (t <= 0.0f) return 0;
    *tout = t;
    return 1;
}

static inline int in_range(float lo, float hi, float x) { return lo <= x && x < hi; }   /* Range::contains */

/* Normal for Box, src/rt.rs:414-445 */
static v3 box_normal(v3 sizes, v3 hit, v3 pos)
{
    v3 p = v3_hadam(v3_sub(hit, pos), v3_muls(v3_recip(sizes), 2.0f));
    const float plo = 1.0f - E_, phi = 1.0f + E_, nlo = -1.0f - E_, nhi = -1.0f + E_;
    v3 n = V3(0.0f, 0.0f, 0.0f);
    if (in_range(plo, phi, p.x)) n = V3(1.0f, 0.0f, 0.0f);
    else if (in_range(nlo, nhi, p.x)) n = V3(-1.0f, -0.0f, -0.0f);
    else if (in_range(plo, phi, p.y)) n = V3(0.0f, 1.0f, 0.0f);
    else if (in_range(nlo, nhi, p.y)) n = V3(-0.0f, -1.0f, -0.0f);
    if (in_range(plo, phi, p.z)) n = V3(0.0f, 0.0f, 1.0f);
    else if (in_range(nlo, nhi, p.z)) n = V3(-0.0f, -0.0f, -1.0f);
    return n;
}

/* UV for Box, src/rt.rs:468-516 */
static v2 box_uv(v3 sizes, v3 hit, v3 pos)
{
    v3 p = v3_hadam(v3_sub(hit, pos), v3_muls(v3_recip(sizes), 2.0f));
    const float plo = 1.0f - E_, phi = 1.0f + E_, nlo = -1.0f - E_, nhi = -1.0f + E_;
    v2 r;
    if (in_range(plo, phi, p.x)) { r.x = (0.5f + 0.5f * p.y) / 4.0f + 2.0f / 4.0f; r.y = (0.5f - 0.5f * p.z) / 3.0f + 1.0f / 3.0f; }
    else if (in_range(nlo, nhi, p.x)) { r.x = (0.5f - 0.5f * p.y) / 4.0f; r.y = (0.5f - 0.5f * p.z) / 3.0f + 1.0f / 3.0f; }
    else if (in_range(plo, phi, p.y)) { r.x = (0.5f - 0.5f * p.x) / 4.0f + 3.0f / 4.0f; r.y = (0.5f - 0.5f * p.z) / 3.0f + 1.0f / 3.0f; }
    else if (in_range(nlo, nhi, p.y)) { r.x = (0.5f + 0.5f * p.x) / 4.0f + 1.0f / 4.0f; r.y = (0.5f - 0.5f * p.z) / 3.0f + 1.0f / 3.0f; }
    else if (in_range(plo, phi, p.z)) { r.x = (0.5f + 0.5f * p.x) / 4.0f + 1.0f / 4.0f; r.y = (0.5f - 0.5f * p.y) / 3.0f; }
    else if (in_range(nlo, nhi, p.z)) { r.x = (0.5f + 0.5f * p.x) / 4.0f + 1.0f / 4.0f; r.y = (0.5f + 0.5f * p.y) / 3.0f + 2.0f / 3.0f; }
    else { r.x = 0.0f; r.y = 0.0f; }
    return r;
}

/* UV for Sphere, src/rt.rs:518-526 */
static v2 sphere_uv(v3 hit, v3 pos)
{
    v3 v = v3_norm(v3_sub(hit, pos));
    v2 r;
    r.x = 0.5f + 0.5f * om_atan2f(v.x, -v.y) / OM_PI;
    r.y = 0.5f - 0.5f * v.z;
    return r;
}

static inline float f_fract(float x) { return x - truncf(x); }

/* UV for Plane, src/rt.rs:528-542 */
static v2 plane_uv(v3 hit)
{
    v2 r;
    r.x = f_fract(hit.x + 0.5f);
    if (r.x < 0.0f) r.x = 1.0f + r.x;
    r.y = f_fract(hit.y + 0.5f);
    if (r.y < 0.0f) r.y = 1.0f + r.y;
    return r;
}

/* Texture::get_color, src/rt.rs:618-628 (+ D4 clamp) */
static v3 tex_get_color(const tex_t *t, v2 uv)
{
    if (!t->dat) return V3(0.0f, 0.0f, 0.0f);
    uint64_t x = f_to_index(uv.x * (float)t->w);
    uint64_t y = f_to_index(uv.y * (float)t->h);
    uint64_t idx = x + y * (uint64_t)t->w;
    uint64_t last = (uint64_t)t->w * t->h - 1;
    if (idx > last) idx = last;
    return V3(t->dat[idx * 3], t->dat[idx * 3 + 1], t->dat[idx * 3 + 2]);
}

/* ------------------------------------------------------------------ BVH (src/rt.rs:630-703) */
/* Triangle::check_in_aabb, src/rt.rs:227-248 */
static int tri_in_aabb(const tri_t *t, v3 aabb, v3 rel_pos)
{
    v3 v0 = v3_add(rel_pos, v3_muls(aabb, 0.5f));
    v3 v1 = v3_sub(rel_pos, v3_muls(aabb, 0.5f));
    const v3 vs[3] = {t->a, t->b, t->c};
    for (int i = 0; i < 3; i++) {
        v3 v = vs[i];
        if (v.x > v0.x || v.y > v0.y || v.z > v0.z) continue;
        if (v.x < v1.x || v.y < v1.y || v.z < v1.z) continue;
        return 1;
    }
    return 0;
}

static const float GEN_POS[8][3] = { /* src/rt.rs:678-689 */
    {1, 1, 1}, {-1, 1, 1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, -1}, {1, -1, -1}};

/* BVH::construct, src/rt.rs:631-674 */
static bvh_t *bvh_construct(v3 aabb, v3 rel_pos, const tri_t *objs, uint32_t n, uint32_t d, uint32_t deep)
{
    bvh_t *c = (bvh_t *)calloc(1, sizeof(bvh_t));
    c->aabb = aabb; c->rel_pos = rel_pos;
    if (d >= deep) {
        uint32_t cnt = 0;
        uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
        for (uint32_t i = 0; i < n; i++) if (tri_in_aabb(&objs[i], c->aabb, c->rel_pos)) tmp[cnt++] = i;
        if (cnt) { c->content = tmp; c->n_content = cnt; } else free(tmp);
        return c;
    }
    bvh_t **tmp = (bvh_t **)malloc(sizeof(bvh_t *) * 8);
    uint32_t cnt = 0;
    for (int i = 0; i < 8; i++) {
        v3 v = V3(GEN_POS[i][0], GEN_POS[i][1], GEN_POS[i][2]);
        bvh_t *ch = bvh_construct(v3_muls(aabb, 0.5f), v3_add(rel_pos, v3_hadam(aabb, v3_muls(v, 0.25f))), objs, n, d + 1, deep);
        if (ch->content || ch->childs) tmp[cnt++] = ch;
        else free(ch);
    }
    if (cnt) { c->childs = tmp; c->n_childs = cnt; } else free(tmp);
    return c;
}

static void bvh_free(bvh_t *b)
{
    if (!b) return;
    for (uint32_t i = 0; i < b->n_childs; i++) bvh_free(b->childs[i]);
    free(b->childs); free(b->content); free(b);
}

static uint32_t bvh_total_ids(const bvh_t *b)
{
    uint32_t s = b->n_content;
    for (uint32_t i = 0; i < b->n_childs; i++) s += bvh_total_ids(b->childs[i]);
    return s;
}

/* Mesh::gen_aabb, src/rt.rs:261-270 */
static int mesh_gen_aabb(const tri_t *m, uint32_t n, v3 *out)
{
    if (n == 0) return 0;
    /* max_by(total_cmp) over |v|: for sign-cleared floats the total order is the order of the bit
     * patterns taken as unsigned integers (+NaN above +inf). */
    uint32_t mx = 0, my = 0, mz = 0;
    for (uint32_t i = 0; i < n; i++) {
        const v3 vs[3] = {m[i].a, m[i].b, m[i].c};
        for (int k = 0; k < 3; k++) {
            uint32_t ax = om_f2u(fabsf(vs[k].x)), ay = om_f2u(fabsf(vs[k].y)), az = om_f2u(fabsf(vs[k].z));
            if (ax > mx) mx = ax;
            if (ay > my) my = ay;
            if (az > mz) mz = az;
        }
    }
    *out = V3(2.0f * om_u2f(mx), 2.0f * om_u2f(my), 2.0f * om_u2f(mz));
    return 1;
}

/* ------------------------------------------------------------------ Renderer (src/rt.rs:706-864) */
typedef struct { uint32_t *ids; uint32_t n; } idlist_t;

/* Renderer::intersect_bvh, src/rt.rs:707-723: returns 0 for None */
static int intersect_bvh(const inst_t *inst, const ray_t *ray, const bvh_t *b, idlist_t *out)
{
    float t0, t1;
    if (!box_intersect(b->aabb, ray, v3_add(inst->pos, b->rel_pos), &t0, &t1)) return 0;
    if (b->content) {
        memcpy(out->ids + out->n, b->content, sizeof(uint32_t) * b->n_content);
        out->n += b->n_content;
        return 1;
    }
    /* childs.unwrap(): orc_create rejects meshes whose root has neither content nor childs */
    for (uint32_t i = 0; i < b->n_childs; i++) intersect_bvh(inst, ray, b->childs[i], out);
    return 1;
}

typedef struct { m3 rot_y, look; } xf_t;
static inline xf_t inst_xf(const inst_t *inst)   /* src/rt.rs:726-727, 779-780, 796-797 */
{
    xf_t x;
    x.rot_y = m3_rotate_y(v4_neg(inst->dir));
    x.look = m3_lookat(v4_neg(inst->dir), V3(0.0f, 0.0f, 1.0f));
    return x;
}
static inline v3 xf_apply(const xf_t *x, v3 v) { v3 t = m3_mul(&x->look, v); return m3_mul(&x->rot_y, t); }

/* Renderer::intersect, src/rt.rs:725-774 */
static int renderer_intersect(const rend_t *o, const inst_t *inst, const ray_t *ray, idlist_t *scratch,
                              float *t0, int32_t *i0, float *t1, int32_t *i1)
{
    xf_t xf = inst_xf(inst);
    ray_t n_ray = *ray;
    n_ray.orig = v3_add(inst->pos, xf_apply(&xf, v3_sub(ray->orig, inst->pos)));
    n_ray.dir = xf_apply(&xf, ray->dir);
    *i0 = -1; *i1 = -1;
    switch (o->kind) {
    case MRT_KIND_SPHERE: return sphere_intersect(o->r, &n_ray, inst->pos, t0, t1);
    case MRT_KIND_PLANE: { float t; if (!plane_intersect(o->n, &n_ray, inst->pos, &t)) return 0; *t0 = t; *t1 = t; return 1; }
    case MRT_KIND_BOX: return box_intersect(o->sizes, &n_ray, inst->pos, t0, t1);
    case MRT_KIND_TRIANGLE: { float t; if (!tri_intersect(&o->tri, &n_ray, inst->pos, &t)) return 0; *t0 = t; *t1 = t; return 1; }
    case MRT_KIND_MESH: {
        scratch->n = 0;
        if (o->bvh) {
            if (!intersect_bvh(inst, &n_ray, o->bvh, scratch)) return 0;
        } else {
            for (uint32_t i = 0; i < o->n_mesh; i++) scratch->ids[scratch->n++] = i;
        }
        /* Vec::dedup: consecutive duplicates only (src/rt.rs:756) */
        uint32_t m = 0;
        for (uint32_t i = 0; i < scratch->n; i++)
            if (m == 0 || scratch->ids[m - 1] != scratch->ids[i]) scratch->ids[m++] = scratch->ids[i];
        int any = 0; float bt0 = 0, bt1 = 0; int32_t bi0 = -1, bi1 = -1; int32_t k0 = 0, k1 = 0;
        for (uint32_t i = 0; i < m; i++) {
            float t;
            if (!tri_intersect(&o->mesh[scratch->ids[i]], &n_ray, inst->pos, &t)) continue;
            int32_t k = total_key(t);
            if (!any) { any = 1; bt0 = bt1 = t; bi0 = bi1 = (int32_t)scratch->ids[i]; k0 = k1 = k; continue; }
            if (k < k0) { k0 = k; bt0 = t; bi0 = (int32_t)scratch->ids[i]; }     /* min_by: first minimum */
            if (k >= k1) { k1 = k; bt1 = t; bi1 = (int32_t)scratch->ids[i]; }    /* max_by: last maximum */
        }
        if (!any) return 0;
        *t0 = bt0; *i0 = bi0; *t1 = bt1; *i1 = bi1;
        return 1;
    }
    }
    return 0;
}

/* Renderer::normal, src/rt.rs:776-793 */
static v3 renderer_normal(const rend_t *o, const inst_t *inst, const hit_t *hit)
{
    v3 hit_p = ray_point(&hit->ray);
    xf_t xf = inst_xf(inst);
    v3 n_hit = v3_add(inst->pos, xf_apply(&xf, v3_sub(hit_p, inst->pos)));
    v3 n;
    switch (o->kind) {
    case MRT_KIND_SPHERE: n = v3_sub(n_hit, inst->pos); break;                   /* src/rt.rs:447-451 */
    case MRT_KIND_PLANE: n = o->n; break;                                        /* src/rt.rs:453-457 */
    case MRT_KIND_BOX: n = box_normal(o->sizes, n_hit, inst->pos); break;
    case MRT_KIND_TRIANGLE: n = v3_cross(v3_sub(o->tri.b, o->tri.a), v3_sub(o->tri.c, o->tri.a)); break; /* :459-466 */
    default: { const tri_t *t = &o->mesh[hit->idx]; n = v3_cross(v3_sub(t->b, t->a), v3_sub(t->c, t->a)); break; }
    }
    return v3_norm(xf_apply(&xf, n));
}

/* Renderer::to_uv, src/rt.rs:795-809 (triangle / mesh: todo!() => rejected at create, D7) */
static v2 renderer_to_uv(const rend_t *o, const inst_t *inst, v3 hit)
{
    xf_t xf = inst_xf(inst);
    v3 n_hit = v3_add(inst->pos, xf_apply(&xf, v3_sub(hit, inst->pos)));
    switch (o->kind) {
    case MRT_KIND_SPHERE: return sphere_uv(n_hit, inst->pos);
    case MRT_KIND_PLANE: return plane_uv(n_hit);
    case MRT_KIND_BOX: return box_uv(o->sizes, n_hit, inst->pos);
    default: { v2 z = {0.0f, 0.0f}; return z; }
    }
}

/* Renderer::get_color .. get_emit, src/rt.rs:811-863, through RayHit::get_*, src/rt.rs:592-616 */
static v3 hit_get_color(const orc_ctx *c, const hit_t *h)
{
    const mat_t *m = &h->obj->mat;
    if (m->map[0] >= 0) return v3_hadam(m->albedo, tex_get_color(&c->tex[m->map[0]], renderer_to_uv(h->obj, h->inst, ray_point(&h->ray))));
    return m->albedo;
}
static float hit_get_scalar(const orc_ctx *c, const hit_t *h, int slot, float constant)
{
    const mat_t *m = &h->obj->mat;
    if (m->map[slot] >= 0) return tex_get_color(&c->tex[m->map[slot]], renderer_to_uv(h->obj, h->inst, ray_point(&h->ray))).x;
    return constant;
}
#define hit_get_rough(c, h)   hit_get_scalar(c, h, 1, (h)->obj->mat.rough)
#define hit_get_metal(c, h)   hit_get_scalar(c, h, 2, (h)->obj->mat.metal)
#define hit_get_glass(c, h)   hit_get_scalar(c, h, 3, (h)->obj->mat.glass)
#define hit_get_opacity(c, h) hit_get_scalar(c, h, 4, (h)->obj->mat.opacity)
#define hit_get_emit(c, h)    hit_get_scalar(c, h, 5, (h)->obj->mat.emit)

/* ------------------------------------------------------------------ RayTracer (src/rt.rs:866-1066) */
/* RayTracer::closest_hit, src/rt.rs:867-898.  want_hits = 0 => only Some/None (shadow query, :1036). */
static int closest_hit(const orc_ctx *c, const ray_t *ray, idlist_t *scratch, int want_hits, hit_t *h0, hit_t *h1)
{
    int any = 0; int32_t best_key = 0;
    const rend_t *bo = NULL; const inst_t *bi = NULL; float bt0 = 0, bt1 = 0; int32_t bi0 = -1, bi1 = -1;
    for (uint32_t r = 0; r < c->n_rend; r++) {
        const rend_t *o = &c->rend[r];
        for (uint32_t i = 0; i < o->n_inst; i++) {
            float t0, t1; int32_t i0, i1;
            if (!renderer_intersect(o, &o->inst[i], ray, scratch, &t0, &i0, &t1, &i1)) continue;
            if (!want_hits) return 1;
            int32_t k = total_key(t0);
            if (!any || k < best_key) {   /* min_by: first minimum wins */
                any = 1; best_key = k; bo = o; bi = &o->inst[i]; bt0 = t0; bt1 = t1; bi0 = i0; bi1 = i1;
            }
        }
    }
    if (!any) return 0;
    h0->obj = bo; h0->inst = bi; h0->idx = bi0; h0->ray = *ray; h0->ray.t = bt0;
    h0->norm = renderer_normal(bo, bi, h0);
    h1->obj = bo; h1->inst = bi; h1->idx = bi1; h1->ray = *ray; h1->ray.t = bt1;
    h1->norm = renderer_normal(bo, bi, h1);
    return 1;
}

typedef struct {
    v3 dir0;       /* un-jittered camera-space direction (src/rt.rs:904-908) */
    v3 focus;      /* focus point p (src/rt.rs:911-914) */
} pixel_cam_t;

/* RayTracer::iter + first half of RayTracer::cast, src/rt.rs:937-947, 900-914 */
static pixel_cam_t pixel_cam(const orc_ctx *c, float cx, float cy)
{
    float w = (float)c->res_w * c->ssaa;
    float h = (float)c->res_h * c->ssaa;
    float aspect = w / h;
    float uvx = aspect * (cx - 0.5f * w) / w;
    float uvy = (cy - 0.5f * h) / h;
    float tan_fov = tanf((0.5f * c->cam.fov) * (OM_PI / 180.0f));   /* f32::to_radians then tan (libm, host side) */
    pixel_cam_t pc;
    pc.dir0 = v3_norm(V3(uvx, 1.0f / (2.0f * tan_fov), -uvy));
    ray_t r = ray_cast_default(V3(c->cam.pos[0], c->cam.pos[1], c->cam.pos[2]), pc.dir0);
    r.t = c->cam.foc;
    pc.focus = ray_point(&r);
    return pc;
}

/* second half of RayTracer::cast, src/rt.rs:916-931 */
static ray_t camera_ray(const orc_ctx *c, const pixel_cam_t *pc, uint32_t pk)
{
    float u1 = orc_draw_f32(pk, DIM_LENS_X);
    float u2 = orc_draw_f32(pk, DIM_LENS_Z);
    v3 pos = V3(c->cam.pos[0] + (u1 - 0.5f) * c->cam.aprt, c->cam.pos[1], c->cam.pos[2] + (u2 - 0.5f) * c->cam.aprt);
    v3 new_dir = v3_norm(v3_sub(pc->focus, pos));
    v4 cd = {c->cam.dir[0], c->cam.dir[1], c->cam.dir[2], c->cam.dir[3]};
    m3 look = m3_lookat(cd, V3(0.0f, 0.0f, 1.0f));
    m3 rot_y = m3_rotate_y(cd);
    v3 t = m3_mul(&look, new_dir);
    return ray_cast_default(pos, m3_mul(&rot_y, t));
}

/* RayTracer::rand, src/rt.rs:996-1007.  Math contract v3: the reference takes th = acos(1 - 2 u1) and then
 * sin th, cos th through libm; here cos th = 1 - 2 u1 (exact: u1 lies on the 2^-23 lattice) and sin th = sqrt(4 u1 (1 - u1))
 * = sqrt((1 - cos th)(1 + cos th)) with both factors exact, one rounding in the product and a correctly rounded root: the same
 * two numbers to within an ulp (the libm composition is off by up to ~1e-7 near the poles, where acos loses the angle). */
static inline void rt_rand_polar(float u1, float *sth, float *cth)      /* contract v3 (D8) */
{
    *cth = 1.0f - 2.0f * u1;
    *sth = sqrtf((4.0f * u1) * (1.0f - u1));
}
/* the literal reading of src/rt.rs:997, 1000-1002: th = acos(1 - 2 u1), then sin th, cos th (contract functions for libm);
 * kept only so that the test can bound D8 against it */
static inline void rt_rand_polar_literal(float u1, float *sth, float *cth)
{
    float th = om_acosf(1.0f - 2.0f * u1);
    om_sincosf(th, sth, cth);
}
static v3 rt_rand(v3 n, float r, float u1, float u2)
{
    float phi = u2 * 2.0f * OM_PI;
    float cth, sth;
    rt_rand_polar(u1, &sth, &cth);
    float sphi, cphi;
    om_sincosf(phi, &sphi, &cphi);
    v3 v = V3(sth * cphi, sth * sphi, cth);
    return v3_norm(v3_add(n, v3_muls(v, r)));
}

/* Ray::reflect, src/rt.rs:559-572 */
static ray_t ray_reflect(const orc_ctx *c, const ray_t *self, const hit_t *hit, uint32_t pk)
{
    float rough = hit_get_rough(c, hit);
    float opacity = hit_get_opacity(c, hit);
    uint32_t b = self->bounce;
    if (hit->obj->mat.metal == 0.0f && opacity != 0.0f && (orc_draw_u32(pk, dim_of(b, SL_REFL_COIN)) < THR_080)) rough = 1.0f;
    v3 norm = rt_rand(hit->norm, rough, orc_draw_f32(pk, dim_of(b, SL_REFL_U1)), orc_draw_f32(pk, dim_of(b, SL_REFL_U2)));
    v3 dir = v3_norm(v3_reflect(self->dir, norm));
    return ray_cast(ray_point(self), dir, self->pwr * (1.0f - f_min(c->loss, 1.0f)), self->bounce + 1);
}

/* Ray::refract, src/rt.rs:574-589 */
static int ray_refract(const orc_ctx *c, const ray_t *self, const hit_t *hit, uint32_t pk, ray_t *out)
{
    float rough = hit_get_rough(c, hit);
    float opacity = hit_get_opacity(c, hit);
    uint32_t b = self->bounce;
    if (hit->obj->mat.metal == 0.0f && opacity != 0.0f && (orc_draw_u32(pk, dim_of(b, SL_REFR_COIN)) < THR_080)) rough = 1.0f;
    v3 norm = rt_rand(hit->norm, rough, orc_draw_f32(pk, dim_of(b, SL_REFR_U1)), orc_draw_f32(pk, dim_of(b, SL_REFR_U2)));
    float eta = 1.0f + 0.5f * hit_get_glass(c, hit);
    v3 d;
    if (!v3_refract(self->dir, eta, norm, &d)) return 0;
    d = v3_norm(d);
    *out = ray_cast(ray_point(self), d, self->pwr * (1.0f - f_min(c->loss, 1.0f)), self->bounce + 1);
    return 1;
}

typedef struct {
    idlist_t ids;
    hit_t *path;          /* bounce + 1 */
    uint8_t *vis;         /* (bounce + 1) * n_light : light visible from hit0 */
    uint64_t segments;
} scratch_t;

static void scratch_init(const orc_ctx *c, scratch_t *s)
{
    s->ids.ids = (uint32_t *)malloc(sizeof(uint32_t) * (c->max_ids ? c->max_ids : 1));
    s->ids.n = 0;
    s->path = (hit_t *)malloc(sizeof(hit_t) * ((size_t)c->bounce + 1));
    s->vis = (uint8_t *)malloc(((size_t)c->bounce + 1) * (c->n_light ? c->n_light : 1));
    s->segments = 0;
}
static void scratch_free(scratch_t *s) { free(s->ids.ids); free(s->path); free(s->vis); }

static inline v3 light_vec(const light_t *l, v3 hit_p)   /* src/rt.rs:1029-1032, 975-978 */
{
    if (l->kind == MRT_LIGHT_POINT) return v3_sub(l->v, hit_p);
    return v3_neg(v3_norm(l->v));
}

/* reduce_light(iter(x, y)) for one sample: src/rt.rs:937-994 with RaytraceIterator::next, :1014-1066 */
static v3 trace_sample(const orc_ctx *c, scratch_t *s, const pixel_cam_t *pc, uint32_t pixel, uint32_t sample)
{
    uint32_t pk = orc_path_key(c->seed, pixel, sample);
    ray_t next_ray = camera_ray(c, pc, pk);
    uint32_t n = 0;
    /* RaytraceIterator::next, collected (src/rt.rs:961) */
    while (next_ray.bounce <= c->bounce) {
        hit_t h0, h1;
        s->segments++;
        if (!closest_hit(c, &next_ray, &s->ids, 1, &h0, &h1)) break;
        uint8_t *vis = s->vis + (size_t)n * (c->n_light ? c->n_light : 1);
        v3 p0 = ray_point(&h0.ray);
        for (uint32_t li = 0; li < c->n_light; li++) {           /* src/rt.rs:1027-1046 */
            v3 l = light_vec(&c->light[li], p0);
            ray_t ray_l = ray_cast_default(p0, v3_norm(l));
            vis[li] = closest_hit(c, &ray_l, &s->ids, 0, NULL, NULL) ? 0 : 1;
        }
        uint32_t b = h0.ray.bounce;
        next_ray = ray_reflect(c, &h0.ray, &h0, pk);             /* src/rt.rs:1049 */
        hit_t n_hit = h0;
        float opacity = hit_get_opacity(c, &h0);
        if (bernoulli(f_min(1.0f - opacity, 0.85f), orc_draw_u32(pk, dim_of(b, SL_OPAC_COIN)))) {   /* src/rt.rs:1054 */
            ray_t r;
            if (ray_refract(c, &h1.ray, &h1, pk, &r)) { next_ray = r; n_hit = h1; }
        }
        s->path[n++] = n_hit;
    }
    if (n == 0) return c->sky_color;                              /* src/rt.rs:957-959 */

    v3 col = v3_muls(c->sky_color, c->sky_pwr);                   /* src/rt.rs:964 */
    for (uint32_t k = n; k-- > 0;) {
        const hit_t *hit = &s->path[k];
        const uint8_t *vis = s->vis + (size_t)k * (c->n_light ? c->n_light : 1);
        float emit = hit_get_emit(c, hit);
        if (bernoulli(emit, orc_draw_u32(pk, dim_of(hit->ray.bounce, SL_EMIT_COIN)))) {   /* src/rt.rs:968 */
            col = hit_get_color(c, hit);
            continue;
        }
        v3 l_col = V3(0.0f, 0.0f, 0.0f);                          /* src/rt.rs:973-987 */
        v3 hp = ray_point(&hit->ray);
        for (uint32_t li = 0; li < c->n_light; li++) {
            if (!vis[li]) continue;
            const light_t *light = &c->light[li];
            v3 l = light_vec(light, hp);
            v3 ln = v3_norm(l);
            float diff = f_max(v3_dot(ln, hit->norm), 0.0f);
            float sp = f_max(v3_dot(hit->ray.dir, v3_reflect(ln, hit->norm)), 0.0f);
            float s2 = sp * sp, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8, s32 = s16 * s16;   /* powi(32) */
            float spec = s32 * (1.0f - hit_get_rough(c, hit));
            v3 o_col = v3_muls(hit_get_color(c, hit), 1.0f - hit_get_metal(c, hit));
            v3 t = v3_hadam(v3_muls(o_col, diff), light->color);
            t = V3(t.x + spec, t.y + spec, t.z + spec);
            l_col = v3_add(l_col, v3_muls(t, light->pwr));
        }
        v3 hc = hit_get_color(c, hit);
        v3 d_col = v3_add(v3_muls(col, 0.5f), v3_hadam(hc, col)); /* src/rt.rs:990 */
        col = v3_muls(v3_add(d_col, l_col), hit->ray.pwr);        /* src/rt.rs:992 */
    }
    return col;
}

void orc_trace_pixel(const orc_ctx *c, uint32_t x, uint32_t y, uint32_t sidx, float rgb[3], uint32_t *segments)
{
    scratch_t s; scratch_init(c, &s);
    pixel_cam_t pc = pixel_cam(c, (float)x, (float)y);
    v3 col = trace_sample(c, &s, &pc, y * c->nw + x, sidx);
    rgb[0] = col.x; rgb[1] = col.y; rgb[2] = col.z;
    if (segments) *segments = (uint32_t)s.segments;
    scratch_free(&s);
}

/* ------------------------------------------------------------------ Sampler::execute (src/sampler.rs:28-78) */
typedef struct {
    orc_ctx *c;
    uint32_t n_samples, n_dim, g_w, g_h, row0, row1, threads;
    volatile uint32_t *job_counter;   /* one per sample pass */
    pthread_barrier_t *barrier;
    uint64_t segments;
} work_t;

static void *worker_main(void *arg)
{
    work_t *w = (work_t *)arg;
    orc_ctx *c = w->c;
    scratch_t s; scratch_init(c, &s);
    for (uint32_t pass = 0; pass < w->n_samples; pass++) {
        uint32_t sample = c->last_count + pass;
        for (;;) {
            uint32_t job = __atomic_fetch_add(&w->job_counter[pass], 1, __ATOMIC_RELAXED);
            if (job >= w->n_dim * w->n_dim) break;
            uint32_t g_x = job / w->n_dim, g_y = job % w->n_dim;      /* src/sampler.rs:40-41 */
            for (uint32_t lx = 0; lx < w->g_w; lx++) {                 /* src/sampler.rs:45-48 */
                uint32_t x = lx + w->g_w * g_x;
                if (x >= c->nw) break;                                  /* D6 */
                for (uint32_t ly = 0; ly < w->g_h; ly++) {
                    uint32_t y = ly + w->g_h * g_y;
                    if (y >= c->nh || y >= w->row1) break;
                    if (y < w->row0) continue;
                    pixel_cam_t pc = pixel_cam(c, (float)x, (float)y);
                    v3 col = trace_sample(c, &s, &pc, y * c->nw + x, sample);
                    float *dst = c->colors + ((size_t)y * c->nw + x) * 3;   /* src/sampler.rs:62-70 */
                    dst[0] += col.x; dst[1] += col.y; dst[2] += col.z;
                }
            }
        }
        pthread_barrier_wait(w->barrier);   /* pool.scoped join per pass, src/sampler.rs:39-74 */
    }
    w->segments = s.segments;
    scratch_free(&s);
    return NULL;
}

double orc_execute_rows(orc_ctx *c, uint32_t n_samples, uint32_t threads, uint32_t n_dim, uint32_t row0, uint32_t row1)
{
    if (threads == 0) threads = 1;
    if (n_dim == 0) n_dim = 64;
    if (row1 > c->nh) row1 = c->nh;
    struct timespec ta, tb;
    clock_gettime(CLOCK_MONOTONIC, &ta);
    uint32_t g_w = (uint32_t)ceilf((float)c->nw / (float)n_dim);      /* src/sampler.rs:32-33 */
    uint32_t g_h = (uint32_t)ceilf((float)c->nh / (float)n_dim);
    uint32_t *counters = (uint32_t *)calloc(n_samples ? n_samples : 1, sizeof(uint32_t));
    pthread_barrier_t barrier;
    pthread_barrier_init(&barrier, NULL, threads);
    work_t *w = (work_t *)calloc(threads, sizeof(work_t));
    pthread_t *th = (pthread_t *)calloc(threads, sizeof(pthread_t));
    for (uint32_t i = 0; i < threads; i++) {
        w[i].c = c; w[i].n_samples = n_samples; w[i].n_dim = n_dim; w[i].g_w = g_w; w[i].g_h = g_h;
        w[i].row0 = row0; w[i].row1 = row1; w[i].threads = threads; w[i].job_counter = counters; w[i].barrier = &barrier;
        pthread_create(&th[i], NULL, worker_main, &w[i]);
    }
    for (uint32_t i = 0; i < threads; i++) { pthread_join(th[i], NULL); c->segments += w[i].segments; }
    pthread_barrier_destroy(&barrier);
    free(w); free(th); free(counters);
    c->last_count += n_samples;                                        /* src/sampler.rs:76 */
    clock_gettime(CLOCK_MONOTONIC, &tb);
    return (double)(tb.tv_sec - ta.tv_sec) + 1e-9 * (double)(tb.tv_nsec - ta.tv_nsec);
}

double orc_execute(orc_ctx *c, uint32_t n_samples, uint32_t threads, uint32_t n_dim)
{
    return orc_execute_rows(c, n_samples, threads, n_dim, 0, c->nh);
}

/* ------------------------------------------------------------------ Sampler::img (src/sampler.rs:80-99) */
static inline uint8_t f_to_u8(float v)   /* `as u8`: saturating, NaN -> 0 */
{
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

void orc_tonemap_px(const float sum[3], uint32_t count, float gamma, float exp, uint8_t out[3])
{
    float rc = f_recip((float)count);                 /* Vec3f / f32 = * recip, src/lin.rs:296-301 */
    float w = (1.0f - exp) * (1.0f - exp);            /* powi(2) */
    for (int k = 0; k < 3; k++) {
        float col = sum[k] * rc;
        float g = om_powf(col, gamma);                                   /* src/sampler.rs:88 */
        float f = g * (1.0f + g / w) / (1.0f + g);                       /* src/sampler.rs:91 */
        out[k] = f_to_u8(255.0f * f);                                    /* src/sampler.rs:94 */
    }
}

/* image 0.24 imageops::sample: sinc / lanczos3_kernel */
static inline float img_sinc(float t) { float a = t * OM_PI; return (t == 0.0f) ? 1.0f : sinf(a) / a; }
static inline float img_lanczos3(float x) { return (fabsf(x) < 3.0f) ? img_sinc(x) * img_sinc(x / 3.0f) : 0.0f; }

/* weights of output index o when resampling src -> dst samples (image 0.24 horizontal_sample /
 * vertical_sample preamble).  Returns the tap count, or -1 if cap is too small. */
int orc_lanczos3_weights(uint32_t src, uint32_t dst, uint32_t o, uint32_t *left_out, float *w, uint32_t cap)
{
    float ratio = (float)src / (float)dst;
    float sratio = ratio < 1.0f ? 1.0f : ratio;
    float src_support = 3.0f * sratio;
    float input = ((float)o + 0.5f) * ratio;
    int64_t left = (int64_t)floorf(input - src_support);
    if (left < 0) left = 0;
    if (left > (int64_t)src - 1) left = (int64_t)src - 1;
    int64_t right = (int64_t)ceilf(input + src_support);
    if (right < left + 1) right = left + 1;
    if (right > (int64_t)src) right = (int64_t)src;
    input = input - 0.5f;
    uint32_t n = (uint32_t)(right - left);
    if (n > cap) return -1;
    float sum = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        float wi = img_lanczos3(((float)(left + i) - input) / sratio);
        w[i] = wi;
        sum += wi;
    }
    for (uint32_t i = 0; i < n; i++) w[i] /= sum;
    *left_out = (uint32_t)left;
    return (int)n;
}

static inline float f_round_half_away(float x) { return roundf(x); }

/* image::imageops::resize(.., Lanczos3) on an RGB8 image, src/sampler.rs:98 */
int orc_lanczos3_resize(const uint8_t *src, uint32_t sw, uint32_t sh, uint8_t *dst, uint32_t dw, uint32_t dh)
{
    if (sw == dw && sh == dh) { memcpy(dst, src, (size_t)sw * sh * 3); return 0; }
    uint32_t cap_v = (uint32_t)(2.0f * 3.0f * ((float)sh / (float)dh < 1.0f ? 1.0f : (float)sh / (float)dh)) + 4;
    uint32_t cap_h = (uint32_t)(2.0f * 3.0f * ((float)sw / (float)dw < 1.0f ? 1.0f : (float)sw / (float)dw)) + 4;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)sw * dh * 3);
    float *ws = (float *)malloc(sizeof(float) * (cap_v > cap_h ? cap_v : cap_h));
    if (!tmp || !ws) { free(tmp); free(ws); return -1; }
    for (uint32_t oy = 0; oy < dh; oy++) {                       /* vertical_sample */
        uint32_t left; int n = orc_lanczos3_weights(sh, dh, oy, &left, ws, cap_v);
        if (n < 0) { free(tmp); free(ws); return -1; }
        for (uint32_t x = 0; x < sw; x++) {
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
            for (int i = 0; i < n; i++) {
                const uint8_t *p = src + ((size_t)(left + i) * sw + x) * 3;
                t0 += (float)p[0] * ws[i]; t1 += (float)p[1] * ws[i]; t2 += (float)p[2] * ws[i];
            }
            float *q = tmp + ((size_t)oy * sw + x) * 3;
            q[0] = t0; q[1] = t1; q[2] = t2;
        }
    }
    for (uint32_t ox = 0; ox < dw; ox++) {                       /* horizontal_sample */
        uint32_t left; int n = orc_lanczos3_weights(sw, dw, ox, &left, ws, cap_h);
        if (n < 0) { free(tmp); free(ws); return -1; }
        for (uint32_t y = 0; y < dh; y++) {
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
            for (int i = 0; i < n; i++) {
                const float *p = tmp + ((size_t)y * sw + (left + i)) * 3;
                t0 += p[0] * ws[i]; t1 += p[1] * ws[i]; t2 += p[2] * ws[i];
            }
            uint8_t *q = dst + ((size_t)y * dw + ox) * 3;
            float c0 = t0 < 0.0f ? 0.0f : (t0 > 255.0f ? 255.0f : t0);   /* clamp(t, min, max) */
            float c1 = t1 < 0.0f ? 0.0f : (t1 > 255.0f ? 255.0f : t1);
            float c2 = t2 < 0.0f ? 0.0f : (t2 > 255.0f ? 255.0f : t2);
            q[0] = f_to_u8(f_round_half_away(c0)); q[1] = f_to_u8(f_round_half_away(c1)); q[2] = f_to_u8(f_round_half_away(c2));
        }
    }
    free(tmp); free(ws);
    return 0;
}

int orc_img_ss(const orc_ctx *c, uint8_t *rgb8)
{
    if (c->last_count == 0) return -1;    /* colors.get().unwrap() on an empty map panics */
    for (size_t i = 0; i < (size_t)c->nw * c->nh; i++) orc_tonemap_px(c->colors + i * 3, c->last_count, c->cam.gamma, c->cam.exp, rgb8 + i * 3);
    return 0;
}

int orc_img(const orc_ctx *c, uint8_t *rgb8)
{
    uint8_t *ss = (uint8_t *)malloc((size_t)c->nw * c->nh * 3);
    if (!ss) return -1;
    int rc = orc_img_ss(c, ss);
    if (rc == 0) rc = orc_lanczos3_resize(ss, c->nw, c->nh, rgb8, c->res_w, c->res_h);
    free(ss);
    return rc;
}

/* ------------------------------------------------------------------ create / destroy */
static int fail(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return 0; }

static int check_unit(float v) { return v >= 0.0f && v <= 1.0f; }

static int validate_material(const orc_ctx *c, const rend_t *o)
{
    const mat_t *m = &o->mat;
    for (int k = 0; k < 6; k++) {
        if (m->map[k] >= (int32_t)c->n_tex) return fail("material map index out of range");
        if (m->map[k] >= 0 && (o->kind == MRT_KIND_TRIANGLE || o->kind == MRT_KIND_MESH))
            return fail("textured triangle/mesh: reference hits todo!() (src/rt.rs:546,806)");
        if (m->map[k] >= 0 && c->tex[m->map[k]].dat && (c->tex[m->map[k]].w == 0 || c->tex[m->map[k]].h == 0))
            return fail("empty texture: reference would panic on index (src/rt.rs:624)");
    }
    if (m->map[5] < 0 && !check_unit(m->emit)) return fail("emit outside [0,1]: gen_bool panics (src/rt.rs:968)");
    if (m->map[4] < 0 && f_min(1.0f - m->opacity, 0.85f) < 0.0f) return fail("opacity > 1: gen_bool panics (src/rt.rs:1054)");
    if (m->map[5] >= 0 && c->tex[m->map[5]].dat) {
        const tex_t *t = &c->tex[m->map[5]];
        for (size_t i = 0; i < (size_t)t->w * t->h; i++) if (!check_unit(t->dat[i * 3])) return fail("emap texel outside [0,1]");
    }
    if (m->map[4] >= 0 && c->tex[m->map[4]].dat) {
        const tex_t *t = &c->tex[m->map[4]];
        for (size_t i = 0; i < (size_t)t->w * t->h; i++) if (f_min(1.0f - t->dat[i * 3], 0.85f) < 0.0f) return fail("omap texel > 1");
    }
    return 1;
}

orc_ctx *orc_create(const mrt_render_desc *d, uint64_t seed)
{
    g_err[0] = 0;
    if (!d) { fail("null desc"); return NULL; }
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    c->bounce = d->rt.bounce; c->loss = d->rt.loss;
    c->res_w = d->frame.res_w; c->res_h = d->frame.res_h; c->ssaa = d->frame.ssaa; c->cam = d->frame.cam;
    c->seed = seed;
    c->nw = (uint32_t)f_to_index((float)c->res_w * c->ssaa);     /* src/sampler.rs:29-30 */
    c->nh = (uint32_t)f_to_index((float)c->res_h * c->ssaa);
    c->sky_color = V3(d->scene.sky.color[0], d->scene.sky.color[1], d->scene.sky.color[2]);
    c->sky_pwr = d->scene.sky.pwr;
    c->n_tex = d->scene.n_textures;
    c->tex = (tex_t *)calloc(c->n_tex ? c->n_tex : 1, sizeof(tex_t));
    for (uint32_t i = 0; i < c->n_tex; i++) {
        const mrt_texture *t = &d->scene.textures[i];
        c->tex[i].w = t->w; c->tex[i].h = t->h;
        if (t->dat) {
            size_t n = (size_t)t->w * t->h * 3;
            c->tex[i].dat = (float *)malloc(sizeof(float) * (n ? n : 1));
            memcpy(c->tex[i].dat, t->dat, sizeof(float) * n);
        }
    }
    c->n_light = d->scene.n_light;
    c->light = (light_t *)calloc(c->n_light ? c->n_light : 1, sizeof(light_t));
    for (uint32_t i = 0; i < c->n_light; i++) {
        const mrt_light *l = &d->scene.light[i];
        c->light[i].kind = l->kind; c->light[i].v = V3(l->v[0], l->v[1], l->v[2]);
        c->light[i].pwr = l->pwr; c->light[i].color = V3(l->color[0], l->color[1], l->color[2]);
    }
    c->n_rend = d->scene.n_renderer;
    c->rend = (rend_t *)calloc(c->n_rend ? c->n_rend : 1, sizeof(rend_t));
    int ok = 1;
    for (uint32_t i = 0; i < c->n_rend && ok; i++) {
        const mrt_renderer *r = &d->scene.renderer[i];
        rend_t *o = &c->rend[i];
        o->kind = r->kind;
        o->r = r->param[0];
        o->n = V3(r->param[0], r->param[1], r->param[2]);
        o->sizes = o->n;
        o->tri.a = V3(r->param[0], r->param[1], r->param[2]);
        o->tri.b = V3(r->param[3], r->param[4], r->param[5]);
        o->tri.c = V3(r->param[6], r->param[7], r->param[8]);
        o->mat.albedo = V3(r->mat.albedo[0], r->mat.albedo[1], r->mat.albedo[2]);
        o->mat.rough = r->mat.rough; o->mat.metal = r->mat.metal; o->mat.glass = r->mat.glass;
        o->mat.opacity = r->mat.opacity; o->mat.emit = r->mat.emit;
        o->mat.map[0] = r->mat.tex; o->mat.map[1] = r->mat.rmap; o->mat.map[2] = r->mat.mmap;
        o->mat.map[3] = r->mat.gmap; o->mat.map[4] = r->mat.omap; o->mat.map[5] = r->mat.emap;
        o->n_inst = r->n_inst;
        o->inst = (inst_t *)calloc(o->n_inst ? o->n_inst : 1, sizeof(inst_t));
        for (uint32_t k = 0; k < o->n_inst; k++) {
            o->inst[k].pos = V3(r->inst[k].pos[0], r->inst[k].pos[1], r->inst[k].pos[2]);
            v4 dd = {r->inst[k].dir[0], r->inst[k].dir[1], r->inst[k].dir[2], r->inst[k].dir[3]};
            o->inst[k].dir = dd;
        }
        if (r->kind > MRT_KIND_MESH) { ok = fail("unknown renderer kind"); break; }
        if (r->kind == MRT_KIND_MESH) {
            o->n_mesh = r->n_tris;
            o->mesh = (tri_t *)calloc(o->n_mesh ? o->n_mesh : 1, sizeof(tri_t));
            for (uint32_t k = 0; k < o->n_mesh; k++) {
                const float *p = r->tris + (size_t)k * 9;
                o->mesh[k].a = V3(p[0], p[1], p[2]); o->mesh[k].b = V3(p[3], p[4], p[5]); o->mesh[k].c = V3(p[6], p[7], p[8]);
            }
            v3 aabb;
            if (mesh_gen_aabb(o->mesh, o->n_mesh, &aabb)) {           /* src/parser.rs:815-816 */
                o->bvh = bvh_construct(aabb, V3(0.0f, 0.0f, 0.0f), o->mesh, o->n_mesh, 0, 3);
                if (!o->bvh->content && !o->bvh->childs) { ok = fail("mesh octree is empty: reference unwrap() panics (src/rt.rs:717)"); break; }
                o->leaf_ids_total = bvh_total_ids(o->bvh);
                if (o->leaf_ids_total > c->max_ids) c->max_ids = o->leaf_ids_total;
            }
            if (o->n_mesh > c->max_ids) c->max_ids = o->n_mesh;
        }
        if (!validate_material(c, o)) ok = 0;
    }
    if (ok && (c->nw == 0 || c->nh == 0)) ok = fail("empty frame");
    if (!ok) { orc_destroy(c); return NULL; }
    c->colors = (float *)calloc((size_t)c->nw * c->nh * 3, sizeof(float));
    return c;
}

void orc_destroy(orc_ctx *c)
{
    if (!c) return;
    for (uint32_t i = 0; i < c->n_rend; i++) { free(c->rend[i].inst); free(c->rend[i].mesh); bvh_free(c->rend[i].bvh); }
    for (uint32_t i = 0; i < c->n_tex; i++) free(c->tex[i].dat);
    free(c->rend); free(c->tex); free(c->light); free(c->colors); free(c);
}

void orc_dims(const orc_ctx *c, uint32_t *nw, uint32_t *nh) { if (nw) *nw = c->nw; if (nh) *nh = c->nh; }
void orc_accum(const orc_ctx *c, float *rgb, uint32_t *count)
{
    if (rgb) memcpy(rgb, c->colors, sizeof(float) * (size_t)c->nw * c->nh * 3);
    if (count) *count = c->last_count;
}
void orc_set_accum(orc_ctx *c, const float *rgb, uint32_t count)
{
    memcpy(c->colors, rgb, sizeof(float) * (size_t)c->nw * c->nh * 3);
    c->last_count = count;
}
void orc_reset(orc_ctx *c) { memset(c->colors, 0, sizeof(float) * (size_t)c->nw * c->nh * 3); c->last_count = 0; c->segments = 0; }
uint64_t orc_segments(const orc_ctx *c) { return c->segments; }

/* leaves in traversal order */
static void octree_walk(const bvh_t *b, float *boxes, uint32_t *counts, uint32_t *ids, uint32_t cap, uint32_t *n_leaf, uint32_t *n_ids)
{
    if (b->content) {
        if (boxes) { float *q = boxes + (size_t)(*n_leaf) * 6; q[0] = b->rel_pos.x; q[1] = b->rel_pos.y; q[2] = b->rel_pos.z; q[3] = b->aabb.x; q[4] = b->aabb.y; q[5] = b->aabb.z; }
        if (counts) counts[*n_leaf] = b->n_content;
        for (uint32_t i = 0; i < b->n_content; i++) { if (ids && *n_ids < cap) ids[*n_ids] = b->content[i]; (*n_ids)++; }
        (*n_leaf)++;
        return;
    }
    for (uint32_t i = 0; i < b->n_childs; i++) octree_walk(b->childs[i], boxes, counts, ids, cap, n_leaf, n_ids);
}

int orc_mesh_octree(const orc_ctx *c, uint32_t renderer, float *leaf_boxes, uint32_t *leaf_counts, uint32_t *ids, uint32_t ids_cap, uint32_t *n_ids)
{
    if (renderer >= c->n_rend || c->rend[renderer].kind != MRT_KIND_MESH || !c->rend[renderer].bvh) return -1;
    uint32_t nl = 0, ni = 0;
    octree_walk(c->rend[renderer].bvh, leaf_boxes, leaf_counts, ids, ids_cap, &nl, &ni);
    if (n_ids) *n_ids = ni;
    return (int)nl;
}

void orc_math(int op, const float *a, const float *b, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        float x = a[i], y = b ? b[i] : 0.0f;
        switch (op) {
        case 0: out[i] = om_sinf(x); break;
        case 1: out[i] = om_cosf(x); break;
        case 2: out[i] = om_acosf(x); break;
        case 3: out[i] = om_atan2f(x, y); break;
        case 4: out[i] = om_powf(x, y); break;
        case 5: out[i] = 1.0f / x; break;
        case 6: out[i] = sqrtf(x); break;
        case 7: out[i] = x / y; break;
        /* the polar angle of RayTracer::rand for u1 = x: contract v3 (8 sin, 9 cos) and the literal composition (10, 11) */
        case 8: { float s, c; rt_rand_polar(x, &s, &c); out[i] = s; break; }
        case 9: { float s, c; rt_rand_polar(x, &s, &c); out[i] = c; break; }
        case 10: { float s, c; rt_rand_polar_literal(x, &s, &c); out[i] = s; break; }
        case 11: { float s, c; rt_rand_polar_literal(x, &s, &c); out[i] = c; break; }
        default: out[i] = 0.0f;
        }
    }
}
