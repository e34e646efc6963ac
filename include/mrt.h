/*
 * mrt.h — C ABI of libmrt_hip.so, the MI355X (gfx950) path-tracing backend that replaces the
 * thread-pool sampler of micro-raytracer.
 *
 * Every entry point below names the reference interface it stands in for (paths are relative
 * to the reference checkout, file:line).  The boundary is the reference's `Sampler`
 * (src/sampler.rs:11-100): `Sampler::new`, `Sampler::execute`, `Sampler::img`, and the two
 * callers `CLI::raytrace` (src/cli.rs:155-177) and `HttpServer::raytrace` (src/http.rs:136-148).
 *
 * Plain C: pointers + sizes only, no C++/torch types.  All floating point data is IEEE f32.
 * Descriptor pointers are borrowed for the duration of mrt_create only (the library deep-copies
 * into its own packed device layout); output buffers are caller-allocated.
 *
 * Threading: one mrt_ctx has a single owner at a time (like `&mut self` on Sampler); different
 * contexts may be used concurrently from different threads (HttpServer spawns one Sampler per
 * connection, src/http.rs:138,155).  mrt_last_error() is thread-local.
 */
#ifndef MRT_H
#define MRT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRT_ABI_VERSION 3u

/* ---- error codes (reference: Result<_, String> everywhere, src/sampler.rs:80, src/cli.rs:155) ---- */
#define MRT_OK            0
#define MRT_ERR_ARG      -1   /* null / malformed argument                                   */
#define MRT_ERR_SCENE    -2   /* scene rejected: something the reference would panic on       */
#define MRT_ERR_DEVICE   -3   /* HIP / RCCL runtime failure                                   */
#define MRT_ERR_LIMIT    -4   /* scene exceeds a device-side capacity                         */
#define MRT_ERR_STATE    -5   /* call not valid in this state (e.g. img before any sample)    */

/* ---- scene description: flattened rt::Render (src/rt.rs:10-190) ------------------------------ */

/* rt::Camera, src/rt.rs:64-72.  dir is Vec4f in (w,x,y,z) order as in src/lin.rs:18-25. */
typedef struct mrt_camera {
    float pos[3];
    float dir[4];
    float fov, gamma, exp, aprt, foc;
} mrt_camera;

/* rt::Frame, src/rt.rs:75-79. */
typedef struct mrt_frame {
    uint16_t res_w, res_h;
    float ssaa;
    mrt_camera cam;
} mrt_frame;

/* rt::RayTracer, src/rt.rs:17-22 (`sampler: Uniform<f32>` is replaced by mrt_opts.seed). */
typedef struct mrt_rt {
    uint32_t bounce;
    uint32_t sample;   /* informational: the caller owns the sample loop (src/cli.rs:162) */
    float loss;
} mrt_rt;

/* rt::Texture, src/rt.rs:82-86: w*h texels of f32 RGB; dat may be NULL (`dat: None` => black). */
typedef struct mrt_texture {
    uint32_t w, h;
    const float *dat;
} mrt_texture;

/* rt::Material, src/rt.rs:89-103.  Map slots index mrt_scene.textures, -1 = None.
 * Slot order: tex, rmap, mmap, gmap, omap, emap. */
typedef struct mrt_material {
    float albedo[3];
    float rough, metal, glass, opacity, emit;
    int32_t tex, rmap, mmap, gmap, omap, emap;
} mrt_material;

/* rt::RendererInstance, src/rt.rs:147-150. */
typedef struct mrt_instance {
    float pos[3];
    float dir[4];
} mrt_instance;

/* rt::RendererKind, src/rt.rs:138-144. */
#define MRT_KIND_SPHERE   0u   /* param[0] = r                                  */
#define MRT_KIND_PLANE    1u   /* param[0..3) = n                               */
#define MRT_KIND_BOX      2u   /* param[0..3) = sizes                           */
#define MRT_KIND_TRIANGLE 3u   /* param[0..9) = vtx0, vtx1, vtx2                */
#define MRT_KIND_MESH     4u   /* tris -> n_tris * 9 floats (vtx0, vtx1, vtx2)  */

/* rt::Renderer, src/rt.rs:153-158.  The instance list is the one the reference's loader
 * produces (src/parser.rs:838-864); its order is significant (first-minimum tie rule,
 * src/rt.rs:872).  For meshes the library rebuilds the reference's depth-3 octree
 * (src/rt.rs:630-703, called from src/parser.rs:815-816) from the raw triangles. */
typedef struct mrt_renderer {
    uint32_t kind;
    float param[9];
    const float *tris;
    uint32_t n_tris;
    mrt_material mat;
    const mrt_instance *inst;
    uint32_t n_inst;
} mrt_renderer;

/* rt::Light / LightKind, src/rt.rs:161-175. */
#define MRT_LIGHT_POINT 0u   /* v = pos */
#define MRT_LIGHT_DIR   1u   /* v = dir */
typedef struct mrt_light {
    uint32_t kind;
    float v[3];
    float pwr;
    float color[3];
} mrt_light;

/* rt::Sky, src/rt.rs:178-181. */
typedef struct mrt_sky {
    float color[3];
    float pwr;
} mrt_sky;

/* rt::Scene, src/rt.rs:184-190 (`renderer_bvh` is always None in the reference, src/parser.rs:922). */
typedef struct mrt_scene {
    const mrt_renderer *renderer;
    uint32_t n_renderer;
    const mrt_light *light;
    uint32_t n_light;
    mrt_sky sky;
    const mrt_texture *textures;
    uint32_t n_textures;
} mrt_scene;

/* rt::Render, src/rt.rs:10-14. */
typedef struct mrt_render_desc {
    mrt_rt rt;
    mrt_frame frame;
    mrt_scene scene;
} mrt_render_desc;

/* ---- options: what Sampler::new(workers, n_dim) (src/sampler.rs:19) turns into ---------------- */
typedef struct mrt_opts {
    uint32_t abi_version;   /* MRT_ABI_VERSION */
    uint64_t seed;          /* the reference is unseeded (rand::thread_rng, src/rt.rs:564..1054);
                               here the result is a pure function of (desc, seed, sample index)   */
    int32_t  device;        /* HIP device ordinal for this context; -1 = current device           */
    /* Row sharding (replaces the n_dim x n_dim tile jobs of src/sampler.rs:40-41): supersampled
     * rows are dealt in blocks of shard_rows rows, block b belongs to shard (b % shard_count).
     * shard_count = 0 or 1 => this context renders the whole frame. */
    uint32_t shard_index;
    uint32_t shard_count;
    uint32_t shard_rows;    /* 0 => default (8) */
    /* In-process multi-GPU (the single-process reference binary): n_devices > 1 makes one
     * sub-context per device 0..n_devices-1, row-sharded as above, gathered on device 0 by one
     * RCCL ncclGather per mrt_execute (librccl.so is loaded on demand).  Mutually exclusive with
     * shard_count > 1.  n_devices == 0 takes the count from the environment variable MRT_GPUS. */
    uint32_t n_devices;
    uint32_t flags;         /* MRT_FLAG_* */
    uint32_t reserved[4];
} mrt_opts;

#define MRT_FLAG_COUNT_SEGMENTS 1u   /* keep the per-launch path-segment counter (mrt_stats.segments): one wave reduction and
                                        one atomic per wavefront, a read-back when mrt_get_stats is called */
#define MRT_FLAG_NO_EVENT_TIMING 2u  /* no HIP events around the kernels (mrt_stats.kernel_ms / reduce_ms stay 0): what a caller that
                                        runs one sample per call (src/cli.rs:162-170) and never asks for stats wants */
#define MRT_FLAG_NO_LOOKAHEAD 8u     /* one-sample mrt_execute calls always run their own one-sample launch.  Default: a context that
                                        sees one-sample calls arrive back to back traces the NEXT samples ahead on a second stream
                                        (2, 4, ... 32 per launch, each into a plane of its own) and a call only folds its sample into
                                        the accumulator and waits for that: same samples, added in the same order -- bit for bit the
                                        plain loop -- at the rate of batched launches; MRT_LOOKAHEAD=0 in the environment switches it
                                        off too, MRT_LOOKAHEAD=n sets the samples per launch */
#define MRT_FLAG_DEFER 4u            /* mrt_execute only books its samples; they are traced, batched, when 1024 are booked or when
                                        the accumulator is observed or replaced (mrt_accum*, mrt_img*, mrt_set_accum*, mrt_get_stats,
                                        mrt_bind_accum).  Same samples, same image as eager execution (sums re-associated like any
                                        batched call); the Duration of a booking call is ~0.  Also set by the environment variable
                                        MRT_DEFER=1 (a non-zero number; "0" or empty is off) for unmodified callers.  Ignored while the accumulator's device memory is
                                        visible to the caller (mrt_bind_accum / mrt_accum_device_ptr). */

typedef struct mrt_ctx mrt_ctx;

/* Counters of the most recent mrt_execute (reference: the Duration returned by
 * Sampler::execute, src/sampler.rs:35,77, logged at src/cli.rs:164). */
typedef struct mrt_stats {
    double   kernel_ms;      /* HIP-event time of the path-tracing kernel of the last execute    */
    double   gather_ms;      /* time of the multi-GPU gather (0 on one device)                   */
    uint64_t samples;        /* path samples traced by the last execute on this context          */
    uint64_t segments;       /* path segments (closest-hit queries) of the last execute          */
    uint32_t launches;       /* kernel launches of the last execute                              */
    uint32_t lds_bytes;      /* LDS bytes per workgroup of the path-tracing kernel               */
    uint32_t block_threads;  /* workgroup size                                                   */
    uint32_t scene_bytes;    /* packed scene bytes staged per workgroup                          */
    uint32_t k_split;        /* lanes per pixel of the last execute (sample chunks dealt round-robin) */
    uint32_t deferred;       /* 1 when the last mrt_execute was booked under MRT_FLAG_DEFER / MRT_DEFER=1, 0 when it ran at once
                                (no deferral asked for, or the accumulator's device memory is visible to the caller) */
    double   img_ms;         /* HIP-event time of the kernels of the last mrt_img / mrt_img_ss (tone map + resize) */
    double   reduce_ms;      /* HIP-event time of reduce_chunks after the path-tracing kernel (0 when k_split == 1) */
    uint32_t kernel_features; /* which instantiation of the path-tracing kernel serves this context: its FEAT template argument
                                 (csrc/mrt_trace.h F_* bits), i.e. pt_megakernel<scene_in_lds, block_threads, kernel_features> */
    uint32_t scene_in_lds;   /* 1: every workgroup stages the packed scene in LDS; 0: it is read through L2 */
} mrt_stats;

/* Sampler::new + the first half of Sampler::execute's argument list (src/sampler.rs:19,28):
 * validates and flattens the scene, uploads it, allocates the accumulators.
 * Returns NULL on failure (see mrt_last_error / mrt_last_status). */
mrt_ctx *mrt_create(const mrt_render_desc *desc, const mrt_opts *opts);

/* Drop for Sampler. */
void mrt_destroy(mrt_ctx *ctx);

/* n_samples consecutive Sampler::execute calls (src/sampler.rs:28-78): every supersampled pixel
 * of this context's rows gets n_samples more path samples added to its accumulator and
 * last_count += n_samples.  Synchronous (like the scoped-pool join, src/sampler.rs:39-74).
 * *seconds (optional) receives the wall time, the Duration of src/sampler.rs:77. */
int mrt_execute(mrt_ctx *ctx, uint32_t n_samples, double *seconds);

/* Supersampled frame size: nw, nh of src/sampler.rs:29-30; local_rows = rows owned by this shard. */
int mrt_dims(const mrt_ctx *ctx, uint32_t *nw, uint32_t *nh, uint32_t *local_rows);

/* Sampler.colors / Sampler.last_count (src/sampler.rs:14-15): the full-frame sum of per-sample
 * radiance, rgb[nh][nw][3] f32 (rows not owned by this shard are left untouched), and the
 * number of samples accumulated so far. Either pointer may be NULL. */
int mrt_accum(mrt_ctx *ctx, float *rgb, uint32_t *count);

/* Shard-local view of the same data: rows[local_rows] = global row index of each local row;
 * rgb[local_rows][nw][3].  Either pointer may be NULL. */
int mrt_accum_local(mrt_ctx *ctx, float *rgb, uint32_t *rows);

/* Device pointer of the shard-local accumulator ([local_rows][nw][3] f32) for zero-copy
 * hand-off to a collective (one RCCL gather of these per mrt_execute batch). */
int mrt_accum_device_ptr(mrt_ctx *ctx, void **dev_ptr, size_t *bytes);

/* Rows the shard-local accumulator is allocated for: the largest local_rows over all shards of this
 * shard_count (so that equal-sized buffers can be gathered); rows past local_rows stay zero. */
int mrt_padded_rows(const mrt_ctx *ctx, uint32_t *rows);

/* Use caller-owned device memory (e.g. a torch tensor) of at least padded_rows * nw * 3 floats as the
 * shard-local accumulator from now on; current contents are copied over.  The buffer must outlive the
 * context or a later mrt_bind_accum(ctx, NULL, 0), which switches back to library-owned memory. */
int mrt_bind_accum(mrt_ctx *ctx, void *dev_ptr, size_t bytes);

/* mrt_set_accum from device memory on this context's device (e.g. the frame assembled from a gather). */
int mrt_set_accum_device(mrt_ctx *ctx, const void *dev_rgb, uint32_t count);

/* Replace the accumulator contents with a full frame (rgb[nh][nw][3]) and sample count,
 * e.g. on rank 0 after a gather, or to resume (the reference never persists `colors`). */
int mrt_set_accum(mrt_ctx *ctx, const float *rgb, uint32_t count);

/* Sampler::img (src/sampler.rs:80-99): mean, gamma, extended Reinhard, u8 truncation, then the
 * `image` crate's Lanczos3 resize nw x nh -> res_w x res_h.  rgb8[res_h][res_w][3].
 * Needs the whole frame in this context (shard_count <= 1, or after mrt_set_accum). */
int mrt_img(mrt_ctx *ctx, uint8_t *rgb8);

/* The tone-mapped supersampled image before the resize: rgb8[nh][nw][3] (src/sampler.rs:84-96). */
int mrt_img_ss(mrt_ctx *ctx, uint8_t *rgb8);

/* `img.save(&filename)` of the reference's CLI (src/cli.rs:168,174) for the lossless formats it is used with:
 * ".ppm" (P6) and ".png" (8-bit RGB, stored deflate).  Host-only helper; returns MRT_ERR_ARG for other extensions. */
int mrt_save_image(const char *path, const uint8_t *rgb8, uint32_t w, uint32_t h);

/* Zero the accumulators and last_count (a fresh Sampler on the same scene). */
int mrt_reset(mrt_ctx *ctx);

/* Counters of the last execute.  Not const: under deferred execution this is an observation (booked samples are traced
 * first), and the HIP-event times / the segment counter are read back here, lazily. */
int mrt_get_stats(mrt_ctx *ctx, mrt_stats *out);

/* Result<_, String>'s message for the calling thread's last failed call ("" if none). */
const char *mrt_last_error(void);
int mrt_last_status(void);

uint32_t mrt_abi_version(void);

/* Number of HIP devices visible (0 if none); never initialises more than the runtime. */
int mrt_device_count(void);

/* Test hook, host only (no device needed): what mrt_create would stage in LDS for this scene and the workgroup shape of its
 * launches -- the policy of csrc/mrt_api.cpp as data, so that it can be checked where no GPU exists. */
typedef struct mrt_plan {
    uint32_t staging;        /* 0 whole scene | 1 warm: texels in global memory (mesh kernels: + a per-lane leaf queue) | 2 deep: only the first
                                tbvh_hot_nodes nodes of the (level-ordered) triangle-BVH table staged, triangles in global
                                memory | 3 none: everything through L2 */
    uint32_t block_threads;  /* workgroup size of the batched launches */
    uint32_t lds_bytes;      /* LDS per workgroup: staged scene + lane stash + walk areas */
    uint32_t staged_bytes;   /* the staged part of the scene */
    uint32_t scene_bytes;    /* the whole packed scene without the octree leaf lists */
    uint32_t kernel_features;/* FEAT template argument of the kernel instantiation (mrt_stats.kernel_features) */
    uint32_t tbvh_nodes;     /* triangle-BVH nodes of all meshes */
    uint32_t tbvh_hot_nodes; /* of which staged (deep level; otherwise all when staged at all) */
    uint32_t small_plain_grid; /* 1: launches of less than one sample chunk take the plain grid instead of the persistent one */
    uint32_t walk_cap;       /* entries of a lane's walk area (node stack + leaf queue of the mesh walk; 0: no mesh walk) */
    uint32_t reserved[2];
} mrt_plan;
int mrt_plan_launch(const mrt_render_desc *desc, mrt_plan *out);

/* Test hook: run one device math-contract function elementwise on the GPU.
 * op: 0 sin, 1 cos, 2 acos, 3 atan2(a,b), 4 pow(a,b), 5 1/a, 6 sqrt(a), 7 a/b.  b may be NULL for unary ops. */
int mrt_selftest_math(int device, int op, const float *a, const float *b, float *out, size_t n);

/* Test hook: compare, on the device, the fast correctly rounded cores of the math contract (sqrt, 1/x, a/b and the
 * 1/sqrt(m) of Vec3f::norm, src/lin.rs:60-66) with the compiler's full IEEE expansions, on `count` inputs generated
 * from the indices first .. first+count-1: op 0 sqrt and op 1 recip take the index as the f32 bit pattern (first = 0,
 * count = 2^32 covers every float); op 2 divide and op 3 norm scale hash (seed, index) into operands.
 * *mismatches = number of differing results (NaN == NaN); example[4] = {a, b, fast, reference} of one of them. */
int mrt_selftest_sweep(int device, int op, uint64_t first, uint64_t count, uint32_t seed, uint64_t *mismatches, float *example);

#ifdef __cplusplus
}
#endif
#endif /* MRT_H */
