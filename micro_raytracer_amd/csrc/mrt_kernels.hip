// mrt_kernels.hip — gfx950 kernels of libmrt_hip.so.
//
//   pt_megakernel    Sampler::execute (reference src/sampler.rs:28-78) and everything of src/rt.rs
//                    it reaches: one lane per supersampled pixel, all samples of the launch in
//                    registers, scene staged in LDS, one accumulator read-modify-write per launch.
//   tonemap_u8       Sampler::img's per-pixel map (src/sampler.rs:84-96)
//   lanczos3_v/_h    image::imageops::resize(.., Lanczos3) (src/sampler.rs:98): vertical pass to f32,
//                    horizontal pass to u8, taps precomputed on the host.
//   math_selftest    elementwise device math contract, for the parity tests.
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (no fast-math; IEEE divide / sqrt).
#include <hip/hip_runtime.h>

#include "mrt_kernels.h"
#include "mrt_post.h"
#include "mrt_trace.h"

namespace mrt {

// Workgroup = tiles_x x tiles_y wavefronts, each wavefront an 8x8 pixel tile (64 lanes): neighbouring
// pixels share most of their path prefix, which keeps the per-lane predicates of the uniform traversal
// loop coherent.  Rows are the shard-local rows of this context (block-cyclic over shards).
// Register budget per instantiation (second __launch_bounds__ argument = minimum waves per SIMD).  The kernel is
// VALU-issue bound, so the light variants (planes / spheres / boxes, no maps, no lights, no triangles) are squeezed to
// 6 waves per SIMD (80 VGPRs, a few spills: measured +14 % on the Cornell box); the heavier variants lose more to
// spills than they gain from occupancy and keep the compiler's choice.  MRT_WAVES_PER_EU overrides (experiments).
// The per-path LDS stash (mrt_trace.h) is used by every launch shape that has room for it next to the scene: the
// 64- and 256-thread workgroups with the scene in LDS.  It moves 7-25 VGPRs of rarely touched state out of the loop.
constexpr bool lds_stash_for(bool scene_in_lds, int block_threads, u32 feat)
{
#ifdef MRT_NO_STASH
    return false;
#else
    (void)scene_in_lds;      // scenes read through L2 keep the stash too: the LDS is otherwise empty
    return block_threads != 512 && !(feat & F_NOSTASH);
#endif
}
#ifndef MRT_BVH_WAVES
#define MRT_BVH_WAVES 6
#endif
constexpr int waves_for(u32 feat_, int block_threads)
{
#ifdef MRT_WAVES_PER_EU
    return MRT_WAVES_PER_EU;
#else
    const u32 feat = plain_feat(feat_);      // (F_IDENT changes nothing here)
    // Planes and spheres only (the Cornell box), 256-thread workgroups: 8 waves per SIMD (64 VGPRs).  With single-wave
    // workgroups every wavefront brings its own 5.5 KB of LDS (scene copy + stash) and the CU tops out at 29 of them, so a
    // bound of 8 only bought spills there (-3 %); four waves around one copy need 13 KB and all 32 fit: 7.42 -> 7.93
    // Gsamples/s on the headline frame (7.71 with the 7-wave build of the same shape).  With boxes (CornellBox2) 8 loses to 7.
    if (feat == 0u && block_threads == 256) return 8;
    // The instance-BVH kernels without mesh code, warm staging (F_COLD: texels in global memory, so the LDS no longer caps the
    // resident wavefronts at one 1024-thread workgroup): bound to 6 waves per SIMD (80 VGPRs, a few spills).  These walks
    // wait on dependent LDS reads, not on issue slots: the Minecraft-shaped scene gains 12 % with 5 waves, 16 % with 6, 17 %
    // with 7-8 over the 4 its 114 VGPRs allow.  The mesh kernels stay at 4: their LDS footprint caps them at 16 waves per CU.
    if ((feat & F_COLD) && (feat & F_BVH) && !(feat & F_TRI)) return MRT_BVH_WAVES;
    // Scenes with lights but without meshes, triangles or an instance BVH (example/Default.json, dof.json: 106-122 VGPRs as the
    // compiler would have it, 4 waves): bound to 5 waves per SIMD (96 VGPRs).  1080p renders: default scene 110 -> 118
    // Gsamples/s, dof scene 17.7 -> 20.3 (6 waves: 111 / 19.2).  The mesh kernels lose with every register taken from them
    // (kitchen-sink scene: 3284 / 3095 / 2765 Msamples/s at 4 / 5 / 6 waves).
    if ((feat & F_LIGHTS) && !(feat & (F_TRI | F_BVH | F_COLD | F_NOSTASH)) && block_threads <= 256) return 5;      // (larger workgroups: LDS-capped at 16 waves per CU anyway)
    return (feat & ~F_BOX) == 0 ? 7 : ((feat & (F_LIGHTS | F_TRI | F_BVH)) == 0 ? 6 : 4);      // the BVH walks need their registers more than two extra waves
#endif
}

template <bool SCENE_IN_LDS, int BLOCK_THREADS, u32 FEAT>
__global__ void __launch_bounds__(BLOCK_THREADS, waves_for(FEAT, BLOCK_THREADS)) pt_megakernel(const Params P, const u32 *__restrict__ blob_g)
{
    extern __shared__ uint4 lds_blob[];
    const float *F;
    // words of the scene this workgroup stages: all of it, or (F_COLD) the hot prefix -- records, transforms, materials, node
    // arrays; triangles, membership tables and texels are then read from global memory
    const u32 staged_words = !SCENE_IN_LDS ? 0u : staged_words_for(P, FEAT);
    if (SCENE_IN_LDS) {
        const uint4 *g = reinterpret_cast<const uint4 *>(P.blob);
        const u32 n4 = staged_words >> 2;
        for (u32 i = threadIdx.x; i < n4; i += blockDim.x) lds_blob[i] = g[i];
        __syncthreads();
        F = reinterpret_cast<const float *>(lds_blob);
    } else {
        F = reinterpret_cast<const float *>(P.blob);
    }

    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    Scn S;
    S.F = F;
#ifdef MRT_UNIFORM_SMEM
    S.U = reinterpret_cast<const float *>(blob_g);
#else
    S.U = F;
#endif
    S.G = reinterpret_cast<const float *>(blob_g);
    S.P = &P;
    // behind the staged scene (16-byte aligned): [lane stash: ST_SLOTS x blockDim floats] [walk areas: P.walk_cap x blockDim words]
    const u32 stash_base4 = (staged_words + 3u) >> 2;
    constexpr bool kStash = lds_stash_for(SCENE_IN_LDS, BLOCK_THREADS, FEAT);
    S.wk = nullptr; S.wk_stride = BLOCK_THREADS;
    if constexpr (has_walk_area(FEAT))
        S.wk = (void *)(reinterpret_cast<float *>(lds_blob + stash_base4) + (kStash ? stash_slots_for(FEAT, BLOCK_THREADS) * BLOCK_THREADS : 0u) + threadIdx.x);
    u32 segments = 0;
#ifdef MRT_PHASE_TIMING
    unsigned long long wave_ticks[4] = {0ull, 0ull, 0ull, 0ull};
#endif
    // one 8x8 tile of shard-local rows for this wavefront, lane k of the sample split
    auto do_tile = [&](u32 tx, u32 ty, u32 k) {
        const u32 x = tx * 8u + (lane & 7u);
        const u32 ry = ty * 8u + (lane >> 3);
        // shard-local row -> frame row: row block b of this shard is frame row block b * shard_count + shard_index
        const u32 blk = ry / P.shard_rows;
        const u32 y = (blk * P.shard_count + P.shard_index) * P.shard_rows + (ry - blk * P.shard_rows);
        const bool active = x < P.nw && ry < P.local_rows && y < P.nh;
        if (!active) return;
        u32 seg = 0;
        LaneJob job;
        job.k = k;
        job.word = (ry * P.nw + x) * 3u;        // < 2^32: mrt_create limits a shard to 2^30 pixels
        if constexpr (lds_stash_for(SCENE_IN_LDS, BLOCK_THREADS, FEAT)) {
            // per-lane column behind the scene blob (16-byte aligned): ST_SLOTS x blockDim floats
#ifdef MRT_PHASE_TIMING
            unsigned long long tk[4] = {0ull, 0ull, 0ull, 0ull};
#else
            unsigned long long *tk = nullptr;
#endif
            LdsStash<BLOCK_THREADS> st;
            st.base = (lds_vfloat *)(reinterpret_cast<float *>(lds_blob + stash_base4) + threadIdx.x);
            render_pixel<FEAT>(S, st, x, y, job, seg, tk);
#ifdef MRT_PHASE_TIMING
            for (int k = 0; k < 4; ++k) wave_ticks[k] += tk[k];
#endif
        } else {
            RegStash st;
            render_pixel<FEAT>(S, st, x, y, job, seg);
        }
        segments += seg;
    };
    // Persistent workgroup (more than one wavefront, P.persist_grid set): its wavefronts draw tiles from a counter until the
    // launch is out of tiles, so a CU never waits for the slowest wavefront of a workgroup (whose LDS copy of the scene
    // would otherwise keep the next workgroup out).  Otherwise blockIdx addresses the one tile of each wavefront.
    const bool persist = BLOCK_THREADS > 64 && P.persist_grid != 0u;
    const u32 n_tx = (P.nw + 7u) >> 3, n_ty = (P.local_rows + 7u) >> 3;
    const u32 per_k = n_tx * n_ty, total = per_k * P.k_split;
    for (;;) {
        u32 tx = blockIdx.x * P.tiles_x + wave % P.tiles_x, ty = blockIdx.y * P.tiles_y + wave / P.tiles_x, k = blockIdx.z;
        if (persist) {
            u32 t = 0;
            if (lane == 0) t = atomicAdd(P.tile_counter, 1u);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= total) break;
            k = t / per_k;
            const u32 r = t - k * per_k;
            ty = r / n_tx;
            tx = r - ty * n_tx;
        }
        do_tile(tx, ty, k);
        if (!persist) break;
    }
#ifdef MRT_PHASE_TIMING
    // one lane per wavefront (the longest-running one speaks for the wave: every lane carries the wave's clock differences)
    for (int k = 0; k < 4; ++k) {
        unsigned long long v = wave_ticks[k];
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
        if (lane == 0 && v) atomicAdd(P.segments + 2 + k, v);
    }
#endif
    if (P.count_segments) {
        // wave-level sum (every lane of the wavefront is here), one atomic per wavefront
        u32 v = segments;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && v) atomicAdd(P.segments, (unsigned long long)v);
    }
}

// acc[p] += chunk sums in chunk order (the canonical order of mrt_trace.h), one thread per accumulator word
__global__ void __launch_bounds__(256) reduce_chunks(float *__restrict__ accum, const float *__restrict__ partial, size_t n_words,
                                                     size_t stride, u32 n_chunks)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    float a = accum[i];
    for (u32 j = 0; j < n_chunks; ++j) a += partial[(size_t)j * stride + i];
    accum[i] = a;
}

// rows of the gathered shard accumulators -> their place in the frame (multi-device contexts)
__global__ void __launch_bounds__(256) scatter_rows(float *__restrict__ frame, const float *__restrict__ gathered, const u32 *__restrict__ rowmap,
                                                    u32 n_rows, u32 row_words)
{
    const u32 r = blockIdx.y;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows || i >= row_words) return;
    const u32 y = rowmap[r];
    if (y == 0xffffffffu) return;
    frame[(size_t)y * row_words + i] = gathered[(size_t)r * row_words + i];
}

__global__ void __launch_bounds__(256) tonemap_u8(const float *__restrict__ accum, unsigned char *__restrict__ out,
                                                  u32 n_px, float rc, float gamma, float wexp)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_px) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[(size_t)i * 3 + k] = tonemap_channel(accum[(size_t)i * 3 + k], rc, gamma, wexp);
}

// vertical_sample of image 0.24: out[oy][x][c] = sum_i src[left+i][x][c] * w[i], f32, taps in order
__global__ void __launch_bounds__(256) lanczos3_v(const unsigned char *__restrict__ src, float *__restrict__ dst, u32 sw, u32 dh,
                                                  const u32 *__restrict__ left, const u32 *__restrict__ count,
                                                  const float *__restrict__ weight, u32 cap)
{
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;      // element = x * 3 + c
    const u32 oy = blockIdx.y;
    if (e >= sw * 3u || oy >= dh) return;
    const u32 l = left[oy], n = count[oy];
    const float *w = weight + (size_t)oy * cap;
    float t = 0.0f;
    for (u32 i = 0; i < n; ++i) t += (float)src[(size_t)(l + i) * sw * 3u + e] * w[i];
    dst[(size_t)oy * sw * 3u + e] = t;
}

// horizontal_sample: clamp to [0,255], round half away from zero, u8
__global__ void __launch_bounds__(256) lanczos3_h(const float *__restrict__ src, unsigned char *__restrict__ dst, u32 sw, u32 dw, u32 dh,
                                                  const u32 *__restrict__ left, const u32 *__restrict__ count,
                                                  const float *__restrict__ weight, u32 cap)
{
    const u32 ox = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 y = blockIdx.y;
    if (ox >= dw || y >= dh) return;
    const u32 l = left[ox], n = count[ox];
    const float *w = weight + (size_t)ox * cap;
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    for (u32 i = 0; i < n; ++i) {
        const float *p = src + ((size_t)y * sw + (l + i)) * 3u;
        t0 += p[0] * w[i]; t1 += p[1] * w[i]; t2 += p[2] * w[i];
    }
    unsigned char *q = dst + ((size_t)y * dw + ox) * 3u;
    q[0] = resample_to_u8(t0); q[1] = resample_to_u8(t1); q[2] = resample_to_u8(t2);
}

__global__ void math_selftest(int op, const float *a, const float *b, float *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b ? b[i] : 0.0f;
    float s, c, r = 0.0f;
    switch (op) {
    case 0: sincos_(x, s, c); r = s; break;
    case 1: sincos_(x, s, c); r = c; break;
    case 2: r = acos_(x); break;
    case 3: r = atan2_(x, y); break;
    case 4: r = pow_(x, y); break;
    case 5: r = recip_(x); break;
    case 6: r = sqrt_(x); break;
    case 7: r = div_(x, y); break;
    case 8: r = fmax_(x, y); break;
    case 9: r = fmin_(x, y); break;
    case 10: r = u2f((u32)total_key(x)); break;
    case 11: r = u32_to_unit(draw_u32(f2u(x), f2u(y))); break;
    case 12: r = norm(v3(x, y, 0.25f)).x; break;
    default: break;
    }
    out[i] = r;
}

// Exhaustive / randomised comparison of the wave-uniform fast cores of mrt_math.h (sqrt_, recip_, div_, recip_sqrt_)
// with the compiler's full IEEE expansions, on the device.  Inputs are generated from the element index:
//   op 0 sqrt, op 1 recip: x = the bit pattern `first + i` (a sweep of 2^32 indices covers every f32)
//   op 2 divide, op 3 norm scale (1 / sqrt(x*x + y*y + z*z) as norm() computes it): operands hashed from (seed, first + i);
//        three wavefronts in four draw exponents inside the fast window (so the cores run), the fourth draws raw
//        bit patterns (zeros, denormals, infinities, NaNs: the fallback runs).
// A NaN result matches any NaN (payloads are not part of the contract).  Counts mismatches; keeps one example.
__device__ inline float sweep_operand(u32 h, bool windowed)
{
    if (!windowed) return u2f(h);
    const u32 e = 127u - 40u + (h >> 9) % 80u;                   // exponent field inside [2^-40, 2^40)
    return u2f((h & 0x80000000u) | (e << 23) | (mix32(h) & 0x7fffffu));
}
__global__ void __launch_bounds__(256) math_sweep(int op, unsigned long long first, unsigned long long n, u32 seed,
                                                  unsigned long long *mismatches, float *example)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long idx = first + i;
    const bool windowed = ((idx >> 6) & 3ull) != 3ull;           // per wavefront, so that the wave-uniform fast path is taken
    const u32 k = mix32((u32)idx ^ seed) + (u32)(idx >> 32) * kGold;
    float a = 0.0f, b = 0.0f, c = 0.0f, fast = 0.0f, ref = 0.0f;
    switch (op) {
    case 0: a = u2f((u32)idx); fast = sqrt_(a); ref = __builtin_sqrtf(a); break;
    case 1: a = u2f((u32)idx); fast = recip_(a); ref = 1.0f / a; break;
    case 2: a = sweep_operand(mix32(k + 1u), windowed); b = sweep_operand(mix32(k + 2u), windowed); fast = div_(a, b); ref = a / b; break;
    case 3: {
        a = sweep_operand(mix32(k + 1u), windowed); b = sweep_operand(mix32(k + 2u), windowed); c = sweep_operand(mix32(k + 3u), windowed);
        const float m = a * a + b * b + c * c;
        fast = recip_sqrt_(m); ref = 1.0f / __builtin_sqrtf(m);
        break;
    }
    default: break;
    }
    const bool same = f2u(fast) == f2u(ref) || (fast != fast && ref != ref);
    if (!same) {
        if (atomicAdd(mismatches, 1ull) == 0ull) { example[0] = a; example[1] = b; example[2] = fast; example[3] = ref; }
    }
}

// ---- launchers (declared in mrt_kernels.h) ----
// One instantiation per feature set for the two common launch shapes with the scene in LDS: 256 threads (2x2 wave
// tiles) and 64 threads (one 8x8 tile per workgroup, used when the frame has too few tiles to balance 256 CUs with
// 4-wave workgroups).  The 512-thread shape (one LDS copy per CU, scenes of 78-160 KB) and the scene-in-L2 fallback
// and the 1024-thread shape (one LDS copy + stash per CU, 16 waves) carry every feature.
template <int THREADS, u32 FEAT>
static void launch_lds(dim3 grid, size_t lds, hipStream_t stream, const Params &P)
{
    hipLaunchKernelGGL((pt_megakernel<true, THREADS, FEAT>), grid, dim3(THREADS), lds, stream, P, P.blob);
}

size_t pt_lds_bytes(const Params &P, u32 block_threads, bool scene_in_lds, u32 features)
{
    const u32 inst = pt_instantiation(block_threads, scene_in_lds, features);      // what the kernel itself sees as FEAT
    size_t lds = scene_in_lds ? (size_t)staged_words_for(P, inst) * 4u : 0u;
    lds = (lds + 15u) & ~(size_t)15u;
    if (lds_stash_for(scene_in_lds, (int)block_threads, inst)) lds += (size_t)stash_slots_for(inst, block_threads) * block_threads * sizeof(float);
    if (has_walk_area(inst)) lds += (size_t)P.walk_cap * block_threads * sizeof(u32);
    return lds;
}

// The FEAT template argument of the pt_megakernel instantiation that serves a scene with feature set `features` in a
// launch shape: the 64- and 256-thread shapes with the scene in LDS exist for every plain feature set; the large
// shapes (512 / 1024 threads, scene through L2) with and without the triangle / mesh code; the instance-BVH kernels in
// four feature sets per shape -- plain primitives without / with lights (sphere and plane crowds), everything but
// triangles and meshes, everything -- of which the smallest covering one runs (a Minecraft-shaped scene without the
// triangle / mesh code: 113 VGPRs and no scratch instead of 128 + 72 B, +15 %).  Reported in mrt_stats.kernel_features.
u32 pt_instantiation(u32 block_threads, bool scene_in_lds, u32 features)
{
    constexpr u32 FN = F_ALL & ~F_TRI;
    const u32 need = features & F_ALL;
    const u32 big = (need & F_TRI) ? (u32)F_ALL : FN;
    u32 cold = (scene_in_lds && (features & F_COLD)) ? (u32)F_COLD : 0u;             // the cold kernels exist in the big feature sets only
    if (cold && (features & F_DEEP) && (need & F_TRI)) cold |= F_DEEP;               // ... and the deep ones with the mesh code only
    const u32 nostash = (scene_in_lds && block_threads == 1024u && (features & F_NOSTASH) && !cold) ? (u32)F_NOSTASH : 0u;
    if (features & F_BVH) {
        if (!scene_in_lds) return big | F_BVH;
        const u32 pick = (need & (F_BOX | F_TRI | F_MAPS)) == 0 ? (need & F_LIGHTS) : big;
        // sphere / plane crowds whose instances are all untransformed (the reference's Instance.json): F_IDENT builds
        const u32 ident = ((features & F_IDENT) && pick != big && !nostash && !cold && block_threads != 64u) ? (u32)F_IDENT : 0u;
        return pick | F_BVH | nostash | cold | ident;
    }
    if (!scene_in_lds) return big;
    if (cold) return big | cold;
    // scenes whose instances are all untransformed: the F_IDENT builds of the plain 256-thread kernels without triangle / map code
    if (block_threads == 256u && (features & F_IDENT) && (need & (F_TRI | F_MAPS)) == 0u) return need | F_IDENT;
    if (block_threads == 64u || block_threads == 256u) return need;
    return big | nostash;
}

#define MRT_CASE(T, F) case (F): launch_lds<T, (F)>(grid, lds, stream, P); return hipGetLastError();
#define MRT_CASE_L2(F) case (F): hipLaunchKernelGGL((pt_megakernel<false, 256, (F)>), grid, dim3(256), lds, stream, P, P.blob); return hipGetLastError();
#define MRT_PLAIN16(T) MRT_CASE(T, 0) MRT_CASE(T, 1) MRT_CASE(T, 2) MRT_CASE(T, 3) MRT_CASE(T, 4) MRT_CASE(T, 5) MRT_CASE(T, 6) MRT_CASE(T, 7) \
    MRT_CASE(T, 8) MRT_CASE(T, 9) MRT_CASE(T, 10) MRT_CASE(T, 11) MRT_CASE(T, 12) MRT_CASE(T, 13) MRT_CASE(T, 14) MRT_CASE(T, 15)
#define MRT_BVH4(T, X) MRT_CASE(T, F_BVH | (X)) MRT_CASE(T, F_LIGHTS | F_BVH | (X)) MRT_CASE(T, (F_ALL & ~F_TRI) | F_BVH | (X)) MRT_CASE(T, F_ALL | F_BVH | (X))
#define MRT_BIG2(T, X) MRT_CASE(T, (F_ALL & ~F_TRI) | (X)) MRT_CASE(T, F_ALL | (X))
#define MRT_IDENT4(T) MRT_CASE(T, F_IDENT) MRT_CASE(T, F_IDENT | F_BOX) MRT_CASE(T, F_IDENT | F_LIGHTS) MRT_CASE(T, F_IDENT | F_BOX | F_LIGHTS)
#define MRT_IDENT_BVH2(T) MRT_CASE(T, F_IDENT | F_BVH) MRT_CASE(T, F_IDENT | F_LIGHTS | F_BVH)
#define MRT_DEEP2(T) MRT_CASE(T, F_ALL | F_COLD | F_DEEP) MRT_CASE(T, F_ALL | F_BVH | F_COLD | F_DEEP)

hipError_t launch_pt(const Params &P, u32 block_threads, bool scene_in_lds, u32 features, hipStream_t stream)
{
    if (block_threads != P.tiles_x * P.tiles_y * 64u) return hipErrorInvalidConfiguration;
    const u32 tile_w = P.tiles_x * 8u, tile_h = P.tiles_y * 8u;
    dim3 grid((P.nw + tile_w - 1) / tile_w, (P.local_rows + tile_h - 1) / tile_h, P.k_split);
    if (block_threads > 64u && P.persist_grid) {
        const unsigned long long n_wg = (unsigned long long)grid.x * grid.y * grid.z;
        grid = dim3((unsigned)(n_wg < P.persist_grid ? n_wg : P.persist_grid), 1, 1);
    }
    const size_t lds = pt_lds_bytes(P, block_threads, scene_in_lds, features);
    const u32 inst = pt_instantiation(block_threads, scene_in_lds, features);
    if (!scene_in_lds) {
        if (block_threads != 256u) return hipErrorInvalidConfiguration;
        switch (inst) { MRT_CASE_L2(F_ALL & ~F_TRI) MRT_CASE_L2(F_ALL) MRT_CASE_L2((F_ALL & ~F_TRI) | F_BVH) MRT_CASE_L2(F_ALL | F_BVH) default: break; }
    } else if (block_threads == 64u) {
        switch (inst) { MRT_PLAIN16(64) MRT_BVH4(64, 0u) default: break; }
    } else if (block_threads == 256u) {
        switch (inst) { MRT_PLAIN16(256) MRT_IDENT4(256) MRT_BVH4(256, 0u) MRT_IDENT_BVH2(256) MRT_BIG2(256, F_COLD) MRT_BVH4(256, F_COLD) MRT_DEEP2(256) default: break; }
    } else if (block_threads == 512u) {
        switch (inst) { MRT_BIG2(512, 0u) MRT_BVH4(512, 0u) MRT_IDENT_BVH2(512) MRT_BIG2(512, F_COLD) MRT_BVH4(512, F_COLD) MRT_DEEP2(512) default: break; }
    } else if (block_threads == 1024u) {
        switch (inst) { MRT_BIG2(1024, 0u) MRT_BVH4(1024, 0u) MRT_IDENT_BVH2(1024) MRT_BIG2(1024, F_NOSTASH) MRT_BVH4(1024, F_NOSTASH)
                        MRT_BIG2(1024, F_COLD) MRT_BVH4(1024, F_COLD) MRT_DEEP2(1024) default: break; }
    }
    return hipErrorInvalidConfiguration;
}
#undef MRT_CASE
#undef MRT_CASE_L2

template <int THREADS, u32 FEAT>
static hipError_t set_lds_attr(int bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&pt_megakernel<true, THREADS, FEAT>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

hipError_t configure_pt(size_t max_lds_bytes)
{
    const int b = (int)max_lds_bytes;
    hipError_t e;
#define MRT_CASE(T, F) if ((e = set_lds_attr<T, (F)>(b)) != hipSuccess) return e;
    MRT_PLAIN16(64) MRT_BVH4(64, 0u)
    MRT_PLAIN16(256) MRT_IDENT4(256) MRT_BVH4(256, 0u) MRT_IDENT_BVH2(256) MRT_BIG2(256, F_COLD) MRT_BVH4(256, F_COLD) MRT_DEEP2(256)
    MRT_BIG2(512, 0u) MRT_BVH4(512, 0u) MRT_IDENT_BVH2(512) MRT_BIG2(512, F_COLD) MRT_BVH4(512, F_COLD) MRT_DEEP2(512)
    MRT_BIG2(1024, 0u) MRT_BVH4(1024, 0u) MRT_IDENT_BVH2(1024) MRT_BIG2(1024, F_NOSTASH) MRT_BVH4(1024, F_NOSTASH) MRT_BIG2(1024, F_COLD) MRT_BVH4(1024, F_COLD) MRT_DEEP2(1024)
#undef MRT_CASE
    return hipSuccess;
}

hipError_t launch_reduce_chunks(float *accum, const float *partial, size_t n_words, size_t stride, u32 n_chunks, hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_chunks, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, accum, partial, n_words, stride, n_chunks);
    return hipGetLastError();
}

hipError_t launch_scatter_rows(float *frame, const float *gathered, const u32 *rowmap, u32 n_rows, u32 row_words, hipStream_t stream)
{
    hipLaunchKernelGGL(scatter_rows, dim3((row_words + 255) / 256, n_rows), dim3(256), 0, stream, frame, gathered, rowmap, n_rows, row_words);
    return hipGetLastError();
}

hipError_t launch_tonemap(const float *accum, unsigned char *out, u32 n_px, float rc, float gamma, float wexp, hipStream_t stream)
{
    hipLaunchKernelGGL(tonemap_u8, dim3((n_px + 255) / 256), dim3(256), 0, stream, accum, out, n_px, rc, gamma, wexp);
    return hipGetLastError();
}

hipError_t launch_lanczos_v(const unsigned char *src, float *dst, u32 sw, u32 dh, const u32 *left, const u32 *count,
                            const float *weight, u32 cap, hipStream_t stream)
{
    hipLaunchKernelGGL(lanczos3_v, dim3((sw * 3u + 255) / 256, dh), dim3(256), 0, stream, src, dst, sw, dh, left, count, weight, cap);
    return hipGetLastError();
}

hipError_t launch_lanczos_h(const float *src, unsigned char *dst, u32 sw, u32 dw, u32 dh, const u32 *left, const u32 *count,
                            const float *weight, u32 cap, hipStream_t stream)
{
    hipLaunchKernelGGL(lanczos3_h, dim3((dw + 255) / 256, dh), dim3(256), 0, stream, src, dst, sw, dw, dh, left, count, weight, cap);
    return hipGetLastError();
}

hipError_t launch_math_selftest(int op, const float *a, const float *b, float *out, size_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(math_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError();
}

hipError_t launch_math_sweep(int op, unsigned long long first, unsigned long long n, u32 seed, unsigned long long *mismatches, float *example, hipStream_t stream)
{
    hipLaunchKernelGGL(math_sweep, dim3((unsigned)((n + 255ull) / 256ull)), dim3(256), 0, stream, op, first, n, seed, mismatches, example);
    return hipGetLastError();
}

}  // namespace mrt
