// probe.cpp — TEST INFRASTRUCTURE: divergence profile of the megakernel's per-lane loop, computed on the CPU.
// Every lane's loop iterations line up in lock-step inside a wavefront (the loop has one back-edge), so recording
// per lane which phases ran in its k-th iteration tells, per 8x8 wave tile, how many lanes were active whenever the
// wavefront had to execute a phase.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

static thread_local std::vector<uint16_t> *g_rec = nullptr;   // one bitmask per iteration
#define MRT_PROBE(phase) do { if (g_work && (phase) == 0) { g_work->push_back(0); if (g_work_any) g_work_any->push_back(0); if (g_work_r) { g_work_r->push_back(0); g_work_any_r->push_back(0); } } if (g_rec) { if ((phase) == 0) g_rec->push_back(1); else g_rec->back() |= (uint16_t)(1u << (phase)); } } while (0)

static thread_local uint64_t g_cnt[32];
static thread_local int g_fallbacks_shown = 0;
#define MRT_PROBE_FALLBACK(ro, rd, dd) do { if (getenv("MRT_EMU_SHOW_FALLBACKS") && g_fallbacks_shown++ < 30) fprintf(stderr, "fallback ray o = (%g %g %g) d = (%g %g %g) d.d = %.9g   [reference-walk work so far: %llu boxes, %llu triangles]\n", (ro).x, (ro).y, (ro).z, (rd).x, (rd).y, (rd).z, (double)(dd), (unsigned long long)g_cnt[14], (unsigned long long)g_cnt[13]); } while (0)
static thread_local std::vector<uint32_t> *g_work = nullptr;   // per loop iteration: BVH / TBVH nodes + 3 x exact tests of the lane
static thread_local std::vector<uint32_t> *g_work_any = nullptr;   // the same for the shadow queries of the iteration
static thread_local bool g_in_any = false;
static thread_local bool g_right = false;                           // walking the right half of a triangle BVH
static thread_local std::vector<uint32_t> *g_work_r = nullptr, *g_work_any_r = nullptr;   // triangle-BVH work in right halves
#define MRT_COUNT(counter) do { ++g_cnt[counter]; if ((counter) == 0) g_in_any = false; if ((counter) == 10) g_in_any = true; \
    if (g_work && !g_work->empty()) { std::vector<uint32_t> *w_ = (g_in_any && g_work_any) ? g_work_any : g_work; \
        if ((counter) == 2 || (counter) == 6) w_->back() += 1; else if ((counter) == 1 || (counter) == 7) w_->back() += 3; \
        if (g_work_r && g_right && ((counter) == 6 || (counter) == 7)) { std::vector<uint32_t> *r_ = g_in_any ? g_work_any_r : g_work_r; r_->back() += ((counter) == 6 ? 1 : 3); } } } while (0)

#include "../../micro_raytracer_amd/csrc/mrt_pack.h"
#include "../../micro_raytracer_amd/csrc/mrt_trace.h"
using namespace mrt;

extern "C" int probe_divergence(const mrt_render_desc *d, uint64_t seed, uint32_t n_samples, uint32_t tile_x0, uint32_t tile_y0,
                                uint32_t tiles_x, uint32_t tiles_y, double *active /*[PH_COUNT]*/, double *executed /*[PH_COUNT]*/)
{
    Packed pk; std::string err;
    if (pack_scene(d, pk, err)) return -1;
    Params P = pk.P;
    P.local_rows = pk.nh; P.shard_index = 0; P.shard_count = 1; P.shard_rows = 8;
    P.seed_lo = (u32)seed; P.seed_hi = (u32)(seed >> 32); P.n_samples = n_samples; P.sample_base = 0; P.k_split = 1;
    std::vector<float> frame((size_t)pk.nw * pk.nh * 3, 0.0f); P.accum = frame.data();
    Scn S; S.F = reinterpret_cast<const float *>(pk.blob.data()); S.U = S.F; S.G = S.F; S.P = &P;
    for (u32 p = 0; p < PH_COUNT; ++p) { active[p] = 0; executed[p] = 0; }
    for (uint32_t ty = tile_y0; ty < tile_y0 + tiles_y; ++ty)
        for (uint32_t tx = tile_x0; tx < tile_x0 + tiles_x; ++tx) {
            std::vector<std::vector<uint16_t>> rec(64);
            size_t max_it = 0;
            for (int l = 0; l < 64; ++l) {
                const uint32_t x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
                if (x >= pk.nw || y >= pk.nh) continue;
                g_rec = &rec[l];
                u32 sg = 0; RegStash st;
                LaneJob job; job.k = 0; job.word = (y * pk.nw + x) * 3u;
                if (pk.features & F_BVH) render_pixel<F_ALL | F_BVH>(S, st, x, y, job, sg); else render_pixel<F_ALL>(S, st, x, y, job, sg);
                g_rec = nullptr;
                if (rec[l].size() > max_it) max_it = rec[l].size();
            }
            for (size_t k = 0; k < max_it; ++k)
                for (u32 p = 0; p < PH_COUNT; ++p) {
                    int n = 0;
                    for (int l = 0; l < 64; ++l) if (k < rec[l].size() && (rec[l][k] >> p & 1)) ++n;
                    if (n) { active[p] += n; executed[p] += 64; }
                }
        }
    return 0;
}

// work counters (CT_* of mrt_trace.h) summed over every pixel of the frame, n_samples each
extern "C" int probe_counts(const mrt_render_desc *d, uint64_t seed, uint32_t n_samples, uint64_t *counts /*[CT_COUNT]*/, uint64_t *segments)
{
    Packed pk; std::string err;
    const bool deep = getenv("MRT_EMU_DEEP") != nullptr;           // the 4-wide walk of the F_DEEP kernels (one staged node)
    PackOpts po; po.tbvh_wide = deep;
    if (pack_scene(d, pk, err, po)) return -1;
    Params P = pk.P;
    if (deep) P.n_tbvh_hot = 1;
    P.local_rows = pk.nh; P.shard_index = 0; P.shard_count = 1; P.shard_rows = 8;
    P.seed_lo = (u32)seed; P.seed_hi = (u32)(seed >> 32); P.n_samples = n_samples; P.sample_base = 0; P.k_split = 1;
    std::vector<float> frame((size_t)pk.nw * pk.nh * 3, 0.0f); P.accum = frame.data();
    Scn S; S.F = reinterpret_cast<const float *>(pk.blob.data()); S.U = S.F; S.G = S.F; S.P = &P;
    if (const char *v = getenv("MRT_EMU_WALK_CAP")) { const int c = atoi(v); if (c >= 4 && c <= (int)kWalkCapMax) P.walk_cap = (u32)c; }
    memset(g_cnt, 0, sizeof g_cnt);
    uint64_t seg = 0;
    for (uint32_t y = 0; y < pk.nh; ++y)
        for (uint32_t x = 0; x < pk.nw; ++x) {
            u32 sg = 0; RegStash st;
            LaneJob job; job.k = 0; job.word = (y * pk.nw + x) * 3u;
            if (deep) { if (pk.features & F_BVH) render_pixel<F_ALL | F_BVH | F_COLD | F_DEEP>(S, st, x, y, job, sg); else render_pixel<F_ALL | F_COLD | F_DEEP>(S, st, x, y, job, sg); }
            else if (pk.features & F_BVH) render_pixel<F_ALL | F_BVH>(S, st, x, y, job, sg); else render_pixel<F_ALL>(S, st, x, y, job, sg);
            seg += sg;
        }
    for (u32 c = 0; c < CT_COUNT; ++c) counts[c] = g_cnt[c];
    if (segments) *segments = seg;
    return (int)CT_COUNT;
}

// SIMT efficiency of the traversal work inside one loop iteration: per 8x8 wave tile and iteration, the wavefront pays
// max over lanes of the lane's work units (nodes + 3 x exact tests); returns sum(mean over 64 lanes) / sum(max).
extern "C" double probe_traversal_efficiency(const mrt_render_desc *d, uint64_t seed, uint32_t n_samples, uint32_t tile_x0, uint32_t tile_y0,
                                             uint32_t tiles_x, uint32_t tiles_y, double *mean_work, double *max_work)
{
    Packed pk; std::string err;
    if (pack_scene(d, pk, err)) return -1;
    Params P = pk.P;
    P.local_rows = pk.nh; P.shard_index = 0; P.shard_count = 1; P.shard_rows = 8;
    P.seed_lo = (u32)seed; P.seed_hi = (u32)(seed >> 32); P.n_samples = n_samples; P.sample_base = 0; P.k_split = 1;
    std::vector<float> frame((size_t)pk.nw * pk.nh * 3, 0.0f); P.accum = frame.data();
    Scn S; S.F = reinterpret_cast<const float *>(pk.blob.data()); S.U = S.F; S.G = S.F; S.P = &P;
    double sum_mean = 0, sum_max = 0;
    for (uint32_t ty = tile_y0; ty < tile_y0 + tiles_y; ++ty)
        for (uint32_t tx = tile_x0; tx < tile_x0 + tiles_x; ++tx) {
            std::vector<std::vector<uint32_t>> w(64);
            size_t max_it = 0;
            for (int l = 0; l < 64; ++l) {
                const uint32_t x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
                if (x >= pk.nw || y >= pk.nh) continue;
                g_work = &w[l];
                u32 sg = 0; RegStash st;
                LaneJob job; job.k = 0; job.word = (y * pk.nw + x) * 3u;
                if (pk.features & F_BVH) render_pixel<F_ALL | F_BVH>(S, st, x, y, job, sg); else render_pixel<F_ALL>(S, st, x, y, job, sg);
                g_work = nullptr;
                if (w[l].size() > max_it) max_it = w[l].size();
            }
            for (size_t k = 0; k < max_it; ++k) {
                uint32_t mx = 0; double sm = 0;
                for (int l = 0; l < 64; ++l) if (k < w[l].size()) { sm += w[l][k]; if (w[l][k] > mx) mx = w[l][k]; }
                sum_mean += sm / 64.0; sum_max += mx;
            }
        }
    if (mean_work) *mean_work = sum_mean;
    if (max_work) *max_work = sum_max;
    return sum_max > 0 ? sum_mean / sum_max : 1.0;
}

// Per-lane work sequences of one 8x8 tile for offline scheduling studies: out[lane][iteration] = closest-hit work,
// out_any likewise for the shadow queries; returns the number of iterations written per lane in n_it[64].
extern "C" int probe_tile_work(const mrt_render_desc *d, uint64_t seed, uint32_t n_samples, uint32_t tx, uint32_t ty, uint32_t cap,
                               uint32_t *out /*[64][cap]*/, uint32_t *out_any /*[64][cap]*/, uint32_t *n_it /*[64]*/)
{
    Packed pk; std::string err;
    if (pack_scene(d, pk, err)) return -1;
    Params P = pk.P;
    P.local_rows = pk.nh; P.shard_index = 0; P.shard_count = 1; P.shard_rows = 8;
    P.seed_lo = (u32)seed; P.seed_hi = (u32)(seed >> 32); P.n_samples = n_samples; P.sample_base = 0; P.k_split = 1;
    std::vector<float> frame((size_t)pk.nw * pk.nh * 3, 0.0f); P.accum = frame.data();
    Scn S; S.F = reinterpret_cast<const float *>(pk.blob.data()); S.U = S.F; S.G = S.F; S.P = &P;
    for (int l = 0; l < 64; ++l) {
        n_it[l] = 0;
        const uint32_t x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
        if (x >= pk.nw || y >= pk.nh) continue;
        std::vector<uint32_t> w, wa;
        g_work = &w; g_work_any = &wa;
        u32 sg = 0; RegStash st;
        LaneJob job; job.k = 0; job.word = (y * pk.nw + x) * 3u;
        if (pk.features & F_BVH) render_pixel<F_ALL | F_BVH>(S, st, x, y, job, sg); else render_pixel<F_ALL>(S, st, x, y, job, sg);
        g_work = nullptr; g_work_any = nullptr;
        const size_t n = w.size() < cap ? w.size() : cap;
        for (size_t k = 0; k < n; ++k) { out[(size_t)l * cap + k] = w[k]; out_any[(size_t)l * cap + k] = wa[k]; }
        n_it[l] = (uint32_t)n;
    }
    return 0;
}

// probe_tile_work plus the part of the work that lies in the right half of the triangle BVHs (lane-pair splitting study)
extern "C" int probe_tile_work_split(const mrt_render_desc *d, uint64_t seed, uint32_t n_samples, uint32_t tx, uint32_t ty, uint32_t cap,
                                     uint32_t *out, uint32_t *out_any, uint32_t *out_r, uint32_t *out_any_r, uint32_t *n_it)
{
    Packed pk; std::string err;
    if (pack_scene(d, pk, err)) return -1;
    Params P = pk.P;
    P.local_rows = pk.nh; P.shard_index = 0; P.shard_count = 1; P.shard_rows = 8;
    P.seed_lo = (u32)seed; P.seed_hi = (u32)(seed >> 32); P.n_samples = n_samples; P.sample_base = 0; P.k_split = 1;
    std::vector<float> frame((size_t)pk.nw * pk.nh * 3, 0.0f); P.accum = frame.data();
    Scn S; S.F = reinterpret_cast<const float *>(pk.blob.data()); S.U = S.F; S.G = S.F; S.P = &P;
    for (int l = 0; l < 64; ++l) {
        n_it[l] = 0;
        const uint32_t x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
        if (x >= pk.nw || y >= pk.nh) continue;
        std::vector<uint32_t> w, wa, wr, war;
        g_work = &w; g_work_any = &wa; g_work_r = &wr; g_work_any_r = &war;
        u32 sg = 0; RegStash st;
        LaneJob job; job.k = 0; job.word = (y * pk.nw + x) * 3u;
        if (pk.features & F_BVH) render_pixel<F_ALL | F_BVH>(S, st, x, y, job, sg); else render_pixel<F_ALL>(S, st, x, y, job, sg);
        g_work = nullptr; g_work_any = nullptr; g_work_r = nullptr; g_work_any_r = nullptr;
        const size_t n = w.size() < cap ? w.size() : cap;
        for (size_t k = 0; k < n; ++k) { out[(size_t)l * cap + k] = w[k]; out_any[(size_t)l * cap + k] = wa[k]; out_r[(size_t)l * cap + k] = wr[k]; out_any_r[(size_t)l * cap + k] = war[k]; }
        n_it[l] = (uint32_t)n;
    }
    return 0;
}
