// walk_probe.cpp -- TEST INFRASTRUCTURE (not collected by pytest): the distribution of triangle-BVH walk lengths per mesh query,
// and the rays behind the longest ones.  A wavefront waits for the lane with the longest walk, so a handful of queries that
// visit most of the tree can hold a workgroup slot for milliseconds at the end of a launch.
//   g++ -O2 -mfma -std=c++17 -fPIC -ffp-contract=off -shared -o libwalk.so walk_probe.cpp ../../micro_raytracer_amd/csrc/mrt_pack.cpp -lpthread
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
static thread_local uint32_t g_nodes = 0, g_tris = 0;
static thread_local uint64_t g_hist[8], g_calls = 0, g_total_nodes = 0;
static thread_local float g_ray[7];
static thread_local uint32_t g_worst = 0;
static thread_local float g_worst_ray[7];
static void flush_call()
{
    if (g_calls) {
        const uint32_t w = g_nodes + 3 * g_tris;
        int b = 0;
        for (uint32_t lim = 32; b < 7 && w > lim; lim *= 4) ++b;
        g_hist[b]++;
        g_total_nodes += w;
        if (w > g_worst) { g_worst = w; for (int k = 0; k < 7; ++k) g_worst_ray[k] = g_ray[k]; }
    }
    g_nodes = 0; g_tris = 0;
}
#define MRT_PROBE_FALLBACK(ro, rd, dd) do { g_ray[0] = (ro).x; g_ray[1] = (ro).y; g_ray[2] = (ro).z; g_ray[3] = (rd).x; g_ray[4] = (rd).y; g_ray[5] = (rd).z; g_ray[6] = (dd); } while (0)
#define MRT_COUNT(counter) do { if ((counter) == 4) { flush_call(); ++g_calls; } else if ((counter) == 6 || (counter) == 14) ++g_nodes; else if ((counter) == 7 || (counter) == 13) ++g_tris; } while (0)
#include "../../micro_raytracer_amd/csrc/mrt_pack.h"
#include "../../micro_raytracer_amd/csrc/mrt_trace.h"
using namespace mrt;

extern "C" int probe_walks(const mrt_render_desc *d, uint64_t seed, uint32_t s0, uint32_t n_samples, int deep, uint64_t *hist /*[8]*/, double *out /*[10]*/)
{
    Packed pk; std::string err;
    PackOpts po; po.tbvh_wide = deep != 0;
    if (pack_scene(d, pk, err, po)) return -1;
    Params P = pk.P;
    if (deep) P.n_tbvh_hot = 1;
    P.local_rows = pk.nh; P.shard_index = 0; P.shard_count = 1; P.shard_rows = 8;
    P.seed_lo = (u32)seed; P.seed_hi = (u32)(seed >> 32); P.n_samples = n_samples; P.sample_base = s0; P.k_split = 1;
    std::vector<float> frame((size_t)pk.nw * pk.nh * 3, 0.0f); P.accum = frame.data();
    Scn S; S.F = reinterpret_cast<const float *>(pk.blob.data()); S.U = S.F; S.G = S.F; S.P = &P;
    for (int k = 0; k < 8; ++k) g_hist[k] = 0;
    g_calls = 0; g_total_nodes = 0; g_worst = 0;
    for (uint32_t y = 0; y < pk.nh; ++y)
        for (uint32_t x = 0; x < pk.nw; ++x) {
            u32 sg = 0; RegStash st; LaneJob job; job.k = 0; job.word = (y * pk.nw + x) * 3u;
            if (deep) render_pixel<F_ALL | F_COLD | F_DEEP>(S, st, x, y, job, sg); else render_pixel<F_ALL>(S, st, x, y, job, sg);
        }
    flush_call();
    for (int k = 0; k < 8; ++k) hist[k] = g_hist[k];
    out[0] = (double)g_calls; out[1] = (double)g_total_nodes; out[2] = g_worst;
    for (int k = 0; k < 7; ++k) out[3 + k] = g_worst_ray[k];
    return 0;
}
