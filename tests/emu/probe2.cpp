// probe2.cpp — TEST INFRASTRUCTURE: which lanes run the sphere root arithmetic (sqrt + two divisions) for which
// instance in each loop iteration of an 8x8 wave tile: what the wavefront pays today (one masked pass per sphere that
// any lane needs) against a compacted schedule (one pass per pending sphere of the busiest lane).
#include <stdint.h>
#include <string>
#include <vector>

static thread_local std::vector<uint32_t> *g_mask = nullptr;   // per iteration: bit i = sphere math ran for instance i
static thread_local uint32_t g_inst = 0;
static thread_local bool g_shadow = false;
#define MRT_PROBE_INST(i) (g_inst = (i))
#define MRT_PROBE(phase) do { if (g_mask) { if ((phase) == 0) g_mask->push_back(0); else if ((phase) == 2 && !g_mask->empty() && g_inst < 32) g_mask->back() |= 1u << g_inst; } } while (0)

#include "../../micro_raytracer_amd/csrc/mrt_pack.h"
#include "../../micro_raytracer_amd/csrc/mrt_trace.h"
using namespace mrt;

// out[0] = wave iterations, out[1] = masked sphere passes paid today, out[2] = passes of the compacted schedule,
// out[3] = lane-level sphere evaluations, out[4] = lane iterations
extern "C" int probe_sphere_math(const mrt_render_desc *d, uint64_t seed, uint32_t n_samples, uint32_t tile_x0, uint32_t tile_y0,
                                 uint32_t tiles_x, uint32_t tiles_y, double *out)
{
    Packed pk; std::string err;
    if (pack_scene(d, pk, err)) return -1;
    Params P = pk.P;
    P.local_rows = pk.nh; P.shard_index = 0; P.shard_count = 1; P.shard_rows = 8;
    P.seed_lo = (u32)seed; P.seed_hi = (u32)(seed >> 32); P.n_samples = n_samples; P.sample_base = 0; P.k_split = 1;
    std::vector<float> frame((size_t)pk.nw * pk.nh * 3, 0.0f); P.accum = frame.data();
    Scn S; S.F = reinterpret_cast<const float *>(pk.blob.data()); S.U = S.F; S.G = S.F; S.P = &P;
    for (int i = 0; i < 5; ++i) out[i] = 0;
    for (uint32_t ty = tile_y0; ty < tile_y0 + tiles_y; ++ty)
        for (uint32_t tx = tile_x0; tx < tile_x0 + tiles_x; ++tx) {
            std::vector<std::vector<uint32_t>> rec(64);
            size_t max_it = 0;
            for (int l = 0; l < 64; ++l) {
                const uint32_t x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
                if (x >= pk.nw || y >= pk.nh) continue;
                g_mask = &rec[l];
                u32 sg = 0; RegStash st;
                LaneJob job; job.k = 0; job.word = (y * pk.nw + x) * 3u;
                render_pixel<0>(S, st, x, y, job, sg);
                g_mask = nullptr;
                if (rec[l].size() > max_it) max_it = rec[l].size();
            }
            for (size_t k = 0; k < max_it; ++k) {
                uint32_t any = 0; int mx = 0;
                for (int l = 0; l < 64; ++l) if (k < rec[l].size()) {
                    any |= rec[l][k];
                    const int pc = __builtin_popcount(rec[l][k]);
                    if (pc > mx) mx = pc;
                    out[3] += pc; out[4] += 1;
                }
                out[0] += 1; out[1] += __builtin_popcount(any); out[2] += mx;
            }
        }
    return 0;
}
