// round_probe.cpp -- TEST INFRASTRUCTURE (not collected by pytest): what a wavefront PAYS for the triangle-BVH walks of
// its 64 lanes.  The per-lane walk of mrt_trace.h runs in rounds (box steps until two leaves are in hand, then the exact
// tests of those leaves); lanes of a wavefront go through the rounds of one mesh_isect call in lock-step, so a round costs
// the wavefront max(box steps) + max(triangles), whatever the average lane needs.  Records, per lane, loop iteration and
// mesh_isect call, the (steps, triangles) of every round and folds them per 8x8 wave tile.
#include <stdint.h>
#include <string>
#include <vector>
struct Rd { uint16_t steps, tris; };
struct Call { std::vector<Rd> rounds; };
static thread_local std::vector<std::vector<Call>> *g_iters = nullptr;     // [iteration][call]
#define MRT_PROBE(phase) do { if (g_iters && (phase) == 0) g_iters->push_back(std::vector<Call>()); } while (0)
#define MRT_COUNT(counter) do { if (g_iters && !g_iters->empty() && (counter) == 4) g_iters->back().push_back(Call()); } while (0)
#define MRT_PROBE_ROUND(steps, tris, membs) do { if (g_iters && !g_iters->empty() && !g_iters->back().empty()) g_iters->back().back().rounds.push_back(Rd{(uint16_t)(steps), (uint16_t)(tris)}); } while (0)
#include "../../micro_raytracer_amd/csrc/mrt_pack.h"
#include "../../micro_raytracer_amd/csrc/mrt_trace.h"
using namespace mrt;
#ifndef ROUND_FEAT                  // -DROUND_FEAT="(F_ALL | F_COLD)": the warm mesh kernel (every closest-hit leaf queued)
#define ROUND_FEAT F_ALL
#endif

// out: [0] wave iterations, [1] sum over rounds of max steps, [2] sum of max tris, [3] wave rounds, [4] sum of lane steps / 64, [5] sum of lane tris / 64,
//      [6] mesh_isect call slots executed by the wave, [7] lane-calls / 64
extern "C" int probe_rounds(const mrt_render_desc *d, uint64_t seed, uint32_t n_samples, uint32_t tx, uint32_t ty, double *out)
{
    Packed pk; std::string err;
    if (pack_scene(d, pk, err)) return -1;
    Params P = pk.P;
    P.local_rows = pk.nh; P.shard_index = 0; P.shard_count = 1; P.shard_rows = 8;
    P.seed_lo = (u32)seed; P.seed_hi = (u32)(seed >> 32); P.k_split = 1; P.sample_base = 0; P.n_samples = n_samples;
    std::vector<float> frame((size_t)pk.nw * pk.nh * 3, 0.0f); P.accum = frame.data();
    Scn S; S.F = reinterpret_cast<const float *>(pk.blob.data()); S.U = S.F; S.G = S.F; S.P = &P;
    std::vector<std::vector<std::vector<Call>>> rec(64);
    size_t mx = 0;
    for (int l = 0; l < 64; ++l) {
        g_iters = &rec[l];
        u32 sg = 0; RegStash st; LaneJob job; job.k = 0; job.word = ((ty * 8 + (l >> 3)) * pk.nw + tx * 8 + (l & 7)) * 3u;
        render_pixel<ROUND_FEAT>(S, st, tx * 8 + (l & 7), ty * 8 + (l >> 3), job, sg);
        g_iters = nullptr;
        if (rec[l].size() > mx) mx = rec[l].size();
    }
    for (int i = 0; i < 16; ++i) out[i] = 0;
    for (size_t k = 0; k < mx; ++k) {
        out[0] += 1;
        // calls line up by index within the iteration only if every lane makes the same calls; lanes skip calls (no shadow
        // query, ray misses...), and the scan is in lock-step over instances: call slot c = the c-th mesh_isect call of the lane.
        // Closest-hit and shadow calls are distinguished by order: this scene has ONE mesh instance, so a lane makes at most
        // two calls per iteration (closest, then shadow).  Lanes that make only the closest call: slot 0.  A lane that makes
        // only ... the shadow call cannot exist (a shadow query follows a hit).  So slots line up.
        for (size_t c = 0; c < 2; ++c) {
            size_t nr = 0; bool anyc = false;
            for (int l = 0; l < 64; ++l) if (k < rec[l].size() && c < rec[l][k].size()) { anyc = true; if (rec[l][k][c].rounds.size() > nr) nr = rec[l][k][c].rounds.size(); out[7] += 1.0 / 64; }
            if (anyc) out[6] += 1;
            {   // the same call with every leaf postponed: one box phase (max over lanes of the lane's total steps), one exact phase
                unsigned ts = 0, tt = 0;
                for (int l = 0; l < 64; ++l) if (k < rec[l].size() && c < rec[l][k].size()) {
                    unsigned a = 0, b = 0;
                    for (const Rd &q : rec[l][k][c].rounds) { a += q.steps; b += q.tris; }
                    if (a > ts) ts = a;
                    if (b > tt) tt = b;
                }
                out[8 + 2 * c] += ts; out[9 + 2 * c] += tt;
            }
            for (size_t r = 0; r < nr; ++r) {
                unsigned ms = 0, mt = 0;
                for (int l = 0; l < 64; ++l) if (k < rec[l].size() && c < rec[l][k].size() && r < rec[l][k][c].rounds.size()) {
                    const Rd &q = rec[l][k][c].rounds[r];
                    if (q.steps > ms) ms = q.steps;
                    if (q.tris > mt) mt = q.tris;
                    out[4] += q.steps / 64.0; out[5] += q.tris / 64.0;
                }
                out[1] += ms; out[2] += mt; out[3] += 1;
                out[12 + 2 * c] += ms; out[13 + 2 * c] += mt;
            }
        }
    }
    return 0;
}
