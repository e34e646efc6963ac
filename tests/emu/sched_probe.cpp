// sched_probe.cpp -- TEST INFRASTRUCTURE (not collected by pytest): per-path work sequences and per-lane visited-node sets
// of the megakernel's per-lane loop, recorded on the CPU, for the schedule models of tests/emu/sched_models.py
// (what intra-wavefront rescheduling of the BVH scenes could and could not buy).
//   probe_paths   per (pixel, sample): closest-hit and shadow work of every path segment
//                 (work = BVH / triangle-BVH nodes + 3 x exact tests + membership boxes)
//   probe_union   per 8x8 wave tile and loop iteration: max and union over the lanes of the triangle-BVH nodes visited
#include <stdint.h>
#include <set>
#include <string>
#include <vector>
struct It { std::vector<uint32_t> c, s; };
static thread_local std::vector<uint32_t> *g_work = nullptr, *g_any = nullptr;
static thread_local std::vector<It> *g_rec = nullptr;
static thread_local bool g_in_any = false;
#define MRT_PROBE(phase) do { if ((phase) == 0) { if (g_work) { g_work->push_back(0); g_any->push_back(0); } if (g_rec) g_rec->push_back(It()); } } while (0)
#define MRT_COUNT(counter) do { if ((counter) == 0) g_in_any = false; if ((counter) == 10) g_in_any = true; \
    if (g_work && !g_work->empty()) { std::vector<uint32_t> *w_ = g_in_any ? g_any : g_work; \
        if ((counter) == 2 || (counter) == 6 || (counter) == 9) w_->back() += 1; else if ((counter) == 1 || (counter) == 7) w_->back() += 3; } } while (0)
#define MRT_PROBE_TBVH_PART(node, right0) do { if (g_rec && !g_rec->empty()) (g_in_any ? g_rec->back().s : g_rec->back().c).push_back(node); } while (0)
#include "../../micro_raytracer_amd/csrc/mrt_pack.h"
#include "../../micro_raytracer_amd/csrc/mrt_trace.h"
using namespace mrt;

static void setup(const Packed &pk, Params &P, std::vector<float> &frame, Scn &S, uint64_t seed)
{
    P = pk.P;
    P.local_rows = pk.nh; P.shard_index = 0; P.shard_count = 1; P.shard_rows = 8;
    P.seed_lo = (u32)seed; P.seed_hi = (u32)(seed >> 32); P.k_split = 1;
    frame.assign((size_t)pk.nw * pk.nh * 3, 0.0f); P.accum = frame.data();
    S.F = reinterpret_cast<const float *>(pk.blob.data()); S.U = S.F; S.G = S.F; S.P = &P;
}
static void run(const Packed &pk, const Scn &S, uint32_t x, uint32_t y)
{
    u32 sg = 0; RegStash st; LaneJob job; job.k = 0; job.word = (y * pk.nw + x) * 3u;
    if (pk.features & F_BVH) render_pixel<F_ALL | F_BVH>(S, st, x, y, job, sg); else render_pixel<F_ALL>(S, st, x, y, job, sg);
}

// out[((y - y0) * w + (x - x0)) * n_samples + s][k] = closest work | shadow work << 16 of segment k; n_it likewise
extern "C" int probe_paths(const mrt_render_desc *d, uint64_t seed, uint32_t n_samples, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t cap,
                           uint32_t *out, uint32_t *n_it)
{
    Packed pk; std::string err;
    if (pack_scene(d, pk, err)) return -1;
    Params P; std::vector<float> frame; Scn S;
    setup(pk, P, frame, S, seed);
    size_t idx = 0;
    for (uint32_t y = y0; y < y0 + h; ++y) for (uint32_t x = x0; x < x0 + w; ++x)
        for (uint32_t s = 0; s < n_samples; ++s, ++idx) {
            std::vector<uint32_t> wk, wa;
            g_work = &wk; g_any = &wa;
            P.sample_base = s; P.n_samples = 1;
            run(pk, S, x, y);
            g_work = nullptr; g_any = nullptr;
            const size_t n = wk.size() < cap ? wk.size() : cap;
            for (size_t k = 0; k < n; ++k) out[idx * cap + k] = (wk[k] & 0xffff) | (wa[k] << 16);
            n_it[idx] = (uint32_t)n;
        }
    return 0;
}

// out: [0] wave iterations, [1] sum of max (closest), [2] sum of union (closest), [3] max (shadow), [4] union (shadow),
//      [5] sum of mean (closest), [6] mean (shadow)
extern "C" int probe_union(const mrt_render_desc *d, uint64_t seed, uint32_t n_samples, uint32_t tx, uint32_t ty, double *out)
{
    Packed pk; std::string err;
    if (pack_scene(d, pk, err)) return -1;
    Params P; std::vector<float> frame; Scn S;
    setup(pk, P, frame, S, seed);
    P.sample_base = 0; P.n_samples = n_samples;
    std::vector<std::vector<It>> rec(64);
    size_t mx = 0;
    for (int l = 0; l < 64; ++l) {
        g_rec = &rec[l];
        run(pk, S, tx * 8 + (l & 7), ty * 8 + (l >> 3));
        g_rec = nullptr;
        if (rec[l].size() > mx) mx = rec[l].size();
    }
    for (int i = 0; i < 7; ++i) out[i] = 0;
    for (size_t k = 0; k < mx; ++k) {
        std::set<uint32_t> uc, us; size_t mc = 0, ms = 0; double sc = 0, ss = 0;
        for (int l = 0; l < 64; ++l) if (k < rec[l].size()) {
            const It &it = rec[l][k];
            uc.insert(it.c.begin(), it.c.end()); us.insert(it.s.begin(), it.s.end());
            if (it.c.size() > mc) mc = it.c.size();
            if (it.s.size() > ms) ms = it.s.size();
            sc += it.c.size(); ss += it.s.size();
        }
        out[0] += 1; out[1] += mc; out[2] += uc.size(); out[3] += ms; out[4] += us.size(); out[5] += sc / 64.0; out[6] += ss / 64.0;
    }
    return 0;
}
