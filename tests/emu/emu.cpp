// emu.cpp — TEST INFRASTRUCTURE ONLY.  Compiles the device headers of the HIP kernels
// (mrt_trace.h, mrt_post.h, mrt_math.h) and the host packer for x86 so that `-m "not gpu"` tests
// can check the kernel's per-lane logic and the packed scene layout against the CPU oracle
// without a GPU.  Nothing in the product loads this library; libmrt_hip.so has no CPU path.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>
#include <atomic>

#include "../../micro_raytracer_amd/csrc/mrt_pack.h"
#include "../../micro_raytracer_amd/csrc/mrt_post.h"
#include "../../micro_raytracer_amd/csrc/mrt_trace.h"

using namespace mrt;

static thread_local std::string g_err;

// entries of a lane's walk area (mrt_trace.h): MRT_EMU_WALK_CAP in the environment, so that tests can shrink it until walks
// flush early and overflow into the reference's walk
static uint32_t walk_cap()
{
    const char *v = getenv("MRT_EMU_WALK_CAP");
    const int c = v ? atoi(v) : (int)kWalkCapDefault;
    return c < 4 ? 4u : (c > (int)kWalkCapMax ? kWalkCapMax : (uint32_t)c);
}

extern "C" {

const char *emu_error(void) { return g_err.c_str(); }

// pack only: returns code, fills sizes (blob words, nw, nh)
int emu_pack(const mrt_render_desc *d, uint32_t *blob_words, uint32_t *nw, uint32_t *nh, uint32_t *info /*[8]*/)
{
    Packed pk;
    const int rc = pack_scene(d, pk, g_err);
    if (rc) return rc;
    if (blob_words) *blob_words = pk.P.blob_words;
    if (nw) *nw = pk.nw;
    if (nh) *nh = pk.nh;
    if (info) { info[0] = pk.n_tex_u8; info[1] = pk.n_tex_f32; info[2] = pk.n_nodes; info[3] = pk.n_leaf_ids; info[4] = pk.n_tris; info[5] = pk.n_xf; info[6] = pk.P.n_inst; info[7] = pk.P.n_rend; }
    return 0;
}

// word offsets of the packed tables (mrt_scene.h Params.off_*), in blob order, + lds_words and blob_words: layout tests
int emu_layout(const mrt_render_desc *d, uint32_t *out /*[32]*/)
{
    Packed pk;
    const int rc = pack_scene(d, pk, g_err);
    if (rc) return rc;
    const Params &P = pk.P;
    const uint32_t v[] = {P.off_rend, P.off_cam, P.off_lin, P.off_bvh, P.off_bvhinst, P.off_inst, P.off_instx, P.off_xf, P.off_mat, P.off_light, P.off_tex,
                          P.off_lut, P.off_mesh, P.off_node, P.off_parent, P.off_tbvh, P.off_tri, P.off_memb, P.off_membe, P.off_leaf, P.lds_words, P.blob_words,
                          pk.n_tbvh_nodes, pk.n_nodes, pk.n_tris, pk.n_leaf_ids, pk.n_bvh_nodes, pk.features, P.lds_words_hot};
    for (size_t i = 0; i < sizeof v / sizeof v[0]; ++i) out[i] = v[i];
    return 0;
}

int emu_features(const mrt_render_desc *d)
{
    Packed pk;
    if (pack_scene(d, pk, g_err)) return -1;
    return (int)pk.features;
}

// the megakernel's per-lane body over every pixel of rows [row0,row1) (frame rows), accumulating into accum[nh][nw][3]
// deep_nodes > 0: the F_COLD | F_DEEP form of the lane code on a level-ordered triangle-BVH table of which the first deep_nodes
// nodes count as staged (on the CPU both halves are the same memory: this checks the ordering, the child links and the walk)
int emu_render_deep(const mrt_render_desc *d, uint64_t seed, uint32_t sample_base, uint32_t n_samples, uint32_t row0, uint32_t row1,
                    uint32_t threads, float *accum, uint64_t *segments, uint32_t deep_nodes)
{
    Packed pk;
    const bool warm_only = deep_nodes == 0xffffffffu;      // F_COLD without F_DEEP: the queued closest-hit walk on the binary table
    if (warm_only) deep_nodes = 0;
    PackOpts po;
    po.tbvh_wide = deep_nodes != 0;
    const int rc = pack_scene(d, pk, g_err, po);
    if (rc) return rc;
    if (deep_nodes && !pk.tbvh_wide) { g_err = "no triangle BVH to widen"; return -100; }
    Params P = pk.P;
    P.n_tbvh_hot = deep_nodes;
    if (deep_nodes) P.walk_cap = walk_cap();
    P.local_rows = pk.nh; P.shard_index = 0; P.shard_count = 1; P.shard_rows = 8;
    P.seed_lo = (u32)seed; P.seed_hi = (u32)(seed >> 32);
    P.n_samples = n_samples; P.sample_base = sample_base; P.k_split = 1; P.accum = accum;
    if (row1 > pk.nh) row1 = pk.nh;
    Scn S;
    S.F = reinterpret_cast<const float *>(pk.blob.data());
    S.U = S.F; S.G = S.F;
    S.P = &P;
    std::atomic<uint32_t> next(row0);
    std::atomic<uint64_t> segs(0);
    if (threads == 0) threads = 1;
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < threads; ++t) pool.emplace_back([&]() {
        uint64_t local = 0;
        for (;;) {
            const uint32_t y = next.fetch_add(1);
            if (y >= row1) break;
            for (uint32_t x = 0; x < pk.nw; ++x) {
                u32 sg = 0;
                {
                    RegStash st; LaneJob job; job.k = 0; job.word = (y * pk.nw + x) * 3u;
                    if (warm_only) { if (pk.features & F_BVH) render_pixel<F_ALL | F_BVH | F_COLD>(S, st, x, y, job, sg); else render_pixel<F_ALL | F_COLD>(S, st, x, y, job, sg); }
                    else if (deep_nodes) { if (pk.features & F_BVH) render_pixel<F_ALL | F_BVH | F_COLD | F_DEEP>(S, st, x, y, job, sg); else render_pixel<F_ALL | F_COLD | F_DEEP>(S, st, x, y, job, sg); }
                    else if (pk.features & F_BVH) render_pixel<F_ALL | F_BVH>(S, st, x, y, job, sg); else render_pixel<F_ALL>(S, st, x, y, job, sg);
                }
                local += sg;
            }
        }
        segs += local;
    });
    for (auto &th : pool) th.join();
    if (segments) *segments = segs.load();
    return 0;
}

int emu_render(const mrt_render_desc *d, uint64_t seed, uint32_t sample_base, uint32_t n_samples, uint32_t row0, uint32_t row1,
               uint32_t threads, float *accum, uint64_t *segments)
{
    return emu_render_deep(d, seed, sample_base, n_samples, row0, row1, threads, accum, segments, 0);
}

// Sampler::img through the kernels' per-element bodies
int emu_img(const mrt_render_desc *d, const float *accum, uint32_t count, uint8_t *out_ss, uint8_t *out)
{
    Packed pk;
    const int rc = pack_scene(d, pk, g_err);
    if (rc) return rc;
    const u32 nw = pk.nw, nh = pk.nh, rw = pk.res_w, rh = pk.res_h;
    const float rcnt = 1.0f / (float)count;
    const float wexp = (1.0f - pk.exp) * (1.0f - pk.exp);
    std::vector<uint8_t> ss((size_t)nw * nh * 3);
    for (size_t i = 0; i < ss.size(); ++i) ss[i] = tonemap_channel(accum[i], rcnt, pk.gamma, wexp);
    if (out_ss) memcpy(out_ss, ss.data(), ss.size());
    if (!out) return 0;
    if (rw == nw && rh == nh) { memcpy(out, ss.data(), ss.size()); return 0; }
    ResampleTaps v, h;
    lanczos3_taps(nh, rh, v);
    lanczos3_taps(nw, rw, h);
    std::vector<float> tmp((size_t)nw * rh * 3);
    for (u32 oy = 0; oy < rh; ++oy)
        for (u32 e = 0; e < nw * 3; ++e) {
            float t = 0.0f;
            for (u32 i = 0; i < v.count[oy]; ++i) t += (float)ss[(size_t)(v.left[oy] + i) * nw * 3 + e] * v.weight[(size_t)oy * v.cap + i];
            tmp[(size_t)oy * nw * 3 + e] = t;
        }
    for (u32 y = 0; y < rh; ++y)
        for (u32 ox = 0; ox < rw; ++ox) {
            float t[3] = {0.0f, 0.0f, 0.0f};
            for (u32 i = 0; i < h.count[ox]; ++i) {
                const float *p = tmp.data() + ((size_t)y * nw + (h.left[ox] + i)) * 3;
                const float w = h.weight[(size_t)ox * h.cap + i];
                t[0] += p[0] * w; t[1] += p[1] * w; t[2] += p[2] * w;
            }
            for (int k = 0; k < 3; ++k) out[((size_t)y * rw + ox) * 3 + k] = resample_to_u8(t[k]);
        }
    return 0;
}

void emu_math(int op, const float *a, const float *b, float *out, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        const float x = a[i], y = b ? b[i] : 0.0f;
        float s, c, r = 0.0f;
        switch (op) {
        case 0: sincos_(x, s, c); r = s; break;
        case 1: sincos_(x, s, c); r = c; break;
        case 2: r = acos_(x); break;
        case 3: r = atan2_(x, y); break;
        case 4: r = pow_(x, y); break;
        case 5: r = 1.0f / x; break;
        case 6: r = sqrt_(x); break;
        case 7: r = x / y; break;
        case 8: r = fmax_(x, y); break;
        case 9: r = fmin_(x, y); break;
        case 10: r = u2f((u32)total_key(x)); break;
        case 11: r = u32_to_unit(draw_u32(f2u(x), f2u(y))); break;
        case 12: r = norm(v3(x, y, 0.25f)).x; break;
        default: break;
        }
        out[i] = r;
    }
}

// Mesh arm of Renderer::intersect for n rays against renderer 0 (a mesh), through the TBVH route and through the reference's
// octree walk (forced by dd = NaN, which only steers the route).  out[i*10..]: hit, t0 bits, i0, t1 bits, i1 for each route.
// A third route walks the 4-wide table of the same mesh (F_DEEP lane code).
// Returns the number of rays on which the routes differ (ANY queries included), or < 0.
int emu_mesh_probe(const mrt_render_desc *d, uint32_t n, const float *orig, const float *dir, uint32_t *out, uint32_t *stats /*[2]*/)
{
    Packed pk;
    const int rc = pack_scene(d, pk, g_err);
    if (rc) return rc;
    Params P = pk.P;
    Scn S;
    S.F = reinterpret_cast<const float *>(pk.blob.data());
    S.U = S.F; S.G = S.F; S.P = &P;
    if (P.n_inst == 0) return -100;
    int bad = 0;
    uint32_t hits = 0, with_tbvh = 0;
    const float *M = S.F + P.off_mesh;
    with_tbvh = ldu(M, MESH_TBVH) != NO_NODE;
    // the same mesh with the 4-wide table (what the F_DEEP kernels walk): a third route through the same exact tests
    Packed pkw;
    PackOpts po; po.tbvh_wide = true;
    if (pack_scene(d, pkw, g_err, po)) return -101;
    Params Pw = pkw.P;
    Pw.n_tbvh_hot = 1; Pw.walk_cap = walk_cap();
    Scn Sw;
    Sw.F = reinterpret_cast<const float *>(pkw.blob.data());
    Sw.U = Sw.F; Sw.G = Sw.F; Sw.P = &Pw;
    for (uint32_t i = 0; i < n; ++i) {
        const V3 o = v3(orig[i * 3], orig[i * 3 + 1], orig[i * 3 + 2]), dr = v3(dir[i * 3], dir[i * 3 + 1], dir[i * 3 + 2]);
        const RayPre ray = ray_pre<F_ALL>(o, dr);
        uint32_t r[3][5];
        bool anyq[3];
        for (int route = 0; route < 3; ++route) {
            const Scn &SS = route == 2 ? Sw : S;
            const float *I = SS.F + SS.P->off_inst;
            const F4 ia = ld4(I, 0);
            const V3 pos = v3(ia.x, ia.y, ia.z);
            const float dd = route != 1 ? ray.dd : __builtin_nanf("");
            float t0 = 0, t1 = 0; i32 i0 = -1, i1 = -1;
            const V3 ro = add(pos, sub(ray.o, pos));
            float u0 = 0, u1 = 0; i32 j0 = -1, j1 = -1;
            bool h;
            if (route == 2) {
                h = mesh_isect<false, F_ALL | F_COLD | F_DEEP>(SS, 0, ro, ray.d, dd, ray.m, pos, t0, i0, t1, i1);
                anyq[route] = mesh_isect<true, F_ALL | F_COLD | F_DEEP>(SS, 0, ro, ray.d, dd, ray.m, pos, u0, j0, u1, j1);
            } else {
                h = mesh_isect<false, F_ALL>(SS, 0, ro, ray.d, dd, ray.m, pos, t0, i0, t1, i1);
                anyq[route] = mesh_isect<true, F_ALL>(SS, 0, ro, ray.d, dd, ray.m, pos, u0, j0, u1, j1);
            }
            r[route][0] = h; r[route][1] = h ? f2u(t0) : 0; r[route][2] = h ? (u32)i0 : 0; r[route][3] = h ? f2u(t1) : 0; r[route][4] = h ? (u32)i1 : 0;
        }
        {   // the reference's own walk, called directly: also checks the shortcut mesh_isect takes for all-NaN directions
            const float *I = S.F + S.P->off_inst;
            const F4 ia = ld4(I, 0);
            const V3 pos = v3(ia.x, ia.y, ia.z);
            float t0 = 0, t1 = 0; i32 i0 = -1, i1 = -1;
            const V3 ro = add(pos, sub(ray.o, pos));
            const bool h = mesh_isect_ref<false, F_ALL>(S, 0, ro, ray.d, ray.m, pos, t0, i0, t1, i1);
            uint32_t q[5] = {h, h ? f2u(t0) : 0u, h ? (u32)i0 : 0u, h ? f2u(t1) : 0u, h ? (u32)i1 : 0u};
            // (NaN distances: any NaN matches any NaN, payloads are not part of the contract)
            const bool nan0 = h && r[0][0] && (u2f(q[1]) != u2f(q[1])) && (u2f(r[0][1]) != u2f(r[0][1]));
            const bool nan1 = h && r[0][0] && (u2f(q[3]) != u2f(q[3])) && (u2f(r[0][3]) != u2f(r[0][3]));
            if (q[0] != r[0][0] || (!nan0 && q[1] != r[0][1]) || q[2] != r[0][2] || (!nan1 && q[3] != r[0][3]) || q[4] != r[0][4]) ++bad;
        }
        if (r[0][0]) ++hits;
        // (triangle ids are positions in the table's leaf order, which both tables share: they come from one binary tree)
        if (memcmp(r[0], r[1], sizeof r[0]) != 0 || memcmp(r[0], r[2], sizeof r[0]) != 0 || anyq[0] != anyq[1] || anyq[0] != anyq[2] || anyq[0] != (bool)r[0][0]) ++bad;
        if (out) { memcpy(out + (size_t)i * 10, r[0], sizeof r[0]); memcpy(out + (size_t)i * 10 + 5, r[1], sizeof r[1]); }
    }
    if (stats) { stats[0] = hits; stats[1] = with_tbvh; }
    return bad;
}

// leaves of mesh renderer `mesh_index`-th mesh in traversal order, to compare the packer's octree with the oracle's
int emu_octree(const float *tris, uint32_t n_tris, float *leaf_boxes, uint32_t *leaf_counts, uint32_t *ids, uint32_t cap, uint32_t *n_ids)
{
    OctreeFlat oc;
    build_octree(tris, n_tris, oc);
    if (oc.root == NO_NODE) return oc.empty_root ? -2 : -1;
    uint32_t nl = 0, ni = 0;
    // DFS in child order
    std::vector<uint32_t> stack = {oc.root};
    while (!stack.empty()) {
        const uint32_t n = stack.back(); stack.pop_back();
        const float *q = oc.nodes.data() + (size_t)n * NODE_WORDS;
        const uint32_t first = f2u(q[NODE_FIRST]), cnt = f2u(q[NODE_COUNT]);
        if (cnt & 0x80000000u) {
            const uint32_t c = cnt & 0x7fffffffu;
            if (leaf_boxes) { float *b = leaf_boxes + (size_t)nl * 6; b[0] = q[NODE_REL]; b[1] = q[NODE_REL + 1]; b[2] = q[NODE_REL + 2]; b[3] = 2.0f * q[NODE_HALF]; b[4] = 2.0f * q[NODE_HALF + 1]; b[5] = 2.0f * q[NODE_HALF + 2]; }
            if (leaf_counts) leaf_counts[nl] = c;
            for (uint32_t i = 0; i < c; ++i) { if (ids && ni < cap) ids[ni] = oc.leaf_ids[first + i]; ++ni; }
            ++nl;
        } else {
            for (uint32_t i = cnt; i-- > 0;) stack.push_back(first + i);
        }
    }
    if (n_ids) *n_ids = ni;
    return (int)nl;
}

}  // extern "C"
