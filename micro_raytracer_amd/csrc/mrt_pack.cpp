// mrt_pack.cpp — see mrt_pack.h.  Host code only; compiled with -ffp-contract=off so that the
// hoisted f32 values are the ones the reference would recompute per call.
#include "mrt_pack.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <array>

namespace mrt {
namespace {

constexpr u32 kBvhMinInstances = 24;    // below this the uniform linear scan is faster than a divergent tree walk

struct H3 { float x, y, z; };
inline H3 h3(float x, float y, float z) { H3 r = {x, y, z}; return r; }
inline H3 hadd(H3 a, H3 b) { return h3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline H3 hsub(H3 a, H3 b) { return h3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline H3 hmuls(H3 a, float s) { return h3(a.x * s, a.y * s, a.z * s); }
inline H3 hneg(H3 a) { return h3(-a.x, -a.y, -a.z); }
inline H3 hhadam(H3 a, H3 b) { return h3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float hdot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline H3 hcross(H3 a, H3 b) { return h3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float hmag(H3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
inline H3 hnorm(H3 a) { return hmuls(a, 1.0f / hmag(a)); }

// Mat3f::rotate_y(dir), src/lin.rs:175-183 (only dir.w is used)
void rotate_y(float w, float *m)
{
    const float cw = sqrtf(1.0f - w * w);
    const float r[9] = {cw, 0.0f, w, 0.0f, 1.0f, 0.0f, -w, 0.0f, cw};
    memcpy(m, r, sizeof r);
}
// Mat4f::lookat(dir, up = (0,0,1)) upper-left 3x3, src/lin.rs:197-208
void lookat(H3 dxyz, float *m)
{
    const H3 fwd = hnorm(dxyz);
    const H3 right = hnorm(hcross(fwd, h3(0.0f, 0.0f, 1.0f)));
    const H3 n_up = hcross(right, fwd);
    const float r[9] = {right.x, -right.y, right.z, -fwd.x, fwd.y, -fwd.z, n_up.x, -n_up.y, n_up.z};
    memcpy(m, r, sizeof r);
}
H3 mul3(const float *m, H3 v)
{
    return h3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
bool is_identity(const float *m)
{
    static const float id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) if (!(m[i] == id[i])) return false;
    return true;
}

inline u32 bits(float f) { u32 u; memcpy(&u, &f, 4); return u; }
inline float fbits(u32 u) { float f; memcpy(&f, &u, 4); return f; }

struct Blob {
    std::vector<u32> w;
    u32 align4() { while (w.size() & 3u) w.push_back(0); return (u32)w.size(); }
    void f(float v) { w.push_back(bits(v)); }
    void u(u32 v) { w.push_back(v); }
    void f3(H3 v) { f(v.x); f(v.y); f(v.z); }
};

// ---- octree, reference src/rt.rs:630-703 (BVH::gen / construct) and :227-248 (check_in_aabb) ----
struct TNode {
    H3 aabb, rel;
    std::vector<u32> content;
    std::vector<TNode> childs;
};
const float kSign[8][3] = {{1, 1, 1}, {-1, 1, 1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, -1}, {1, -1, -1}};

bool vtx_in(H3 v, H3 hi, H3 lo)
{
    if (v.x > hi.x || v.y > hi.y || v.z > hi.z) return false;
    if (v.x < lo.x || v.y < lo.y || v.z < lo.z) return false;
    return true;
}

void construct(TNode &n, const float *tris, u32 n_tris, u32 d, u32 deep)
{
    if (d >= deep) {
        const H3 hi = hadd(n.rel, hmuls(n.aabb, 0.5f));
        const H3 lo = hsub(n.rel, hmuls(n.aabb, 0.5f));
        for (u32 i = 0; i < n_tris; ++i) {
            const float *t = tris + (size_t)i * 9;
            if (vtx_in(h3(t[0], t[1], t[2]), hi, lo) || vtx_in(h3(t[3], t[4], t[5]), hi, lo) || vtx_in(h3(t[6], t[7], t[8]), hi, lo))
                n.content.push_back(i);
        }
        return;
    }
    for (int i = 0; i < 8; ++i) {
        TNode c;
        c.aabb = hmuls(n.aabb, 0.5f);
        c.rel = hadd(n.rel, hhadam(n.aabb, hmuls(h3(kSign[i][0], kSign[i][1], kSign[i][2]), 0.25f)));
        construct(c, tris, n_tris, d + 1, deep);
        if (!c.content.empty() || !c.childs.empty()) n.childs.push_back(std::move(c));
    }
}

void put_node(std::vector<float> &nodes, u32 idx, const TNode &n, u32 first, u32 count_word)
{
    float *q = nodes.data() + (size_t)idx * NODE_WORDS;
    const H3 half = hmuls(n.aabb, 0.5f);
    q[NODE_HALF] = half.x; q[NODE_HALF + 1] = half.y; q[NODE_HALF + 2] = half.z;
    q[NODE_REL] = n.rel.x; q[NODE_REL + 1] = n.rel.y; q[NODE_REL + 2] = n.rel.z;
    q[NODE_FIRST] = fbits(first);
    q[NODE_COUNT] = fbits(count_word);
}

// children of a node occupy consecutive slots, in the reference's child order
void flatten(const TNode &n, u32 idx, OctreeFlat &out)
{
    if (!n.content.empty()) {
        const u32 first = (u32)out.leaf_ids.size();
        out.leaf_ids.insert(out.leaf_ids.end(), n.content.begin(), n.content.end());
        put_node(out.nodes, idx, n, first, (u32)n.content.size() | 0x80000000u);
        return;
    }
    const u32 first = (u32)(out.nodes.size() / NODE_WORDS);
    out.nodes.resize(out.nodes.size() + n.childs.size() * NODE_WORDS);
    put_node(out.nodes, idx, n, first, (u32)n.childs.size());
    for (size_t i = 0; i < n.childs.size(); ++i) flatten(n.childs[i], first + (u32)i, out);
}

}  // namespace

void build_octree(const float *tris, u32 n_tris, OctreeFlat &out)
{
    out = OctreeFlat();
    if (n_tris == 0) return;                       // Mesh::gen_aabb -> None, src/rt.rs:261-270
    // 2 * max |coordinate| per axis; max_by(total_cmp) over sign-cleared floats == max of the bit patterns
    u32 mx = 0, my = 0, mz = 0;
    for (size_t i = 0; i < (size_t)n_tris * 3; ++i) {
        const u32 ax = bits(fabsf(tris[i * 3])), ay = bits(fabsf(tris[i * 3 + 1])), az = bits(fabsf(tris[i * 3 + 2]));
        if (ax > mx) mx = ax;
        if (ay > my) my = ay;
        if (az > mz) mz = az;
    }
    TNode root;
    root.aabb = h3(2.0f * fbits(mx), 2.0f * fbits(my), 2.0f * fbits(mz));
    root.rel = h3(0.0f, 0.0f, 0.0f);
    construct(root, tris, n_tris, 0, 3);           // BVH::gen(aabb, &mesh, 3), src/parser.rs:816
    if (root.content.empty() && root.childs.empty()) { out.empty_root = true; return; }
    out.nodes.resize(NODE_WORDS);
    out.root = 0;
    flatten(root, 0, out);
}

// ---- triangle BVH of a mesh (mrt_scene.h): a pure accelerator, any valid tree gives the same render ----
// Top-down sweep SAH on triangle centroids (median split above kSahMax triangles per node), leaves of at most
// leaf_max items, nodes in depth-first order with skip links; `order` receives the triangles in leaf order.
namespace {
#ifndef MRT_TBVH_LEAF_MAX          // build-time experiment knobs (make EXTRA=-D...)
#define MRT_TBVH_LEAF_MAX 4
#endif
#ifndef MRT_IBVH_LEAF_MAX
#define MRT_IBVH_LEAF_MAX 2u
#endif
#ifndef MRT_TBVH_CNODE
#define MRT_TBVH_CNODE 1.0          // cost of the two extra box tests of a split, in triangle tests
#endif
constexpr u32 kTbvhLeaf = MRT_TBVH_LEAF_MAX;
constexpr size_t kSahMax = 8192;
constexpr u32 kSahDepth = 48;        // recursion stays shallow whatever the input (callers may run on small thread stacks)
struct TriBox { float mn[3], mx[3], c[3]; };
struct TbvhBuild {
    const std::vector<TriBox> &tb;
    std::vector<float> &nodes;
    std::vector<u32> &order;
    u32 leaf_max;
    static double area(const float *mn, const float *mx)
    {
        const double x = (double)mx[0] - mn[0], y = (double)mx[1] - mn[1], z = (double)mx[2] - mn[2];
        return x * y + y * z + z * x;
    }
    void make(std::vector<u32> &v, size_t lo, size_t hi, u32 depth = 0)
    {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = lo; k < hi; ++k) for (int a = 0; a < 3; ++a) {
            const TriBox &b = tb[v[k]];
            if (b.mn[a] < mn[a]) mn[a] = b.mn[a];
            if (b.mx[a] > mx[a]) mx[a] = b.mx[a];
        }
        const u32 me = (u32)(nodes.size() / BVH_WORDS);
        nodes.resize(nodes.size() + BVH_WORDS);
        const size_t n = hi - lo;
        u32 leaf = 0;
        size_t mid = 0;
        int best_ax = -1;
        if (n > (leaf_max < 2u ? leaf_max : 2u)) {
            if (n > kSahMax || depth > kSahDepth) {      // big nodes, and chains of lopsided SAH splits, fall back to the median
                float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (size_t k = lo; k < hi; ++k) for (int a = 0; a < 3; ++a) { const float c = tb[v[k]].c[a]; if (c < cmn[a]) cmn[a] = c; if (c > cmx[a]) cmx[a] = c; }
                best_ax = 0;
                for (int a = 1; a < 3; ++a) if (cmx[a] - cmn[a] > cmx[best_ax] - cmn[best_ax]) best_ax = a;
                mid = (lo + hi) / 2;
            } else {
                // cost of a split = area(L) * |L| + area(R) * |R| (triangle tests weighted by hit probability)
                double best = INFINITY;
                std::vector<double> right(n);
                for (int a = 0; a < 3; ++a) {
                    std::sort(v.begin() + lo, v.begin() + hi, [&](u32 x, u32 y) { return tb[x].c[a] < tb[y].c[a] || (tb[x].c[a] == tb[y].c[a] && x < y); });
                    float rmn[3] = {INFINITY, INFINITY, INFINITY}, rmx[3] = {-INFINITY, -INFINITY, -INFINITY};
                    for (size_t k = n; k-- > 1;) {
                        const TriBox &b = tb[v[lo + k]];
                        for (int q = 0; q < 3; ++q) { if (b.mn[q] < rmn[q]) rmn[q] = b.mn[q]; if (b.mx[q] > rmx[q]) rmx[q] = b.mx[q]; }
                        right[k] = area(rmn, rmx) * (double)(n - k);
                    }
                    float lmn[3] = {INFINITY, INFINITY, INFINITY}, lmx[3] = {-INFINITY, -INFINITY, -INFINITY};
                    for (size_t k = 1; k < n; ++k) {
                        const TriBox &b = tb[v[lo + k - 1]];
                        for (int q = 0; q < 3; ++q) { if (b.mn[q] < lmn[q]) lmn[q] = b.mn[q]; if (b.mx[q] > lmx[q]) lmx[q] = b.mx[q]; }
                        const double cost = area(lmn, lmx) * (double)k + right[k];
                        if (cost < best) { best = cost; best_ax = a; mid = lo + k; }
                    }
                }
                // a small node stays a leaf when splitting does not pay for the two extra box tests
                if (n <= leaf_max && !(best + MRT_TBVH_CNODE * area(mn, mx) < area(mn, mx) * (double)n)) best_ax = -1;
            }
        }
        if (best_ax < 0) {
            leaf = ((u32)n << 24) | (u32)order.size();
            for (size_t k = lo; k < hi; ++k) order.push_back(v[k]);
        } else {
            if (n > kSahMax || depth > kSahDepth) std::nth_element(v.begin() + lo, v.begin() + mid, v.begin() + hi, [&](u32 x, u32 y) { return tb[x].c[best_ax] < tb[y].c[best_ax] || (tb[x].c[best_ax] == tb[y].c[best_ax] && x < y); });
            else if (best_ax != 2) std::sort(v.begin() + lo, v.begin() + hi, [&](u32 x, u32 y) { return tb[x].c[best_ax] < tb[y].c[best_ax] || (tb[x].c[best_ax] == tb[y].c[best_ax] && x < y); });
            make(v, lo, mid, depth + 1);
            make(v, mid, hi, depth + 1);
        }
        float *q = nodes.data() + (size_t)me * BVH_WORDS;
        for (int a = 0; a < 3; ++a) {
            // centre / half size, rounded so that the stored box contains [mn, mx]
            const float c = 0.5f * mn[a] + 0.5f * mx[a];
            float h = fmaxf(mx[a] - c, c - mn[a]);
            h = nextafterf(h, INFINITY);
            q[BVH_C + a] = c; q[BVH_H + a] = h;
        }
        q[BVH_SKIP] = fbits((u32)(nodes.size() / BVH_WORDS));      // first node after this subtree (mesh-relative)
        q[BVH_LEAF] = fbits(leaf);
    }
};
}  // namespace

bool build_tbvh(const float *tris, u32 n_tris, std::vector<float> &nodes, std::vector<u32> &order)
{
    nodes.clear(); order.clear();
    if (n_tris == 0 || n_tris >= (1u << 24)) return false;
    std::vector<TriBox> tb(n_tris);
    for (u32 t = 0; t < n_tris; ++t) {
        const float *p = tris + (size_t)t * 9;
        TriBox &b = tb[t];
        for (int a = 0; a < 3; ++a) {
            const float x = p[a], y = p[3 + a], z = p[6 + a];
            if (!(fabsf(x) <= 1e6f) || !(fabsf(y) <= 1e6f) || !(fabsf(z) <= 1e6f)) return false;     // also refuses NaN
            b.mn[a] = fminf(x, fminf(y, z)); b.mx[a] = fmaxf(x, fmaxf(y, z));
            b.c[a] = (float)(((double)x + y + z) / 3.0);
        }
    }
    std::vector<u32> v(n_tris);
    for (u32 t = 0; t < n_tris; ++t) v[t] = t;
    TbvhBuild build{tb, nodes, order, kTbvhLeaf};
    build.make(v, 0, n_tris);
    return true;
}

// ---- 4-wide collapse of the binary triangle BVH (mrt_scene.h) ----
// `bin` is build_tbvh's depth-first threaded table (mesh-relative skip links: children of an internal node n are n + 1 and
// skip(n + 1)).  A 4-wide node takes the two children of a binary node and keeps opening the internal slot with the largest
// box until four slots are full (or only leaves remain).  Returns the nodes in creation order (node 0 = root) with child
// words whose internal form holds the LOCAL node index; depth[] is each node's level.
namespace {
struct Wide4 {
    float c[4][3], h[4][3];
    u32 child[4];              // 0 empty | leaf word | B4_INTERNAL | local index
    u32 depth;
};
struct Collapse4 {
    const std::vector<float> &bin;
    std::vector<Wide4> out;
    u32 skip_of(u32 n) const { return bits(bin[(size_t)n * BVH_WORDS + BVH_SKIP]); }
    u32 leaf_of(u32 n) const { return bits(bin[(size_t)n * BVH_WORDS + BVH_LEAF]); }
    double area_of(u32 n) const
    {
        const float *q = bin.data() + (size_t)n * BVH_WORDS;
        const double x = q[BVH_H], y = q[BVH_H + 1], z = q[BVH_H + 2];
        return x * y + y * z + z * x;
    }
    u32 make(u32 bnode, u32 depth)
    {
        const u32 me = (u32)out.size();
        out.emplace_back();
        std::vector<u32> slots;
        if (leaf_of(bnode) != 0u) slots.push_back(bnode);                       // a one-leaf mesh: the root holds it
        else { slots.push_back(bnode + 1u); slots.push_back(skip_of(bnode + 1u)); }
        while (slots.size() < 4u) {
            int pick = -1;
            double best = -1.0;
            for (size_t k = 0; k < slots.size(); ++k) if (leaf_of(slots[k]) == 0u && area_of(slots[k]) > best) { best = area_of(slots[k]); pick = (int)k; }
            if (pick < 0) break;
            const u32 n = slots[pick];
            slots[pick] = n + 1u;                                               // the opened slot keeps its place, its sibling follows
            slots.insert(slots.begin() + pick + 1, skip_of(n + 1u));
        }
        // internal children first (in their depth-first order), leaves behind them: the internal children of a node are then
        // consecutive nodes of the level-ordered table, child k at (first child) + k -- what lets ONE stack entry of the
        // walk stand for all pending siblings (mrt_trace.h mesh_isect)
        std::stable_partition(slots.begin(), slots.end(), [&](u32 n) { return leaf_of(n) == 0u; });
        Wide4 w;
        memset(&w, 0, sizeof w);
        w.depth = depth;
        for (size_t k = 0; k < slots.size(); ++k) {
            const float *q = bin.data() + (size_t)slots[k] * BVH_WORDS;
            for (int a = 0; a < 3; ++a) { w.c[k][a] = q[BVH_C + a]; w.h[k][a] = q[BVH_H + a]; }
            w.child[k] = leaf_of(slots[k]) != 0u ? leaf_of(slots[k]) : (B4_INTERNAL | make(slots[k], depth + 1u));
        }
        out[me] = w;
        return me;
    }
};
}  // namespace

// image 0.24 imageops::sample: per-output-index taps of horizontal_sample / vertical_sample
void lanczos3_taps(u32 src, u32 dst, ResampleTaps &out)
{
    const float ratio = (float)src / (float)dst;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float support = 3.0f * sratio;
    out.cap = (u32)(2.0f * support) + 4;
    out.left.assign(dst, 0);
    out.count.assign(dst, 0);
    out.weight.assign((size_t)dst * out.cap, 0.0f);
    auto sinc = [](float t) { const float a = t * kPi; return (t == 0.0f) ? 1.0f : sinf(a) / a; };
    auto kernel = [&](float x) { return (fabsf(x) < 3.0f) ? sinc(x) * sinc(x / 3.0f) : 0.0f; };
    for (u32 o = 0; o < dst; ++o) {
        float input = ((float)o + 0.5f) * ratio;
        long long left = (long long)floorf(input - support);
        if (left < 0) left = 0;
        if (left > (long long)src - 1) left = (long long)src - 1;
        long long right = (long long)ceilf(input + support);
        if (right < left + 1) right = left + 1;
        if (right > (long long)src) right = (long long)src;
        input = input - 0.5f;
        u32 n = (u32)(right - left);
        if (n > out.cap) n = out.cap;               // cannot happen: cap >= 2*support + 2
        float *w = out.weight.data() + (size_t)o * out.cap;
        float sum = 0.0f;
        for (u32 i = 0; i < n; ++i) { w[i] = kernel(((float)(left + i) - input) / sratio); sum += w[i]; }
        for (u32 i = 0; i < n; ++i) w[i] /= sum;
        out.left[o] = (u32)left;
        out.count[o] = n;
    }
}

namespace {

inline float min_num(float a, float b) { if (a != a) return b; if (b != b) return a; return b < a ? b : a; }
inline bool unit(float v) { return v >= 0.0f && v <= 1.0f; }

u32 to_usize_u32(float v)   // (res as f32 * ssaa) as usize, src/sampler.rs:29-30
{
    if (!(v > 0.0f)) return 0;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (u32)v;
}

}  // namespace

int pack_scene(const mrt_render_desc *d, Packed &out, std::string &err, const PackOpts &opts)
{
    char msg[256];
    if (!d) { err = "null render description"; return MRT_ERR_ARG; }
    const mrt_scene &sc = d->scene;
    if ((sc.n_renderer && !sc.renderer) || (sc.n_light && !sc.light) || (sc.n_textures && !sc.textures)) {
        err = "null array with non-zero count"; return MRT_ERR_ARG;
    }
    out = Packed();
    Params &P = out.P;
    memset(&P, 0, sizeof P);

    // ---- frame ----
    const mrt_frame &fr = d->frame;
    out.res_w = fr.res_w; out.res_h = fr.res_h;
    out.gamma = fr.cam.gamma; out.exp = fr.cam.exp;
    P.w = (float)fr.res_w * fr.ssaa;
    P.h = (float)fr.res_h * fr.ssaa;
    out.nw = to_usize_u32(P.w);
    out.nh = to_usize_u32(P.h);
    if (out.nw == 0 || out.nh == 0) { err = "empty frame (res * ssaa truncates to 0)"; return MRT_ERR_SCENE; }
    if ((unsigned long long)out.nw * out.nh > (1ull << 30)) { err = "frame has more than 2^30 supersampled pixels"; return MRT_ERR_LIMIT; }
    P.nw = out.nw; P.nh = out.nh;
    P.aspect = P.w / P.h;
    const float tan_fov = tanf((0.5f * fr.cam.fov) * (kPi / 180.0f));    // f32::to_radians().tan(), src/rt.rs:902
    P.inv2tan = 1.0f / (2.0f * tan_fov);
    for (int k = 0; k < 3; ++k) P.cam_pos[k] = fr.cam.pos[k];
    P.aprt = fr.cam.aprt; P.foc = fr.cam.foc;
    lookat(h3(fr.cam.dir[1], fr.cam.dir[2], fr.cam.dir[3]), P.cam_L);
    rotate_y(fr.cam.dir[0], P.cam_R);
    P.cam_ident = (is_identity(P.cam_L) && is_identity(P.cam_R)) ? 1u : 0u;
    P.bounce = d->rt.bounce;
    if (P.bounce > 0x0fffffffu) { err = "bounce too large"; return MRT_ERR_LIMIT; }
    P.q = 1.0f - min_num(d->rt.loss, 1.0f);
    for (int k = 0; k < 3; ++k) { P.sky[k] = sc.sky.color[k]; P.sky_init[k] = sc.sky.color[k] * sc.sky.pwr; }

    // ---- validation: everything the reference would panic on is refused here ----
    for (u32 t = 0; t < sc.n_textures; ++t) {
        const mrt_texture &tx = sc.textures[t];
        if (tx.dat && ((unsigned long long)tx.w * tx.h == 0 || (unsigned long long)tx.w * tx.h > 0x7fffffffull)) {
            snprintf(msg, sizeof msg, "texture %u: %ux%u texels (reference would index out of bounds, src/rt.rs:624)", t, tx.w, tx.h);
            err = msg; return MRT_ERR_SCENE;
        }
    }
    for (u32 r = 0; r < sc.n_renderer; ++r) {
        const mrt_renderer &o = sc.renderer[r];
        if (o.kind > MRT_KIND_MESH) { snprintf(msg, sizeof msg, "renderer %u: unknown kind %u", r, o.kind); err = msg; return MRT_ERR_SCENE; }
        if (o.n_inst && !o.inst) { err = "null instance array"; return MRT_ERR_ARG; }
        if (o.kind == MRT_KIND_MESH && o.n_tris && !o.tris) { err = "null triangle array"; return MRT_ERR_ARG; }
        const int32_t maps[6] = {o.mat.tex, o.mat.rmap, o.mat.mmap, o.mat.gmap, o.mat.omap, o.mat.emap};
        for (int k = 0; k < 6; ++k) {
            if (maps[k] >= (int32_t)sc.n_textures) { snprintf(msg, sizeof msg, "renderer %u: map %d index out of range", r, k); err = msg; return MRT_ERR_SCENE; }
            if (maps[k] >= 0 && (o.kind == MRT_KIND_TRIANGLE || o.kind == MRT_KIND_MESH)) {
                snprintf(msg, sizeof msg, "renderer %u: texture maps on a triangle/mesh hit todo!() in the reference (src/rt.rs:546,806)", r);
                err = msg; return MRT_ERR_SCENE;
            }
        }
        if (o.mat.emap < 0 && !unit(o.mat.emit)) {
            snprintf(msg, sizeof msg, "renderer %u: emit %g outside [0,1] (gen_bool panics, src/rt.rs:968)", r, (double)o.mat.emit); err = msg; return MRT_ERR_SCENE;
        }
        if (o.mat.omap < 0 && min_num(1.0f - o.mat.opacity, 0.85f) < 0.0f) {
            snprintf(msg, sizeof msg, "renderer %u: opacity %g > 1 (gen_bool panics, src/rt.rs:1054)", r, (double)o.mat.opacity); err = msg; return MRT_ERR_SCENE;
        }
        if (o.mat.emap >= 0 && sc.textures[o.mat.emap].dat) {
            const mrt_texture &tx = sc.textures[o.mat.emap];
            for (size_t i = 0; i < (size_t)tx.w * tx.h; ++i) if (!unit(tx.dat[i * 3])) { err = "emap texel outside [0,1] (gen_bool panics, src/rt.rs:968)"; return MRT_ERR_SCENE; }
        }
        if (o.mat.omap >= 0 && sc.textures[o.mat.omap].dat) {
            const mrt_texture &tx = sc.textures[o.mat.omap];
            for (size_t i = 0; i < (size_t)tx.w * tx.h; ++i) if (min_num(1.0f - tx.dat[i * 3], 0.85f) < 0.0f) { err = "omap texel > 1 (gen_bool panics, src/rt.rs:1054)"; return MRT_ERR_SCENE; }
        }
    }

    // ---- tables ----
    Blob B;
    std::map<std::array<u32, 4>, u32> xf_index;
    std::vector<float> xf_tab;
    auto xf_of = [&](const float *dir) -> u32 {
        std::array<u32, 4> key = {bits(dir[0]), bits(dir[1]), bits(dir[2]), bits(dir[3])};
        auto it = xf_index.find(key);
        if (it != xf_index.end()) return it->second;
        float rec[XF_WORDS] = {0};
        // objects use -inst.dir, src/rt.rs:726-727
        lookat(h3(-dir[1], -dir[2], -dir[3]), rec + XF_L);
        rotate_y(-dir[0], rec + XF_R);
        rec[XF_IDENT] = fbits((is_identity(rec + XF_L) && is_identity(rec + XF_R)) ? 1u : 0u);
        const u32 id = (u32)(xf_tab.size() / XF_WORDS);
        xf_tab.insert(xf_tab.end(), rec, rec + XF_WORDS);
        xf_index[key] = id;
        return id;
    };

    std::vector<u32> rend_tab, inst_tab, instx_tab, mat_tab, mesh_tab, leaf_tab;
    struct Bound { float mn[3], mx[3]; bool ok; };
    std::vector<Bound> bounds;           // world-space box of every flat instance (ok = false: cannot be bounded)
    std::vector<float> tri_tab, node_tab, tbvh_tab;
    std::vector<std::vector<Wide4>> wide;       // per mesh with a triangle BVH: its 4-wide nodes (local child indices)
    std::vector<u32> wide_mesh;                 // ... and the mesh-table record it belongs to
    std::vector<u32> memb_tab, membe_tab, parent_tab;
    u32 n_inst_total = 0;
    for (u32 r = 0; r < sc.n_renderer; ++r) {
        const mrt_renderer &o = sc.renderer[r];
        u32 rec[REND_WORDS] = {0};
        rec[REND_KIND] = o.kind;
        rec[REND_INST_OFF] = n_inst_total;
        rec[REND_INST_CNT] = o.n_inst;
        const int32_t maps[6] = {o.mat.tex, o.mat.rmap, o.mat.mmap, o.mat.gmap, o.mat.omap, o.mat.emap};
        bool any_map = false;
        for (int k = 0; k < 6; ++k) any_map |= maps[k] >= 0;
        if (any_map) out.features |= 4u;                                             // F_MAPS
        if (o.kind == MRT_KIND_BOX || o.kind == MRT_KIND_MESH) out.features |= 1u;   // F_BOX
        if (o.kind == MRT_KIND_TRIANGLE || o.kind == MRT_KIND_MESH) out.features |= 2u;   // F_TRI
        rec[REND_FLAGS] = any_map ? RF_HAS_MAPS : 0u;
        H3 nn = h3(0, 0, 0), nraw = h3(0, 0, 0);
        if (o.kind == MRT_KIND_SPHERE) {
            rec[REND_GEO] = bits(o.param[0] * o.param[0]);
        } else if (o.kind == MRT_KIND_PLANE) {
            nraw = h3(o.param[0], o.param[1], o.param[2]);
            nn = hnorm(nraw);
            rec[REND_GEO] = bits(nn.x); rec[REND_GEO + 1] = bits(nn.y); rec[REND_GEO + 2] = bits(nn.z);
            rec[REND_GEO + 3] = bits(nraw.x); rec[REND_GEO + 4] = bits(nraw.y); rec[REND_GEO + 5] = bits(nraw.z);
        } else if (o.kind == MRT_KIND_BOX) {
            const H3 sz = h3(o.param[0], o.param[1], o.param[2]);
            const H3 half = hmuls(sz, 0.5f);
            const H3 inv2 = hmuls(h3(1.0f / sz.x, 1.0f / sz.y, 1.0f / sz.z), 2.0f);
            rec[REND_GEO] = bits(half.x); rec[REND_GEO + 1] = bits(half.y); rec[REND_GEO + 2] = bits(half.z);
            rec[REND_GEO + 3] = bits(inv2.x); rec[REND_GEO + 4] = bits(inv2.y); rec[REND_GEO + 5] = bits(inv2.z);
        } else if (o.kind == MRT_KIND_TRIANGLE) {
            const H3 a = h3(o.param[0], o.param[1], o.param[2]), b = h3(o.param[3], o.param[4], o.param[5]), c = h3(o.param[6], o.param[7], o.param[8]);
            const H3 e0 = hsub(b, a), e1 = hsub(c, a);
            const float g[9] = {a.x, a.y, a.z, e0.x, e0.y, e0.z, e1.x, e1.y, e1.z};
            for (int k = 0; k < 9; ++k) rec[REND_GEO + k] = bits(g[k]);
        } else {
            const u32 mesh_id = (u32)(mesh_tab.size() / MESH_WORDS);
            rec[REND_GEO] = mesh_id;
            OctreeFlat oc;
            build_octree(o.tris, o.n_tris, oc);
            if (oc.empty_root) {
                snprintf(msg, sizeof msg, "renderer %u: mesh octree is empty (reference unwrap() panics, src/rt.rs:717)", r);
                err = msg; return MRT_ERR_SCENE;
            }
            const u32 node0 = (u32)(node_tab.size() / NODE_WORDS);
            const u32 leaf0 = (u32)leaf_tab.size();
            const u32 tri0 = (u32)(tri_tab.size() / TRI_WORDS);
            const u32 n_oc_nodes = (u32)(oc.nodes.size() / NODE_WORDS);
            // triangle BVH: triangles are stored in its leaf order (new id -> old id in `order`)
            std::vector<float> tbn;
            std::vector<u32> order, new_of(o.n_tris);
            bool tb_ok = build_tbvh(o.tris, o.n_tris, tbn, order);
            tb_ok = tb_ok && oc.root != NO_NODE && n_oc_nodes <= (1u << (32 - MEMB_SLOT_BITS)) && oc.leaf_ids.size() <= MEMB_SLOT_MASK;
            if (!tb_ok) { order.resize(o.n_tris); for (u32 t = 0; t < o.n_tris; ++t) order[t] = t; tbn.clear(); }
            for (u32 t = 0; t < o.n_tris; ++t) new_of[order[t]] = t;
            // membership of every triangle: (octree leaf, slot) of each occurrence in the leaf lists; parents of the nodes
            std::vector<std::vector<u32>> memb(tb_ok ? o.n_tris : 0);
            const size_t parent0 = parent_tab.size();
            parent_tab.resize(parent0 + n_oc_nodes, NO_NODE);
            for (u32 n = 0; n < n_oc_nodes; ++n) {
                float *q = oc.nodes.data() + (size_t)n * NODE_WORDS;
                const u32 first = bits(q[NODE_FIRST]), cw = bits(q[NODE_COUNT]);
                if (cw & 0x80000000u) {
                    if (tb_ok) for (u32 k = 0; k < (cw & 0x7fffffffu); ++k) memb[new_of[oc.leaf_ids[first + k]]].push_back((n << MEMB_SLOT_BITS) | (first + k));
                } else {
                    for (u32 k = 0; k < cw; ++k) parent_tab[parent0 + first + k] = node0 + n;
                    q[NODE_FIRST] = fbits(first + node0);      // node-relative child indices -> absolute node indices
                }
            }
            for (u32 t = 0; tb_ok && t < o.n_tris; ++t) if (memb[t].size() > 255u) tb_ok = false;
            if (tb_ok && membe_tab.size() + oc.leaf_ids.size() >= (1u << 24)) tb_ok = false;
            float mesh_c[3] = {0, 0, 0}, mesh_h[3] = {0, 0, 0};
            u32 mesh_tb = NO_NODE;
            if (tb_ok) {
                for (int a = 0; a < 3; ++a) { mesh_c[a] = tbn[BVH_C + a]; mesh_h[a] = tbn[BVH_H + a]; }      // bounds of the mesh = the root's box
                if (opts.tbvh_wide) {
                    // 4-wide collapse (emitted below, the nodes of all meshes in level order)
                    Collapse4 col{tbn, {}};
                    col.make(0u, 0u);
                    wide_mesh.push_back((u32)(mesh_tab.size() / MESH_WORDS));
                    wide.push_back(std::move(col.out));
                } else {
                    // binary threaded table: mesh-relative skip links -> absolute
                    mesh_tb = (u32)(tbvh_tab.size() / BVH_WORDS);
                    const u32 nn = (u32)(tbn.size() / BVH_WORDS);
                    for (u32 k = 0; k < nn; ++k) {
                        float *q = tbn.data() + (size_t)k * BVH_WORDS;
                        const u32 skip = bits(q[BVH_SKIP]);
                        q[BVH_SKIP] = fbits(skip >= nn ? BVH_END : mesh_tb + skip);
                    }
                    tbvh_tab.insert(tbvh_tab.end(), tbn.begin(), tbn.end());
                    // a sentinel behind the tree: the walks go to node + 1 on every hit (a leaf's skip link is its successor in
                    // depth-first order), which for the LAST leaf of the tree is this node -- a box no ray hits (negative half
                    // sizes: the near plane lies behind the far plane on every axis), then the end
                    float sent[BVH_WORDS] = {0.0f, 0.0f, 0.0f, -1e30f, -1e30f, -1e30f, fbits(BVH_END), fbits(0u)};
                    static_assert(BVH_C == 0 && BVH_H == 3 && BVH_SKIP == 6 && BVH_LEAF == 7 && BVH_WORDS == 8, "sentinel layout");
                    tbvh_tab.insert(tbvh_tab.end(), sent, sent + BVH_WORDS);
                }
            }
            for (u32 t = 0; t < o.n_tris; ++t) {
                u32 head = 0;
                if (tb_ok) {
                    std::sort(memb[t].begin(), memb[t].end(), [](u32 a, u32 b) { return (a & MEMB_SLOT_MASK) < (b & MEMB_SLOT_MASK); });
                    head = ((u32)memb[t].size() << 24) | (u32)membe_tab.size();
                    membe_tab.insert(membe_tab.end(), memb[t].begin(), memb[t].end());
                }
                memb_tab.push_back(head);
            }
            mesh_tab.push_back(tri0);
            mesh_tab.push_back(o.n_tris);
            mesh_tab.push_back(oc.root == NO_NODE ? NO_NODE : node0 + oc.root);
            mesh_tab.push_back(leaf0);
            mesh_tab.push_back(mesh_tb);                       // MESH_TBVH (wide table: set when its nodes are laid out)
            mesh_tab.push_back((u32)oc.leaf_ids.size());       // MESH_NIDS
            for (int a = 0; a < 3; ++a) mesh_tab.push_back(bits(mesh_c[a]));
            for (int a = 0; a < 3; ++a) mesh_tab.push_back(bits(mesh_h[a]));
            node_tab.insert(node_tab.end(), oc.nodes.begin(), oc.nodes.end());
            for (u32 id : oc.leaf_ids) leaf_tab.push_back(new_of[id]);
            for (u32 t = 0; t < o.n_tris; ++t) {
                const float *p = o.tris + (size_t)order[t] * 9;
                const H3 a = h3(p[0], p[1], p[2]), b = h3(p[3], p[4], p[5]), c = h3(p[6], p[7], p[8]);
                const H3 e0 = hsub(b, a), e1 = hsub(c, a);
                const float g[9] = {a.x, a.y, a.z, e0.x, e0.y, e0.z, e1.x, e1.y, e1.z};
                tri_tab.insert(tri_tab.end(), g, g + 9);
            }
        }
        rend_tab.insert(rend_tab.end(), rec, rec + REND_WORDS);

        for (u32 i = 0; i < o.n_inst; ++i) {
            const mrt_instance &in = o.inst[i];
            u32 ir[INST_WORDS] = {0};
            u32 ix[INSTX_WORDS] = {0};
            const H3 pos = h3(in.pos[0], in.pos[1], in.pos[2]);
            ir[INST_POS] = bits(pos.x); ir[INST_POS + 1] = bits(pos.y); ir[INST_POS + 2] = bits(pos.z);
            const u32 xf = xf_of(in.dir);
            if ((unsigned long long)xf * XF_WORDS >= (1ull << 28)) { err = "too many distinct instance directions"; return MRT_ERR_LIMIT; }
            ir[INST_TAG] = o.kind | (bits(xf_tab[(size_t)xf * XF_WORDS + XF_IDENT]) ? TAG_IDENT : 0u) | ((xf * XF_WORDS) << TAG_XF_SHIFT);      // word offset of the transform
            ix[INSTX_REND] = r;
            if (o.kind == MRT_KIND_SPHERE) {
                ir[INST_P3] = rec[REND_GEO];                                                    // r * r
            } else if (o.kind == MRT_KIND_PLANE) {
                ir[INST_P3] = bits(hdot(hneg(nn), pos));                                        // src/rt.rs:404
                ir[INST_P5] = bits(nn.x); ir[INST_P5 + 1] = bits(nn.y); ir[INST_P5 + 2] = bits(nn.z);
                const float *X = xf_tab.data() + (size_t)xf * XF_WORDS;
                const H3 nw = hnorm(mul3(X + XF_R, mul3(X + XF_L, nraw)));                      // src/rt.rs:786,792
                ix[INSTX_PLANE_NW] = bits(nw.x); ix[INSTX_PLANE_NW + 1] = bits(nw.y); ix[INSTX_PLANE_NW + 2] = bits(nw.z);
            } else if (o.kind == MRT_KIND_BOX) {
                ir[INST_P3] = rec[REND_GEO]; ir[INST_P5] = rec[REND_GEO + 1]; ir[INST_P5 + 1] = rec[REND_GEO + 2];   // 0.5 * sizes
            }
            inst_tab.insert(inst_tab.end(), ir, ir + INST_WORDS);
            instx_tab.insert(instx_tab.end(), ix, ix + INSTX_WORDS);
            // world-space bounds: the object-space box moved to pos when the instance transform is the identity as values,
            // else the cube around the bounding sphere (centre pos, radius = largest object-space extent), which is valid
            // when the transform preserves lengths (it is a rotation unless the direction is degenerate)
            Bound bd;
            bd.ok = o.kind != MRT_KIND_PLANE;
            for (int a = 0; a < 3; ++a) bd.mn[a] = bd.mx[a] = 0.0f;
            if (bd.ok) {
                double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, rad = 0.0;
                if (o.kind == MRT_KIND_SPHERE) { rad = fabs((double)o.param[0]); for (int a = 0; a < 3; ++a) { lo[a] = -rad; hi[a] = rad; } }
                else if (o.kind == MRT_KIND_BOX) {
                    rad = 0.5 * sqrt((double)o.param[0] * o.param[0] + (double)o.param[1] * o.param[1] + (double)o.param[2] * o.param[2]);
                    for (int a = 0; a < 3; ++a) { hi[a] = 0.5 * fabs((double)o.param[a]); lo[a] = -hi[a]; }
                } else {
                    const float *vs = o.kind == MRT_KIND_TRIANGLE ? o.param : o.tris;
                    const size_t nv = o.kind == MRT_KIND_TRIANGLE ? 3 : (size_t)o.n_tris * 3;
                    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
                    for (size_t v = 0; v < nv; ++v) {
                        const float *q = vs + v * 3;
                        const double m = sqrt((double)q[0] * q[0] + (double)q[1] * q[1] + (double)q[2] * q[2]);
                        if (!(m <= rad)) rad = m;
                        for (int a = 0; a < 3; ++a) { if (!(q[a] >= lo[a])) lo[a] = q[a]; if (!(q[a] <= hi[a])) hi[a] = q[a]; }
                    }
                    if (nv == 0) for (int a = 0; a < 3; ++a) lo[a] = hi[a] = 0.0;
                }
                const float *X = xf_tab.data() + (size_t)xf * XF_WORDS;
                const bool ident = bits(X[XF_IDENT]) != 0;
                bool ortho = true;
                if (!ident) {
                    double M[9];
                    for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) { double acc = 0; for (int k = 0; k < 3; ++k) acc += (double)X[XF_R + a * 3 + k] * X[XF_L + k * 3 + b2]; M[a * 3 + b2] = acc; }
                    for (int a = 0; a < 3 && ortho; ++a) for (int b2 = 0; b2 < 3; ++b2) { double acc = 0; for (int k = 0; k < 3; ++k) acc += M[a * 3 + k] * M[b2 * 3 + k]; if (!(fabs(acc - (a == b2 ? 1.0 : 0.0)) < 1e-4)) ortho = false; }
                    for (int a = 0; a < 3; ++a) { lo[a] = -rad; hi[a] = rad; }
                }
                const double pp[3] = {pos.x, pos.y, pos.z};
                bd.ok = ortho && rad < 1e6;
                for (int a = 0; a < 3 && bd.ok; ++a) {
                    const double slack = 1e-3 * fmax(fabs(lo[a]), fabs(hi[a])) + 1e-6;
                    bd.mn[a] = (float)(pp[a] + lo[a] - slack); bd.mx[a] = (float)(pp[a] + hi[a] + slack);
                    if (!(fabs(pp[a]) < 1e6) || !(bd.mn[a] <= bd.mx[a])) bd.ok = false;     // also refuses NaN
                }
            }
            bounds.push_back(bd);
        }
        n_inst_total += o.n_inst;

        u32 mr[MAT_WORDS] = {0};
        mr[MAT_ALBEDO] = bits(o.mat.albedo[0]); mr[MAT_ALBEDO + 1] = bits(o.mat.albedo[1]); mr[MAT_ALBEDO + 2] = bits(o.mat.albedo[2]);
        mr[MAT_ROUGH] = bits(o.mat.rough); mr[MAT_METAL] = bits(o.mat.metal); mr[MAT_GLASS] = bits(o.mat.glass);
        mr[MAT_OPACITY] = bits(o.mat.opacity); mr[MAT_EMIT] = bits(o.mat.emit);
        for (int k = 0; k < 6; ++k) mr[MAT_MAP + k] = (u32)maps[k];
        mat_tab.insert(mat_tab.end(), mr, mr + MAT_WORDS);
    }
    if (xf_tab.empty()) { const float dflt[4] = {-0.0f, -0.0f, -1.0f, -0.0f}; xf_of(dflt); }

    P.n_rend = sc.n_renderer; P.n_inst = n_inst_total; P.n_light = sc.n_light;
    if (sc.n_light) out.features |= 8u;                                              // F_LIGHTS
    P.off_rend = B.align4(); B.w.insert(B.w.end(), rend_tab.begin(), rend_tab.end());
    // instance BVH: only worth it for many instances; a pure speed-up, the hit it returns is the linear scan's
    std::vector<u32> lin_list, bvh_inst;
    std::vector<float> bvh_nodes;
    {
        std::vector<u32> elig;
        for (u32 i = 0; i < n_inst_total; ++i) (bounds[i].ok ? elig : lin_list).push_back(i);
        if (n_inst_total < kBvhMinInstances || elig.size() < kBvhMinInstances / 2) {
            lin_list.clear(); elig.clear();
        } else {
            std::vector<TriBox> ib(n_inst_total);
            for (u32 i : elig) for (int a = 0; a < 3; ++a) { ib[i].mn[a] = bounds[i].mn[a]; ib[i].mx[a] = bounds[i].mx[a]; ib[i].c[a] = 0.5f * bounds[i].mn[a] + 0.5f * bounds[i].mx[a]; }
            TbvhBuild build{ib, bvh_nodes, bvh_inst, MRT_IBVH_LEAF_MAX};
            build.make(elig, 0, elig.size());
            const u32 n_nodes = (u32)(bvh_nodes.size() / BVH_WORDS);
            for (u32 k = 0; k < n_nodes; ++k) { float *q = bvh_nodes.data() + (size_t)k * BVH_WORDS; if (bits(q[BVH_SKIP]) >= n_nodes) q[BVH_SKIP] = fbits(BVH_END); }
            out.features |= 16u;                                                         // F_BVH
        }
    }
    P.n_lin = (u32)lin_list.size(); P.n_bvh_nodes = (u32)(bvh_nodes.size() / BVH_WORDS);
    out.all_ident = n_inst_total > 0;
    for (u32 i = 0; i < n_inst_total; ++i) out.all_ident = out.all_ident && (inst_tab[(size_t)i * INST_WORDS + INST_TAG] & TAG_IDENT) != 0u;
    {
        // the culling margin of the instance BVH (mrt_trace.h), one per ray from the root box: 1e-4 of the origin distance (boxes,
        // triangles, mesh root boxes: rounding proportional to the distance) + 4e-6 / r_min of its SQUARE when spheres are bounded
        // (Sphere::intersect's discriminant b*b - 4ac cancels: a ray passing eps |oo|^2 / r outside a sphere can answer "hit")
        float r_min = 3.0e38f;
        bool sphere = false;
        for (u32 i : bvh_inst) {
            if ((inst_tab[(size_t)i * INST_WORDS + INST_TAG] & TAG_KIND_MASK) != KIND_SPHERE) continue;
            sphere = true;
            for (int a = 0; a < 3; ++a) r_min = std::min(r_min, 0.5f * (bounds[i].mx[a] - bounds[i].mn[a]));
        }
        P.inst_k = 1e-4f;
        P.inst_ksq = !sphere ? 0.0f : (r_min > 1e-12f ? 4e-6f / r_min : 1e30f);
        P.inst_kpos = sphere ? 1e-5f : 2e-6f;
    }
    out.n_lin = P.n_lin; out.n_bvh_nodes = P.n_bvh_nodes;
    P.off_cam = B.align4(); for (int k = 0; k < 9; ++k) B.f(P.cam_L[k]); for (int k = 0; k < 9; ++k) B.f(P.cam_R[k]);      // read by the kernel from here (cold path)
    P.off_lin = B.align4(); B.w.insert(B.w.end(), lin_list.begin(), lin_list.end());
    P.off_bvh = B.align4(); for (float v : bvh_nodes) B.f(v);
    P.off_bvhinst = B.align4(); B.w.insert(B.w.end(), bvh_inst.begin(), bvh_inst.end());
    P.off_inst = B.align4(); B.w.insert(B.w.end(), inst_tab.begin(), inst_tab.end());
    P.off_instx = B.align4(); B.w.insert(B.w.end(), instx_tab.begin(), instx_tab.end());
    P.off_xf = B.align4(); for (float v : xf_tab) B.f(v);
    P.off_mat = B.align4(); B.w.insert(B.w.end(), mat_tab.begin(), mat_tab.end());
    P.off_light = B.align4();
    for (u32 l = 0; l < sc.n_light; ++l) {
        const mrt_light &li = sc.light[l];
        if (li.kind > MRT_LIGHT_DIR) { err = "unknown light kind"; return MRT_ERR_SCENE; }
        B.u(li.kind);
        if (li.kind == MRT_LIGHT_POINT) B.f3(h3(li.v[0], li.v[1], li.v[2]));
        else B.f3(hnorm(hneg(hnorm(h3(li.v[0], li.v[1], li.v[2])))));      // (-dir.norm()).norm(), src/rt.rs:1031-1034
        B.f(li.pwr);
        B.f3(h3(li.color[0], li.color[1], li.color[2]));
    }
    // texture descriptors and the k/255 LUT are hot; the texels themselves are cold (one lookup per shaded hit) and go behind
    // the node arrays, next to the triangles
    P.off_tex = B.align4();
    const u32 tex_desc0 = (u32)B.w.size();
    B.w.resize(B.w.size() + (size_t)sc.n_textures * TEX_WORDS, 0);
    P.off_lut = B.align4();
    for (int k = 0; k < 256; ++k) B.f((float)k / 255.0f);
    if (opts.tbvh_wide) {
        // the 4-wide nodes of all meshes in level order (roots first): a prefix of the table is the top of every tree
        struct Ref { u32 depth, mesh, local; };
        std::vector<Ref> order;
        std::vector<std::vector<u32>> newi(wide.size());
        for (size_t m = 0; m < wide.size(); ++m) { newi[m].resize(wide[m].size()); for (size_t k = 0; k < wide[m].size(); ++k) order.push_back({wide[m][k].depth, (u32)m, (u32)k}); }
        std::stable_sort(order.begin(), order.end(), [](const Ref &a, const Ref &b) { return a.depth < b.depth; });     // creation order within a level
        if (order.size() >= (1u << 24)) { err = "triangle BVHs too large"; return MRT_ERR_LIMIT; }      // node index << 4 | mask in a stack entry
        for (size_t k = 0; k < order.size(); ++k) newi[order[k].mesh][order[k].local] = (u32)k;
        tbvh_tab.assign(order.size() * B4_WORDS, 0.0f);
        for (size_t k = 0; k < order.size(); ++k) {
            const Wide4 &w = wide[order[k].mesh][order[k].local];
            float *q = tbvh_tab.data() + k * B4_WORDS;
            for (int c = 0; c < 4; ++c) {
                q[B4_CX + c] = w.c[c][0]; q[B4_CY + c] = w.c[c][1]; q[B4_CZ + c] = w.c[c][2];
                q[B4_HX + c] = w.h[c][0]; q[B4_HY + c] = w.h[c][1]; q[B4_HZ + c] = w.h[c][2];
                u32 cw = w.child[c];
                if (cw & B4_INTERNAL) cw = B4_INTERNAL | newi[order[k].mesh][cw & ~B4_INTERNAL];
                q[B4_CHILD + c] = fbits(cw);
            }
        }
        for (size_t m = 0; m < wide.size(); ++m) mesh_tab[(size_t)wide_mesh[m] * MESH_WORDS + MESH_TBVH] = newi[m][0];
        out.tbvh_wide = true;
    }
    P.off_mesh = B.align4(); B.w.insert(B.w.end(), mesh_tab.begin(), mesh_tab.end());
    P.off_node = B.align4(); for (float v : node_tab) B.f(v);
    P.off_parent = B.align4(); B.w.insert(B.w.end(), parent_tab.begin(), parent_tab.end());
    P.off_tbvh = B.align4(); for (float v : tbvh_tab) B.f(v);
    // ---- tables a kernel may leave in global memory (mrt_scene.h Params.lds_words_hot / lds_words_warm) ----
    P.lds_words_hot = B.align4();
    P.off_tri = B.align4(); for (float v : tri_tab) B.f(v);
    P.off_memb = B.align4(); B.w.insert(B.w.end(), memb_tab.begin(), memb_tab.end());
    P.off_membe = B.align4(); B.w.insert(B.w.end(), membe_tab.begin(), membe_tab.end());
    P.lds_words_warm = B.align4();
    // textures: RGB8 + LUT when every texel is exactly k/255 (what a decoded image file is, src/parser.rs:665)
    for (u32 t = 0; t < sc.n_textures; ++t) {
        const mrt_texture &tx = sc.textures[t];
        u32 *desc = B.w.data() + tex_desc0 + (size_t)t * TEX_WORDS;
        desc[TEX_W] = tx.w; desc[TEX_H] = tx.h;
        if (!tx.dat) { desc[TEX_FMT] = TEXFMT_NONE; continue; }
        const size_t n = (size_t)tx.w * tx.h * 3;
        bool exact = true;
        for (size_t i = 0; i < n && exact; ++i) {
            const float v = tx.dat[i];
            const float kf = rintf(v * 255.0f);
            exact = kf >= 0.0f && kf <= 255.0f && bits(kf / 255.0f) == bits(v);
        }
        const u32 off = B.align4();
        desc = B.w.data() + tex_desc0 + (size_t)t * TEX_WORDS;   // align4 may have reallocated
        if (exact) {
            desc[TEX_FMT] = TEXFMT_U8; desc[TEX_OFF] = off * 4u;
            std::vector<unsigned char> bytes(n);
            for (size_t i = 0; i < n; ++i) bytes[i] = (unsigned char)rintf(tx.dat[i] * 255.0f);
            const size_t words = (n + 3) / 4;
            const size_t at = B.w.size();
            B.w.resize(at + words, 0);
            memcpy(B.w.data() + at, bytes.data(), n);
            out.n_tex_u8++;
        } else {
            desc[TEX_FMT] = TEXFMT_F32; desc[TEX_OFF] = off;
            for (size_t i = 0; i < n; ++i) B.f(tx.dat[i]);
            out.n_tex_f32++;
        }
    }
    // the octree leaf lists come last: they are not staged in LDS (only rays the TBVH cannot cull read them)
    P.off_leaf = B.align4(); B.w.insert(B.w.end(), leaf_tab.begin(), leaf_tab.end());
    P.lds_words = P.off_leaf;
    B.align4();
    P.blob_words = (u32)B.w.size();
    P.walk_cap = tbvh_tab.empty() ? 0u : (out.tbvh_wide ? kWalkCapDefault : 8u);      // mrt_create adjusts it to the LDS budget (plan_launch)
    out.n_tbvh_nodes = (u32)(tbvh_tab.size() / (out.tbvh_wide ? B4_WORDS : BVH_WORDS));
    out.blob.swap(B.w);
    out.n_nodes = (u32)(node_tab.size() / NODE_WORDS);
    out.n_leaf_ids = (u32)leaf_tab.size();
    out.n_tris = (u32)(tri_tab.size() / TRI_WORDS);
    out.n_xf = (u32)(xf_tab.size() / XF_WORDS);
    return MRT_OK;
}

}  // namespace mrt
