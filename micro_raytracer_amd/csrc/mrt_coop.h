// mrt_coop.h — the workgroup-cooperative form of the per-lane path tracer, for scenes whose rays only sometimes meet a mesh.
//
// In pt_megakernel a lane answers its own mesh queries: a wavefront pays the longest triangle-BVH walk of its 64 lanes even
// when most of them miss the mesh's root box and walk nothing (967-triangle bench scene: 56 % of the closest-hit queries and
// ~20 % of the lanes with a shadow query reach a walk).  Here the 16 wavefronts of a 1024-thread workgroup run their loop
// iterations in step and hand their mesh queries to a REQUEST QUEUE in LDS:
//   owner side     the lane traces its path exactly as render_pixel does (same statements, same arithmetic); for a mesh instance
//                  it runs only the reference's root-box test (src/rt.rs:745) and, when the ray passes, appends (ray in the
//                  instance's frame, instance) to the queue;
//   barrier
//   consumer side  every wavefront of the workgroup -- also those whose own lanes have nothing to ask -- pulls batches of 64
//                  requests and runs mesh_isect on them: full wavefronts of rays that DO walk; a closest-hit answer goes back
//                  through a 64-bit LDS atomic minimum on (total_cmp key of t0, request slot) per owner (a lane's requests
//                  get increasing slots in instance order, so the minimum is the reference's first minimum), a shadow
//                  answer through a flag;
//   barrier        the owner merges the answer with its scan of the other instances ((key, instance index) order).
// The image cannot tell which lane walked a ray: the result is the same function of (scene, seed, sample index).
// Every barrier below is at the top level of a loop that is uniform over the wavefronts that share it, whose only exit is a
// value all of them read after the same barrier.
#pragma once
#include "mrt_trace.h"

namespace mrt {

#if defined(__HIPCC__)

typedef __attribute__((address_space(3))) u32 lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;

// The wavefronts of a workgroup cooperate in GROUPS of kCoopGroupWaves (a queue, counters and a barrier per group): one group
// for the whole workgroup serialises its phases -- all 16 wavefronts walk (latency-bound, half of them without a batch), then
// all shade (issue-bound) -- where independent groups overlap each other's phases like the independent wavefronts of
// pt_megakernel do.
#ifndef MRT_COOP_GROUP
#define MRT_COOP_GROUP 4
#endif
constexpr u32 kCoopGroupWaves = MRT_COOP_GROUP;

struct Coop {
    lds_u64 *best;          // [threads]: closest-hit answers (key << 32 | slot), or the shadow flag in the low word
    lds_u32 *rec;           // this group's records [capacity][kCoopRecWords]
    lds_u32 *cnt;           // this group's counters [2][CC_WORDS]
    lds_u32 *bar;           // this group's barrier word (monotonic arrivals)
    u32 capacity;           // records per group
    u32 epoch;              // arrivals expected at the next barrier (kept per wavefront, the same in all wavefronts of a group)
};

// Barrier of one group of wavefronts.  The whole workgroup: the hardware barrier.  A part of it: arrivals on an LDS word that
// only grows; every wavefront of the group passes the same sequence of barriers (the loops that contain them are
// group-uniform), so the k-th barrier is complete when the word has reached k x (wavefronts in the group).
template <int BLOCK_THREADS>
__device__ inline void coop_barrier(Coop &C)
{
    if constexpr (kCoopGroupWaves * 64u >= (u32)BLOCK_THREADS) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        C.epoch += kCoopGroupWaves;
        if ((threadIdx.x & 63u) == 0u) {
            __hip_atomic_fetch_add((u32 *)C.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while ((i32)(__hip_atomic_load((u32 *)C.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - C.epoch) < 0) __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// Renderer::intersect for everything but meshes (the owner's part of the scan): isect_instance of mrt_trace.h without its mesh arm
template <u32 FEAT>
__device__ inline bool isect_plain(const Scn &S, const RayPre &ray, u32 i, const F4 &ia, const F4 &ib, float &t0, float &t1)
{
    const float *F = S.U;
    const Params &P = *S.P;
    const V3 pos = v3(ia.x, ia.y, ia.z);
    const u32 tag = f2u(ib.x);
    const u32 kind = tag & TAG_KIND_MASK;
    const bool ident = (tag & TAG_IDENT) != 0;
    const float *X = F + P.off_xf + (tag >> TAG_XF_SHIFT);
    const V3 ro = add(pos, xf_vec(X, ident, sub(ray.o, pos)));
    const bool fast_d = ident && ray.d_ok;
    V3 rd = ray.d, m = ray.m;
    float dd = ray.dd;
    if (!fast_d) {
        rd = xf_full(X, ray.d);
        m = recip_patched(rd);
        dd = dot(rd, rd);
    }
    t0 = 0.0f; t1 = 0.0f;
    if (kind == KIND_SPHERE) return sphere_isect(ia.w, sub(ro, pos), rd, dd, t0, t1);
    if (kind == KIND_PLANE) { const bool h = plane_isect(v3(ib.y, ib.z, ib.w), ia.w, ro, rd, t0); t1 = t0; return h; }
    if (kind == KIND_BOX) return box_isect(v3(ia.w, ib.y, ib.z), ro, m, pos, t0, t1);
    if (kind == KIND_TRIANGLE) {
        const float *R = F + P.off_rend + ldu(F, P.off_instx + i * INSTX_WORDS + INSTX_REND) * REND_WORDS;
        const bool h = tri_isect(add(ld3(R, REND_GEO), pos), ld3(R, REND_GEO + 3), ld3(R, REND_GEO + 6), ro, rd, t0);
        t1 = t0;
        return h;
    }
    return false;
}

// Owner: mesh instance i.  The ray in the instance's frame; the reference's root-box test decides whether there is anything to
// walk (mesh_isect repeats it on the consumer side: a few instructions against a walk).  Returns true when a request was
// queued; `inline_hit` is the answer when the queue is full and the lane had to walk itself.
template <bool ANY, u32 FEAT>
__device__ inline void coop_mesh_request(const Scn &S, const Coop &C, u32 parity, const RayPre &ray, u32 i, const F4 &ia, const F4 &ib,
                                         bool &inline_done, bool &inline_hit, float &t0, float &t1, i32 &i0, i32 &i1)
{
    const float *F = S.U;
    const Params &P = *S.P;
    const V3 pos = v3(ia.x, ia.y, ia.z);
    const u32 tag = f2u(ib.x);
    const bool ident = (tag & TAG_IDENT) != 0;
    const float *X = F + P.off_xf + (tag >> TAG_XF_SHIFT);
    const V3 ro = add(pos, xf_vec(X, ident, sub(ray.o, pos)));
    const bool fast_d = ident && ray.d_ok;
    V3 rd = ray.d, m = ray.m;
    float dd = ray.dd;
    if (!fast_d) {
        rd = xf_full(X, ray.d);
        m = recip_patched(rd);
        dd = dot(rd, rd);
    }
    inline_done = false; inline_hit = false;
    const float *R = F + P.off_rend + ldu(F, P.off_instx + i * INSTX_WORDS + INSTX_REND) * REND_WORDS;
    const u32 mesh = ldu(R, REND_GEO);
    const float *M = S.F + P.off_mesh + mesh * MESH_WORDS;
    const u32 root = ldu(M, MESH_ROOT), tb = ldu(M, MESH_TBVH);
    const V3 ol = sub(ro, pos);
    if (tb != NO_NODE && cull_ok(ol, dd) && fabs_(pos.x) < 1e6f && fabs_(pos.y) < 1e6f && fabs_(pos.z) < 1e6f) {
        const float *N0 = S.F + P.off_node;
        float a0, a1;
        if (!box_isect(ld3(N0, root * NODE_WORDS + NODE_HALF), ro, m, add(pos, ld3(N0, root * NODE_WORDS + NODE_REL)), a0, a1)) return;   // no candidates, src/rt.rs:745
    }
    const u32 slot = atomicAdd((u32 *)(C.cnt + parity * CC_WORDS + CC_NREQ), 1u);
    if (slot >= C.capacity) {                      // queue full: this lane walks itself (same answer, the old cost)
        inline_done = true;
        inline_hit = mesh_isect<ANY, FEAT>(S, mesh, ro, rd, dd, m, pos, t0, i0, t1, i1);
        return;
    }
    lds_u32 *r = C.rec + slot * kCoopRecWords;
    r[0] = f2u(ro.x); r[1] = f2u(ro.y); r[2] = f2u(ro.z); r[3] = f2u(rd.x); r[4] = f2u(rd.y); r[5] = f2u(rd.z);
    r[6] = (threadIdx.x << kCoopInstBits) | i;
}

// One query round of the workgroup: [owners have queued] barrier [everybody walks] barrier.  Returns the number of wavefronts
// that still had live lanes when the round began (closest-hit rounds count them; the loop ends on 0).
template <bool ANY, int BLOCK_THREADS, u32 FEAT>
__device__ inline u32 coop_round(const Scn &S, Coop &C, u32 parity)
{
    coop_barrier<BLOCK_THREADS>(C);
    lds_u32 *cn = C.cnt + parity * CC_WORDS;
    const u32 n_raw = cn[CC_NREQ], live = cn[CC_LIVE];
    const u32 n = n_raw < C.capacity ? n_raw : C.capacity;
    if ((threadIdx.x & (kCoopGroupWaves * 64u - 1u)) == 0u) {      // the other parity's counters: nobody touches them before the next barrier
        lds_u32 *o = C.cnt + (parity ^ 1u) * CC_WORDS;
        o[CC_NREQ] = 0u; o[CC_NEXT] = 0u; o[CC_LIVE] = 0u;
    }
    const u32 lane = threadIdx.x & 63u;
    const u32 n_batches = (n + 63u) >> 6;
    const Params &P = *S.P;
    for (;;) {
        u32 bt = 0;
        if (lane == 0) bt = atomicAdd((u32 *)(cn + CC_NEXT), 1u);
        bt = __builtin_amdgcn_readfirstlane(bt);
        if (bt >= n_batches) break;
        const u32 q = bt * 64u + lane;
        if (q < n) {
            lds_u32 *r = C.rec + q * kCoopRecWords;
            const V3 ro = v3(u2f(r[0]), u2f(r[1]), u2f(r[2])), rd = v3(u2f(r[3]), u2f(r[4]), u2f(r[5]));
            const u32 who = r[6];
            const u32 owner = who >> kCoopInstBits, i = who & ((1u << kCoopInstBits) - 1u);
            const V3 pos = ld3(S.F, P.off_inst + i * INST_WORDS + INST_POS);
            const float *R = S.F + P.off_rend + ldu(S.F, P.off_instx + i * INSTX_WORDS + INSTX_REND) * REND_WORDS;
            const u32 mesh = ldu(R, REND_GEO);
            const V3 m = recip_patched(rd);
            const float dd = dot(rd, rd);
            float t0 = 0.0f, t1 = 0.0f;
            i32 i0 = -1, i1 = -1;
            if (mesh_isect<ANY, FEAT>(S, mesh, ro, rd, dd, m, pos, t0, i0, t1, i1)) {
                if (ANY) {
                    *(lds_u32 *)(C.best + owner) = 1u;
                } else {
                    r[0] = f2u(t0); r[1] = f2u(t1); r[2] = (u32)i0; r[3] = (u32)i1;
                    const unsigned long long pair = ((unsigned long long)((u32)total_key(t0) ^ 0x80000000u) << 32) | q;
                    atomicMin((unsigned long long *)(C.best + owner), pair);
                }
            }
        }
    }
    coop_barrier<BLOCK_THREADS>(C);
    return live;
}

// The cooperative main loop of one workgroup: the statements of render_pixel (mrt_trace.h), regrouped so that every mesh query
// sits between two workgroup barriers.  Wavefronts draw (tile, sample-split lane) items from P.tile_counter like the
// persistent pt_megakernel.
template <int BLOCK_THREADS, u32 FEAT>
__device__ inline void render_coop(const Scn &S, Coop &C, LdsStash<BLOCK_THREADS> &st, u32 &segments)
{
    const Params &P = *S.P;
    const u32 lane = threadIdx.x & 63u;
    const V3 sky_init = v3(P.sky_init[0], P.sky_init[1], P.sky_init[2]);
    const u32 s_base = P.sample_base, s_stop = P.sample_base + P.n_samples;
    const u32 g0 = s_base / kChunk;
    const u32 n_chunks = (s_stop - 1u) / kChunk - g0 + 1u;
    const bool direct = P.k_split == 1u;
    const u32 n_tx = (P.nw + 7u) >> 3, n_ty = (P.local_rows + 7u) >> 3;
    const u32 per_k = n_tx * n_ty, total = per_k * P.k_split;
    const float *I = S.U + P.off_inst;

    bool alive = false, wave_done = false;
    V3 csum = v3(0.0f, 0.0f, 0.0f);
    u32 pk = 0, b = 0, s = 0, seg = 0;
    V3 o = v3(0, 0, 0), d = v3(0, 1, 0);
    V3 T = v3(1, 1, 1), L = v3(0, 0, 0);
    float pwr = 1.0f;
    u32 parity = 0;

    for (;;) {
        // ---- a wavefront whose lanes have all finished draws its next tile (wave-level, no barrier) ----
        while (!wave_done && !wave_any(alive)) {
            u32 t = 0;
            if (lane == 0) t = atomicAdd(P.tile_counter, 1u);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= total) { wave_done = true; break; }
            const u32 k = t / per_k, r = t - k * per_k;
            const u32 ty = r / n_tx, tx = r - ty * n_tx;
            const u32 x = tx * 8u + (lane & 7u), ry = ty * 8u + (lane >> 3);
            const u32 blk = ry / P.shard_rows;
            const u32 y = (blk * P.shard_count + P.shard_index) * P.shard_rows + (ry - blk * P.shard_rows);
            if (x < P.nw && ry < P.local_rows && y < P.nh) {
                const u32 pix_key = mix32(y * P.nw + x + P.seed_lo) ^ P.seed_hi;
                st_put3(st, ST_FOCUS, pixel_focus(P, (float)x, (float)y));
                st.put(ST_PIXKEY, u2f(pix_key));
                st.put(ST_WORD, u2f((ry * P.nw + x) * 3u));
                st.put(ST_CHUNK, u2f(k));
                s = (g0 + k) * kChunk;
                if (s < s_base) s = s_base;
                u32 e = (g0 + k + 1u) * kChunk;
                if (e > s_stop) e = s_stop;
                st.put(ST_SEND, u2f(e));
                alive = k < n_chunks;
                csum = v3(0.0f, 0.0f, 0.0f);
                T = v3(1, 1, 1); L = v3(0, 0, 0); pwr = 1.0f; b = 0;
                if (alive) {
                    pk = mix32(pix_key + s * kGold);
                    camera_ray(P, S.F + P.off_cam, st_get3(st, ST_FOCUS), pk, o, d);
                }
            }
        }

        // ---- closest-hit query (RayTracer::closest_hit, src/rt.rs:867-898) ----
        Hit h;
        h.rend = -1; h.inst = 0; h.t0 = 0.0f; h.t1 = 0.0f; h.i0 = -1; h.i1 = -1;
        i32 best_key = 0x7fffffff;
        RayPre ray = ray_pre<FEAT>(o, d);
        auto better = [&](i32 key, u32 i) {
            const unsigned long long pair = ((unsigned long long)((u32)key ^ 0x80000000u) << 32) | i;
            const unsigned long long best_pair = ((unsigned long long)((u32)best_key ^ 0x80000000u) << 32) | (h.rend < 0 ? 0xffffffffu : h.inst);
            return pair < best_pair;
        };
        if (alive) {
            C.best[threadIdx.x] = ~0ull;
            for (u32 j = 0; j < P.n_inst; ++j) {
                const F4 ia = ld4(I, j * INST_WORDS), ib = ld4(I, j * INST_WORDS + 4);
                float t0, t1;
                i32 i0 = -1, i1 = -1;
                bool hit;
                if ((f2u(ib.x) & TAG_KIND_MASK) == KIND_MESH) {
                    bool done;
                    coop_mesh_request<false, FEAT>(S, C, parity, ray, j, ia, ib, done, hit, t0, t1, i0, i1);
                    if (!done) hit = false;
                } else {
                    hit = isect_plain<FEAT>(S, ray, j, ia, ib, t0, t1);
                }
                if (hit) {
                    const i32 key = total_key(t0);
                    if (better(key, j)) { best_key = key; h.rend = 0; h.inst = j; h.t0 = t0; h.t1 = t1; h.i0 = i0; h.i1 = i1; }
                }
            }
        }
        const bool wave_live = wave_any(alive);
        if (lane == 0 && wave_live) atomicAdd((u32 *)(C.cnt + parity * CC_WORDS + CC_LIVE), 1u);
        const u32 live = coop_round<false, BLOCK_THREADS, FEAT>(S, C, parity);
        parity ^= 1u;
        if (live == 0u) break;                                          // (every thread read the same word after the same barrier)

        bool ended = false, lit = false;
        V3 contrib = v3(0.0f, 0.0f, 0.0f), X = v3(0.0f, 1.0f, 0.0f), base = o;
        V3 color = v3(0, 0, 0), hp = v3(0, 0, 0), hn = v3(0, 0, 0), p0 = v3(0, 0, 0), l_col = v3(0, 0, 0);
        float rough_h = 0.0f, metal_h = 0.0f;
        if (alive) {
            ++seg;
            {   // the workgroup's answer for this lane's mesh requests
                const unsigned long long a = C.best[threadIdx.x];
                if (a != ~0ull) {
                    const u32 slot = (u32)a;
                    const lds_u32 *r = C.rec + slot * kCoopRecWords;
                    const i32 key = (i32)((u32)(a >> 32) ^ 0x80000000u);
                    const u32 j = r[6] & ((1u << kCoopInstBits) - 1u);
                    if (better(key, j)) { best_key = key; h.rend = 0; h.inst = j; h.t0 = u2f(r[0]); h.t1 = u2f(r[1]); h.i0 = (i32)r[2]; h.i1 = (i32)r[3]; }
                }
            }
            if (h.rend < 0) {
                contrib = (b == 0) ? v3(P.sky[0], P.sky[1], P.sky[2]) : add(L, hadam(T, sky_init));      // src/rt.rs:957-959, 964
                ended = true;
            } else {
                h.rend = (i32)ldu(S.F, P.off_instx + h.inst * INSTX_WORDS + INSTX_REND);
                const Obj ob = obj_of(S, h);
                p0 = add(o, muls(d, h.t0));
                const V3 nh0 = to_object(ob, p0);
                const Surf sf0 = surf_of<FEAT>(S, h, ob, nh0);
                const float opacity0 = surf_scalar<FEAT>(S, sf0, MAP_OPACITY, MAT_OPACITY);
                const float metal_c = sf0.M[MAT_METAL];
                bool refr = coin(fmin_(1.0f - opacity0, 0.85f), pk, dim_of(b, SL_OPAC_COIN));      // src/rt.rs:1054
                Surf sfh;
                for (;;) {                                                  // Ray::reflect / Ray::refract, src/rt.rs:559-589, 1049-1058
                    hp = refr ? add(o, muls(d, h.t1)) : p0;
                    const V3 nhh = refr ? to_object(ob, hp) : nh0;
                    hn = hit_normal<FEAT>(S, ob, nhh, refr ? h.i1 : h.i0);
                    sfh = sf0;
                    if (refr) sfh = surf_of<FEAT>(S, h, ob, nhh);
                    float rough = surf_scalar<FEAT>(S, sfh, MAP_ROUGH, MAT_ROUGH);
                    const float opac = refr ? surf_scalar<FEAT>(S, sfh, MAP_OPACITY, MAT_OPACITY) : opacity0;
                    const u32 dbase = dim_of(b, refr ? SL_REFR_COIN : SL_REFL_COIN);
                    if (metal_c == 0.0f && opac != 0.0f && draw_u32(pk, dbase) < kThr080) rough = 1.0f;
                    const V3 nn = rand_normal(hn, rough, u32_to_unit(draw_u32(pk, dbase + 1u)), u32_to_unit(draw_u32(pk, dbase + 2u)));
                    if (refr) {
                        const float eta = 1.0f + 0.5f * surf_scalar<FEAT>(S, sfh, MAP_GLASS, MAT_GLASS);
                        if (refract(d, eta, nn, X)) break;
                        refr = false;
                        continue;
                    }
                    X = reflect(d, nn);
                    break;
                }
                base = hp;
                color = surf_color<FEAT>(S, sfh);
                const float emit = surf_scalar<FEAT>(S, sfh, MAP_EMIT, MAT_EMIT);
                if (coin(emit, pk, dim_of(b, SL_EMIT_COIN))) {              // src/rt.rs:966-970
                    contrib = add(L, hadam(T, color));
                    ended = true;
                } else {
                    lit = true;
                    rough_h = surf_scalar<FEAT>(S, sfh, MAP_ROUGH, MAT_ROUGH);
                    metal_h = surf_scalar<FEAT>(S, sfh, MAP_METAL, MAT_METAL);
                }
            }
        }

        // ---- direct light: one shadow round per light (src/rt.rs:1027-1046, 973-987) ----
        for (u32 li = 0; li < P.n_light; ++li) {
            bool ask = false, blocked = false;
            V3 term = v3(0, 0, 0);
            if (alive && lit) {
                const float *Lt = S.F + P.off_light + li * LIGHT_WORDS;
                const bool point = ldu(Lt, LIGHT_KIND) == LK_POINT;
                const V3 lv = ld3(Lt, LIGHT_V);
                const V3 ln = point ? norm(sub(lv, hp)) : lv;
                const float diff = fmax_(dot(ln, hn), 0.0f);
                const float sp = fmax_(dot(d, reflect(ln, hn)), 0.0f);
                const float s2 = sp * sp, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8, s32 = s16 * s16;
                const float spec = s32 * (1.0f - rough_h);
                const V3 o_col = muls(color, 1.0f - metal_h);
                V3 t = hadam(muls(o_col, diff), ld3(Lt, LIGHT_COLOR));
                t = v3(t.x + spec, t.y + spec, t.z + spec);
                term = muls(t, Lt[LIGHT_PWR]);
                if (!(term.x == 0.0f && term.y == 0.0f && term.z == 0.0f)) {      // (a zero term cannot change l_col: no shadow ray)
                    ask = true;
                    const V3 ls = point ? norm(sub(lv, p0)) : lv;
                    const V3 so = add(p0, muls(ls, kE));
                    const RayPre sray = ray_pre<FEAT>(so, ls);
                    *(lds_u32 *)(C.best + threadIdx.x) = 0u;
                    for (u32 j = 0; j < P.n_inst; ++j) {                    // any hit at all blocks the light, src/rt.rs:1036
                        const F4 ia = ld4(I, j * INST_WORDS), ib = ld4(I, j * INST_WORDS + 4);
                        if (blocked) continue;
                        float t0, t1;
                        i32 i0, i1;
                        if ((f2u(ib.x) & TAG_KIND_MASK) == KIND_MESH) {
                            bool done, hit;
                            coop_mesh_request<true, FEAT>(S, C, parity, sray, j, ia, ib, done, hit, t0, t1, i0, i1);
                            if (done && hit) blocked = true;
                        } else if (isect_plain<FEAT>(S, sray, j, ia, ib, t0, t1)) {
                            blocked = true;
                        }
                    }
                }
            }
            (void)coop_round<true, BLOCK_THREADS, FEAT>(S, C, parity);
            parity ^= 1u;
            if (ask && !blocked && *(lds_u32 *)(C.best + threadIdx.x) == 0u) l_col = add(l_col, term);
        }

        if (alive) {
            if (lit) {
                if (P.n_light) L = add(L, hadam(T, muls(l_col, pwr)));       // the fold step, src/rt.rs:990-992, front to back
                T = hadam(T, muls(v3(0.5f + color.x, 0.5f + color.y, 0.5f + color.z), pwr));
                pwr = pwr * P.q;                                            // Ray::cast, src/rt.rs:571
                ++b;
                if (b > P.bounce) { contrib = add(L, hadam(T, sky_init)); ended = true; }      // src/rt.rs:1018
            }
            bool from_camera = false;
            if (ended) {
                csum = add(csum, contrib);
                ++s;
                if (s == f2u(st.get(ST_SEND))) {                        // chunk complete: flush its sum (canonical order, mrt_trace.h)
                    u32 j = f2u(st.get(ST_CHUNK));
                    if (direct) {
                        float *q = P.accum + f2u(st.get(ST_WORD));
                        q[0] += csum.x; q[1] += csum.y; q[2] += csum.z;
                    } else {
                        float *q = P.partial + ((size_t)j * P.partial_stride + f2u(st.get(ST_WORD)));
                        q[0] = csum.x; q[1] = csum.y; q[2] = csum.z;
                    }
                    csum = v3(0.0f, 0.0f, 0.0f);
                    j += P.k_split;
                    alive = j < n_chunks;
                    s = (g0 + j) * kChunk;
                    u32 e = s + kChunk;
                    if (e > s_stop) e = s_stop;
                    st.put(ST_CHUNK, u2f(j));
                    st.put(ST_SEND, u2f(e));
                }
                if (alive) {                                            // next sample: RayTracer::cast, src/rt.rs:916-922
                    pk = mix32(f2u(st.get(ST_PIXKEY)) + s * kGold);
                    base = lens_pos(P, pk);
                    X = sub(st_get3(st, ST_FOCUS), base);
                    T = v3(1.0f, 1.0f, 1.0f); L = v3(0.0f, 0.0f, 0.0f);
                    pwr = 1.0f; b = 0;
                    from_camera = true;
                }
            }
            V3 nd = norm(X);                                            // Ray::cast src/rt.rs:551-553; cast_default :555-557
            if (from_camera && !(P.cam_ident && nzfin3(nd))) nd = m3mul(S.F + P.off_cam + 9, m3mul(S.F + P.off_cam, nd));
            o = add(base, muls(nd, kE));
            d = nd;
        }
    }
    segments = seg;
}

#endif  // __HIPCC__

}  // namespace mrt
