// mrt_api.cpp — the C ABI of include/mrt.h: context management, uploads, launches, read-back.
// Replaces the reference's Sampler (src/sampler.rs:11-100); there is no CPU rendering path here:
// without a HIP device every entry point that needs one fails with MRT_ERR_DEVICE.
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mrt.h"
#include "mrt_kernels.h"
#include "mrt_pack.h"
#include "mrt_trace.h"

using namespace mrt;

namespace {

thread_local std::string g_err;
thread_local int g_status = MRT_OK;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    g_status = code;
    return code;
}
void ok() { g_status = MRT_OK; }

// A HIP call whose failure is tolerated: HIP 7 keeps the last *real* error pending (hipGetLastError no longer reports the
// last call's status), so the pending error is cleared here or the next launch's hipGetLastError() would report it.
bool hip_tolerated(hipError_t e)
{
    if (e == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(MRT_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

// ---- RCCL, loaded on demand (only in-process multi-device contexts need it; signatures from rccl/rccl.h) ----
typedef struct ncclComm *ncclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;                                        // rccl.h:236
    int (*CommDestroy)(ncclComm_t) = nullptr;                                                              // rccl.h:260
    int (*Gather)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;             // rccl.h:745
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;                                                          // rccl.h:339
    std::mutex mu;
    bool load(std::string &err)
    {
        std::lock_guard<std::mutex> lock(mu);
        if (lib) return true;
        for (const char *name : {"/opt/rocm/lib/librccl.so", "librccl.so", "librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        Gather = (decltype(Gather))dlsym(lib, "ncclGather");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !Gather || !GroupStart || !GroupEnd || !GetErrorString) { err = "librccl.so lacks a required symbol"; return false; }
        return true;
    }
};
Rccl g_rccl;
constexpr int kNcclFloat = 7;                        // ncclFloat32, rccl.h:466

constexpr size_t kLdsLimit = 160u * 1024u;          // LDS per CU on gfx950
constexpr size_t kSmallScene = 6u * 1024u;          // <= this: launches of less than one sample chunk take the plain grid (no tile counter)
// Sample-split until the launch has ~25 rounds of 32 waves per CU: shorter wavefronts balance the tail of a launch (tiles
// differ in path length).  Measured on the 1080p x 1024 spp Cornell box (tests/gpu_shard_probe.py): whole frame 328 -> 315 ms
// with 4 lanes per pixel, one shard of 8 GPUs 47.1 -> 42.4 ms with 16; round 3, persistent 256-thread workgroups: 1 / 2 / 4 / 8
// lanes per pixel 292.9 / 274.7 / 267.7 / 265.7 ms, hence 8 at 1080p (the target was 120 000 wavefronts: 4).
constexpr unsigned long long kSplitTargetWaves = 200000ull;
// A sample-split launch writes one f32x3 chunk sum per pixel per 16 samples; the buffer is bounded by cutting one
// mrt_execute into several launches of at most this many sample chunks (1024 samples) and this many bytes.  Launch
// boundaries are chunk boundaries, so the canonical accumulation order -- and every bit -- is unchanged.
constexpr u32 kMaxChunksPerLaunch = 64u;
constexpr size_t kPartialBudgetBytes = (size_t)4u << 30;

// an environment switch is on when it holds a non-zero number ("MRT_DEFER=0" and an empty value are off)
bool env_on(const char *name)
{
    const char *v = getenv(name);
    return v && *v && strtol(v, nullptr, 10) != 0;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for every instantiation: once per device, not per mrt_create.  A device counts
// as configured only after a SUCCESSFUL pass, so a transient failure is retried by the next mrt_create instead of being
// returned for the rest of the process; the table grows with the device index.
std::mutex g_cfg_mu;
std::vector<char> g_cfg_done;
hipError_t configure_pt_once(int device)      // the caller has made `device` current
{
    if (device < 0) return configure_pt(kLdsLimit);
    std::lock_guard<std::mutex> lock(g_cfg_mu);
    if ((size_t)device < g_cfg_done.size() && g_cfg_done[device]) return hipSuccess;
    const hipError_t e = configure_pt(kLdsLimit);
    if (e != hipSuccess) return e;
    if ((size_t)device >= g_cfg_done.size()) g_cfg_done.resize((size_t)device + 1, 0);
    g_cfg_done[device] = 1;
    return hipSuccess;
}

}  // namespace

struct mrt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;    // img timing; on a group context: around scatter_rows
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;  // sub-context of a group: around this rank's part of the ncclGather, on its stream
    std::vector<hipEvent_t> evs;                  // 3 per launch of the last execute: start, after pt_megakernel, after reduce_chunks
    u32 ev_used = 0;
    bool stats_pending = false;                   // event times / segment counter of the last execute not read back yet
    bool count_segments = false;                  // MRT_FLAG_COUNT_SEGMENTS
    bool event_timing = true;                     // !MRT_FLAG_NO_EVENT_TIMING
    bool defer = false;                           // MRT_FLAG_DEFER / MRT_DEFER=1
    bool handed_out = false;                      // mrt_accum_device_ptr gave the raw device pointer away: sticky, the caller may read it at any time
    bool bound = false;                           // the accumulator lives in caller memory (a successful mrt_bind_accum)
    bool exposed() const { return handed_out || bound; }   // either way the memory is visible behind the library's back: no deferral
    // test / experiment knobs, read from the environment ONCE in mrt_create (a thread-per-connection server calls
    // mrt_execute per sample: no getenv on that path)
    u32 knob_k_split = 0;                         // MRT_K_SPLIT: forced lanes per pixel (0: policy)
    u32 knob_max_chunks = 0;                      // MRT_MAX_CHUNKS: chunks per launch (0: kMaxChunksPerLaunch)
    size_t knob_partial_budget = 0;               // MRT_PARTIAL_LIMIT_BYTES (0: kPartialBudgetBytes)
    bool knob_partial_fail = false;               // MRT_PARTIAL_FAIL_ALLOC: the chunk-plane allocation asks for an impossible size
    bool debug_fallbacks = false;                 // MRT_DEBUG_FALLBACKS: mrt_get_stats prints the reference-walk fallbacks of the mesh queries
    u32 pending = 0;                              // samples requested by deferred mrt_execute calls and not traced yet
    // Look-ahead of the eager per-call path (the reference's callers run one Sampler::execute per sample, src/cli.rs:162-170):
    // see run_lookahead.  Two sets of per-sample planes [n][padded_rows][nw][3]; set i holds samples [la_base[i], la_base[i] + la_n[i]).
    bool la_enabled = false;                      // eager single-device context without MRT_FLAG_NO_LOOKAHEAD / MRT_LOOKAHEAD=0
    u32 la_max = 32;                              // samples per look-ahead launch at most (MRT_LOOKAHEAD=n)
    u32 la_streak = 0;                            // consecutive one-sample calls so far
    hipStream_t la_stream = nullptr;              // the look-ahead launches; the folds run on `stream`
    float *d_la[2] = {nullptr, nullptr};
    size_t la_floats[2] = {0, 0};
    u32 la_base[2] = {0, 0}, la_n[2] = {0, 0};
    hipEvent_t la_ev0[2] = {nullptr, nullptr}, la_ev1[2] = {nullptr, nullptr};   // around set i's trace launch, on la_stream
    u32 *d_la_counter = nullptr;                  // tile counter of persistent look-ahead launches (the eager launches keep their own)
    Packed pk;
    Params P;
    u32 *d_blob = nullptr;
    float *d_accum = nullptr;            // [padded_rows][nw][3], rows past local_rows stay zero
    float *d_accum_own = nullptr;        // library-owned allocation (d_accum may point to caller memory)
    float *d_partial = nullptr;          // chunk sums of a sample-split launch
    size_t partial_floats = 0;
    u32 padded_rows = 0;
    unsigned long long *d_segments = nullptr;
    u32 count = 0;                       // Sampler.last_count
    uint64_t seed = 0;
    u32 shard_index = 0, shard_count = 1, shard_rows = 8, local_rows = 0;
    std::vector<u32> row_of;             // local row -> frame row
    bool whole_frame = true;             // accumulator holds every row (shard_count == 1 or after set_accum)
    float *d_full = nullptr;             // [nh][nw][3] when a sharded context received a full frame
    u32 full_count = 0;
    u32 block_threads = 256;             // workgroup size of the batched launches
    bool small_plain_grid = false;       // launches of less than one sample chunk (the per-sample calls of the reference's callers) take
                                         // the plain grid, one workgroup per 2x2 wave tiles: no tile counter to reset and draw from
    u32 persist_grid = 0;                // persistent grid of the batched shape
    bool scene_in_lds = true;
    // img resources (lazy)
    unsigned char *d_ss = nullptr, *d_out = nullptr;
    float *d_tmp = nullptr;
    u32 *d_vl = nullptr, *d_vc = nullptr, *d_hl = nullptr, *d_hc = nullptr;
    float *d_vw = nullptr, *d_hw = nullptr;
    u32 vcap = 0, hcap = 0;
    mrt_stats stats;
    // in-process multi-device context (mrt_opts.n_devices > 1): one sharded sub-context per device, gathered on device 0
    std::vector<mrt_ctx *> subs;
    std::vector<ncclComm_t> comms;
    float *d_gather = nullptr;           // [n_devices][padded_rows][nw][3] on device 0
    u32 *d_rowmap = nullptr;             // [n_devices][padded_rows] frame row of each gathered row (0xffffffff: padding)
};

namespace {

int set_device(const mrt_ctx *c)
{
    HIP_TRY(hipSetDevice(c->device));
    return MRT_OK;
}

void free_ctx(mrt_ctx *c)
{
    if (!c) return;
    for (mrt_ctx *sub : c->subs) free_ctx(sub);
    for (ncclComm_t cm : c->comms) if (cm) g_rccl.CommDestroy(cm);
    (void)hipSetDevice(c->device);
    if (c->d_gather) (void)hipFree(c->d_gather);
    if (c->d_rowmap) (void)hipFree(c->d_rowmap);
    if (c->la_stream) { (void)hipStreamSynchronize(c->la_stream); (void)hipStreamDestroy(c->la_stream); }
    for (int i = 0; i < 2; ++i) { if (c->la_ev0[i]) (void)hipEventDestroy(c->la_ev0[i]); if (c->la_ev1[i]) (void)hipEventDestroy(c->la_ev1[i]); }
    void *ptrs[] = {c->d_blob, c->d_accum_own, c->d_partial, c->d_segments, c->d_full, c->d_ss, c->d_out, c->d_tmp, c->d_vl, c->d_vc, c->d_hl, c->d_hc, c->d_vw, c->d_hw,
                    c->d_la[0], c->d_la[1], c->d_la_counter};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_g0) (void)hipEventDestroy(c->ev_g0);
    if (c->ev_g1) (void)hipEventDestroy(c->ev_g1);
    for (hipEvent_t e : c->evs) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

}  // namespace

// error text / status setter for the other translation units of the library (mrt_image_io.cpp)
int mrt_internal_fail(int code, const char *msg) { return fail(code, "%s", msg); }
void mrt_internal_ok() { ok(); }

extern "C" {

uint32_t mrt_abi_version(void) { return MRT_ABI_VERSION; }
const char *mrt_last_error(void) { return g_err.c_str(); }
int mrt_last_status(void) { return g_status; }

int mrt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}


namespace {

// What a context stages in LDS and the shape of its launches: a pure function of the packed scene (and of the experiment /
// test knobs of the environment) -- no device involved, so that the policy can be checked without one (mrt_plan_launch).
struct Plan {
    bool in_lds = true;
    u32 block_threads = 256;
    bool small_plain_grid = false;
    size_t staged_bytes = 0;
};

void plan_launch(const mrt_render_desc *desc, Packed &pk, Plan &pl)
{
    // ---- what is staged in LDS, and the launch shape -------------------------------------------------------------------
    // LDS per workgroup = staged scene + lane stash (+ the mesh kernels' walk areas): pt_lds_bytes knows.  Staging levels:
    //   all     the whole packed scene (minus the octree leaf lists);
    //   warm    F_COLD: texels stay in global memory (touched at most once per shaded hit); mesh kernels get a per-lane walk area;
    //   deep    F_COLD | F_DEEP: meshes beyond the LDS.  The triangle-BVH table is in level order, so as many of its first
    //           nodes -- the top levels of every tree -- as fit next to the small tables are staged; deeper nodes and the
    //           triangles are read from global memory too;
    //   none    everything through L2 (the small tables themselves do not fit).
    // Launch shape, chosen for resident wavefronts per CU (the kernel is VALU-issue bound and wants >= 16): the smallest
    // workgroup that reaches 16 waves per CU wins (smaller workgroups balance better), else the shape with the most:
    //   256 threads (2x2 wave tiles of 8x8 pixels) + 10 KB lane stash per copy of the scene (64-thread workgroups -- one
    //   wavefront, its own 5.5 KB of LDS -- remain as a forced shape for the tests);
    //   512 threads (4x2 tiles), no stash; 1024 threads (4x4 tiles) + 40 KB stash: one LDS copy serves 16 waves.
    // Mesh kernels of the warm and deep levels own a per-lane walk area (Params.walk_cap entries, mrt_trace.h): the leaf queue
    // of the binary walk (8 to 16 entries, warm: what the LDS has left), or node stack + leaf queue of the 4-wide walk (16
    // entries, deep: the scene is packed again with 4-wide triangle BVHs).
    // Environment (experiments, tests; read here, once): MRT_COLD=0/1 forbids / forces the warm level, MRT_DEEP_NODES=n forces
    // the deep level with n staged nodes, MRT_SCENE_IN_L2 forces none, MRT_BLOCK_THREADS forces a workgroup size, MRT_WALK_CAP
    // the entries of the deep level's walk area.
    const bool no_lds = getenv("MRT_SCENE_IN_L2") != nullptr;
    const char *force = getenv("MRT_BLOCK_THREADS");
    const bool mesh_walk = pk.n_tbvh_nodes != 0u && (pk.features & 3u) == 3u;
    u32 deep_cap = kWalkCapDefault;
    if (const char *fw = getenv("MRT_WALK_CAP")) { const int v = atoi(fw); if (v >= 4 && v <= (int)kWalkCapMax) deep_cap = (u32)v; }
    pk.P.walk_cap = mesh_walk ? kLeafQueue : 0u;         // (only kernels with a walk area count it: pt_lds_bytes)
    auto lds_of = [&](u32 shape, u32 marker) { return pt_lds_bytes(pk.P, shape, true, (pk.features & 31u) | marker); };
    auto fits = [&](u32 shape, u32 marker) { return lds_of(shape, marker) <= kLdsLimit; };
    auto fits_any = [&](u32 marker) { return fits(256u, marker) || fits(512u, marker) || fits(1024u, marker); };
    auto waves = [&](u32 shape, u32 marker) {            // resident wavefronts per CU of this shape, LDS-wise
        const size_t l = lds_of(shape, marker);
        return l > kLdsLimit ? (size_t)0 : (shape / 64u) * (kLdsLimit / (l ? l : 1));
    };
    constexpr u32 kWarm = 64u, kDeep = 64u | 128u;       // F_COLD, F_COLD | F_DEEP
    const bool has_warm = pk.P.lds_words_warm < pk.P.lds_words || mesh_walk;     // (mesh kernels: the warm marker also buys the walk area)
    u32 cold = 0u;
    bool in_lds = !no_lds;
    if (in_lds) {
        const char *fc = getenv("MRT_COLD");
        const char *fd = getenv("MRT_DEEP_NODES");
        const bool warm_ok = has_warm && fits_any(kWarm) && !(fc && !atoi(fc));
        const bool all_ok = fits_any(0u) && !(fc && atoi(fc) && warm_ok);
        // a mesh scene takes the warm level when a 16-wave workgroup fits with stash and leaf queues (closest-hit walks in one
        // round: VALU -8.5 %, time -2 % on the 967-triangle bench scene); everything else takes the whole scene when it fits;
        // an instance-BVH scene whose texels alone force a single 1024-thread workgroup per CU takes the warm level too: its
        // kernel is built for 6 waves per SIMD, which 256-thread workgroups around an LDS copy without the texels can supply
        const bool bvh_no_mesh = (pk.features & 16u) != 0u && (pk.features & 2u) == 0u;
        if (fd && mesh_walk) cold = kDeep;
        else if (mesh_walk && warm_ok && fits(1024u, kWarm)) cold = kWarm;
        else if (bvh_no_mesh && warm_ok && !fc && waves(256u, 0u) < 16u && waves(256u, kWarm) >= 24u) cold = kWarm;
        else if (all_ok) cold = 0u;
        else if (warm_ok) cold = kWarm;
        else if (mesh_walk) cold = kDeep;
        else in_lds = false;
        if (cold == kDeep) {
            // packed again with 4-wide triangle BVHs in level order; everything hot in front of the node table + lane stash +
            // walk areas of one 1024-thread workgroup; the rest of the LDS holds the first nodes of the table
            PackOpts po; po.tbvh_wide = true;
            Packed again; std::string err2;
            bool ok2 = pack_scene(desc, again, err2, po) == MRT_OK && again.tbvh_wide;
            const size_t fixed = (size_t)ST_SLOTS * 1024u * sizeof(float) + (size_t)deep_cap * 1024u * sizeof(u32) + 1024u;
            const size_t front = ok2 ? (size_t)again.P.off_tbvh * 4 : 0;
            ok2 = ok2 && front + fixed < kLdsLimit;
            if (ok2) {
                const size_t room = (kLdsLimit - fixed - front) / (B4_WORDS * 4);
                size_t n = fd ? (size_t)strtoul(fd, nullptr, 10) : room;
                if (n > room) n = room;
                if (n > again.n_tbvh_nodes) n = again.n_tbvh_nodes;
                const u32 n_mesh = (again.P.off_node - again.P.off_mesh) / MESH_WORDS;
                ok2 = n >= n_mesh && n_mesh > 0u;            // every root is staged
                if (ok2) {
                    const u32 keep = pk.features;
                    pk = again;
                    pk.features = keep;
                    pk.P.walk_cap = deep_cap;
                    pk.P.n_tbvh_hot = (u32)n;
                    pk.P.lds_words_hot = (pk.P.off_tbvh + (u32)n * B4_WORDS + 3u) & ~3u;
                }
            }
            if (!ok2) { cold = 0u; in_lds = false; }
        }
    }
    const size_t full_bytes = (size_t)pk.P.lds_words * 4;
    const size_t blob_bytes = in_lds ? (size_t)staged_words_for(pk.P, cold) * 4 : full_bytes;
    u32 want = 256u, marker = cold;
    pl.small_plain_grid = false;
    if (in_lds) {
        const size_t w256 = waves(256u, cold), w512 = waves(512u, cold), w1024 = waves(1024u, cold);
        // small scenes (<= 6 KB: ~29 single-wave workgroups per CU would fit): 256-thread workgroups all the same -- four waves
        // around one LDS copy, so that 32 waves per CU fit: +4 % on the headline frame, +7 % with the 8-wave build of the plane /
        // sphere kernel, +4 % on CornellBox2 -- persistent for batched launches, on the plain grid for launches of less than
        // one sample chunk (a one-sample pass over the 1080p frame: persistent 0.56 ms, single-wave workgroups 0.39, this 0.37)
        if (w256 >= 16u) { want = 256u; pl.small_plain_grid = blob_bytes <= kSmallScene && !cold; }
        else if (w512 >= 16u) want = 512u;
        else if (w1024 >= 16u) want = 1024u;
        else if (w256 >= w512 && w256 >= 8u) want = 256u;
        else if (!cold && fits(1024u, 32u)) { want = 1024u; marker |= 32u; }      // F_NOSTASH: one LDS copy for 16 waves, lane state in registers
        else want = w1024 ? 1024u : (w512 ? 512u : 256u);
    }
    if (force && in_lds) {
        const u32 f = (u32)atoi(force);
        if ((f == 64u && !cold) || f == 256u || f == 512u || f == 1024u) { if (fits(f, cold)) { want = f; marker = cold; pl.small_plain_grid = false; } }
    }
    pk.features = (pk.features & 31u) | marker | (pk.all_ident ? (u32)F_IDENT : 0u);      // (pt_instantiation: which shapes have F_IDENT builds)
    // the leaf queue of the warm mesh kernels takes what the LDS has left while the workgroups per CU stay the same (967-triangle
    // bench scene: 13 entries, +2 % over 8: fewer walks need a second round)
    if (in_lds && marker == kWarm && mesh_walk && has_walk_area(pk.features)) {
        const size_t w0 = waves(want, marker);
        while (pk.P.walk_cap < kWalkCapMax) {
            ++pk.P.walk_cap;
            if (waves(want, marker) != w0) { --pk.P.walk_cap; break; }
        }
    }
    pl.in_lds = in_lds;
    pl.block_threads = want;
    pl.staged_bytes = blob_bytes;
}

}  // namespace

static mrt_ctx *create_single(const mrt_render_desc *desc, const mrt_opts *opts)
{
    const u32 shard_count = opts->shard_count ? opts->shard_count : 1;
    if (opts->shard_index >= shard_count) { fail(MRT_ERR_ARG, "mrt_create: shard_index %u >= shard_count %u", opts->shard_index, shard_count); return nullptr; }

    mrt_ctx *c = new mrt_ctx();
    std::string err;
    const int rc = pack_scene(desc, c->pk, err);
    if (rc != MRT_OK) { fail(rc, "mrt_create: %s", err.c_str()); delete c; return nullptr; }
    Plan plan;
    plan_launch(desc, c->pk, plan);          // may re-pack the scene (deep staging: 4-wide triangle BVHs), before anything is uploaded

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fail(MRT_ERR_DEVICE, "mrt_create: no HIP device (this backend has no CPU path)");
        delete c; return nullptr;
    }
    int dev = opts->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { fail(MRT_ERR_ARG, "mrt_create: device %d of %d", dev, ndev); delete c; return nullptr; }
    c->device = dev;
    c->seed = opts->seed;
    c->shard_index = opts->shard_index; c->shard_count = shard_count;
    c->shard_rows = opts->shard_rows ? opts->shard_rows : 8;
    c->whole_frame = shard_count == 1;

    // rows of this shard: row block b (shard_rows rows) belongs to shard b % shard_count
    const u32 nh = c->pk.nh, nw = c->pk.nw;
    for (u32 y = 0; y < nh; ++y) if ((y / c->shard_rows) % shard_count == c->shard_index) c->row_of.push_back(y);
    c->local_rows = (u32)c->row_of.size();

    auto bail = [&](int code, const char *what, hipError_t e) { fail(code, "mrt_create: %s: %s", what, hipGetErrorString(e)); free_ctx(c); return (mrt_ctx *)nullptr; };
    hipError_t e;
    if ((e = hipSetDevice(dev)) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipEventCreate", e);
    const size_t all_bytes = (size_t)c->pk.blob.size() * 4;
    if ((e = hipMalloc((void **)&c->d_blob, all_bytes ? all_bytes : 16)) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipMalloc(scene)", e);
    if ((e = hipMemcpy(c->d_blob, c->pk.blob.data(), all_bytes, hipMemcpyHostToDevice)) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipMemcpy(scene)", e);
    {
        const u32 n_blocks = (nh + c->shard_rows - 1) / c->shard_rows;
        c->padded_rows = ((n_blocks + shard_count - 1) / shard_count) * c->shard_rows;
        if (shard_count == 1) c->padded_rows = nh;
    }
    const size_t acc_bytes = (size_t)c->padded_rows * nw * 3 * sizeof(float);
    if ((e = hipMalloc((void **)&c->d_accum_own, acc_bytes)) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipMalloc(accumulator)", e);
    c->d_accum = c->d_accum_own;
    if ((e = hipMemset(c->d_accum, 0, acc_bytes)) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipMemset", e);
    if ((e = hipMalloc((void **)&c->d_segments, 8 * sizeof(unsigned long long))) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipMalloc", e);   // + tile counter + 4 phase clocks (debug builds)
    if ((e = hipMemset(c->d_segments, 0, 8 * sizeof(unsigned long long))) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipMemset", e);

    c->scene_in_lds = plan.in_lds;
    c->small_plain_grid = plan.small_plain_grid;
    const u32 want = plan.block_threads;
    const size_t blob_bytes = plan.staged_bytes;
    c->block_threads = want;
    c->pk.P.tiles_x = want == 64u ? 1u : (want == 256u ? 2u : 4u);
    c->pk.P.tiles_y = want == 64u ? 1u : (want == 1024u ? 4u : 2u);
    if (c->scene_in_lds && (e = configure_pt_once(dev)) != hipSuccess) return bail(MRT_ERR_DEVICE, "hipFuncSetAttribute", e);

    c->P = c->pk.P;
    c->P.local_rows = c->local_rows; c->P.shard_index = c->shard_index; c->P.shard_count = c->shard_count; c->P.shard_rows = c->shard_rows;
    c->P.seed_lo = (u32)c->seed; c->P.seed_hi = (u32)(c->seed >> 32);
    c->P.blob = c->d_blob; c->P.accum = c->d_accum; c->P.segments = c->d_segments;
    c->P.tile_counter = reinterpret_cast<u32 *>(c->d_segments + 1);
    {
        // persistent launches (workgroups of more than one wavefront): as many workgroups as fit the device at once; each
        // wavefront then draws 8x8 tiles from a counter, so no CU waits for the slowest wavefront of a workgroup
        int n_cu = 0;
        if (!hip_tolerated(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device)) || n_cu <= 0) n_cu = 256;
        const size_t lds = pt_lds_bytes(c->pk.P, c->block_threads, c->scene_in_lds, c->pk.features);
        size_t per_cu = 32u / (c->block_threads / 64u);
        if (lds && kLdsLimit / lds < per_cu) per_cu = kLdsLimit / lds;
        if (per_cu < 1u) per_cu = 1u;
        c->persist_grid = c->block_threads > 64u && !getenv("MRT_NO_PERSIST") ? (u32)(n_cu * per_cu) : 0u;
        c->P.persist_grid = c->persist_grid;
    }
    c->count_segments = (opts->flags & MRT_FLAG_COUNT_SEGMENTS) != 0;
    c->event_timing = (opts->flags & MRT_FLAG_NO_EVENT_TIMING) == 0;
    c->P.count_segments = c->count_segments ? 1u : 0u;
    memset(&c->stats, 0, sizeof c->stats);
    c->stats.lds_bytes = (u32)pt_lds_bytes(c->pk.P, c->block_threads, c->scene_in_lds, c->pk.features);
    c->stats.block_threads = c->block_threads;
    c->stats.scene_bytes = (u32)blob_bytes;
    c->stats.kernel_features = pt_instantiation(c->block_threads, c->scene_in_lds, c->pk.features);
    c->stats.scene_in_lds = c->scene_in_lds ? 1u : 0u;
    c->defer = (opts->flags & MRT_FLAG_DEFER) != 0 || (env_on("MRT_DEFER") && opts->shard_count <= 1);
    if (const char *f = getenv("MRT_K_SPLIT")) { const int v = atoi(f); c->knob_k_split = v < 1 ? 1u : (u32)v; }                        // experiments / tests
    if (const char *f = getenv("MRT_MAX_CHUNKS")) { const int v = atoi(f); if (v > 0) c->knob_max_chunks = (u32)v; }                   // tests
    if (const char *f = getenv("MRT_PARTIAL_LIMIT_BYTES")) c->knob_partial_budget = (size_t)strtoull(f, nullptr, 10);                  // tests
    c->knob_partial_fail = getenv("MRT_PARTIAL_FAIL_ALLOC") != nullptr;                                                                // tests
    c->debug_fallbacks = env_on("MRT_DEBUG_FALLBACKS");
    c->la_enabled = !c->defer && (opts->flags & MRT_FLAG_NO_LOOKAHEAD) == 0;
    if (const char *f = getenv("MRT_LOOKAHEAD")) { const int v = atoi(f); if (v <= 1) c->la_enabled = false; else c->la_max = v > 64 ? 64u : (u32)v; }
    {   // both plane sets together stay below 4 GiB (32 samples of a 1080p frame: 2 x 0.8 GB; a 4K frame gets 20 per launch)
        const size_t plane_bytes = (size_t)c->padded_rows * nw * 3 * sizeof(float);
        const size_t fit = plane_bytes ? ((size_t)2u << 30) / plane_bytes : 0;
        if (fit < 2) c->la_enabled = false; else if (fit < c->la_max) c->la_max = (u32)fit;
    }
    ok();
    return c;
}

// In-process multi-device context: n sharded sub-contexts (device r renders row blocks b = r mod n), one RCCL
// ncclGather of the padded shard accumulators to device 0 per mrt_execute, rows placed into the frame by scatter_rows.
static mrt_ctx *create_group(const mrt_render_desc *desc, const mrt_opts *opts, u32 n)
{
    std::string err;
    if (!g_rccl.load(err)) { fail(MRT_ERR_DEVICE, "mrt_create: %s", err.c_str()); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || (u32)ndev < n) { fail(MRT_ERR_DEVICE, "mrt_create: n_devices = %u but %d HIP device(s) are visible", n, ndev); return nullptr; }
    mrt_ctx *g = new mrt_ctx();
    const int rc = pack_scene(desc, g->pk, err);
    if (rc != MRT_OK) { fail(rc, "mrt_create: %s", err.c_str()); delete g; return nullptr; }
    g->device = 0; g->seed = opts->seed;
    g->defer = (opts->flags & MRT_FLAG_DEFER) != 0 || env_on("MRT_DEFER");
    g->shard_count = 1; g->shard_index = 0; g->shard_rows = opts->shard_rows ? opts->shard_rows : 8;
    g->local_rows = g->pk.nh; g->padded_rows = g->pk.nh;
    for (u32 y = 0; y < g->pk.nh; ++y) g->row_of.push_back(y);
    for (u32 r = 0; r < n; ++r) {
        mrt_opts o = *opts;
        o.n_devices = 0; o.device = (int)r; o.shard_index = r; o.shard_count = n; o.shard_rows = g->shard_rows;
        o.flags &= ~MRT_FLAG_DEFER;                   // the group defers, not its shards
        mrt_ctx *sub = create_single(desc, &o);
        if (!sub) { free_ctx(g); return nullptr; }
        g->subs.push_back(sub);
    }
    auto bail = [&](const char *what, const char *why) { fail(MRT_ERR_DEVICE, "mrt_create: %s: %s", what, why); free_ctx(g); return (mrt_ctx *)nullptr; };
    hipError_t e;
    if ((e = hipSetDevice(0)) != hipSuccess) return bail("hipSetDevice", hipGetErrorString(e));
    if ((e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", hipGetErrorString(e));
    if ((e = hipEventCreate(&g->ev0)) != hipSuccess || (e = hipEventCreate(&g->ev1)) != hipSuccess) return bail("hipEventCreate", hipGetErrorString(e));
    const u32 pr = g->subs[0]->padded_rows, nw = g->pk.nw, nh = g->pk.nh;
    const size_t plane = (size_t)pr * nw * 3;
    if ((e = hipMalloc((void **)&g->d_full, (size_t)nh * nw * 3 * sizeof(float))) != hipSuccess) return bail("hipMalloc(frame)", hipGetErrorString(e));
    if ((e = hipMemset(g->d_full, 0, (size_t)nh * nw * 3 * sizeof(float))) != hipSuccess) return bail("hipMemset", hipGetErrorString(e));
    if ((e = hipMalloc((void **)&g->d_gather, plane * n * sizeof(float))) != hipSuccess) return bail("hipMalloc(gather)", hipGetErrorString(e));
    std::vector<u32> rowmap((size_t)n * pr, 0xffffffffu);
    for (u32 r = 0; r < n; ++r) for (u32 i = 0; i < g->subs[r]->local_rows; ++i) rowmap[(size_t)r * pr + i] = g->subs[r]->row_of[i];
    if ((e = hipMalloc((void **)&g->d_rowmap, rowmap.size() * sizeof(u32))) != hipSuccess) return bail("hipMalloc(rowmap)", hipGetErrorString(e));
    if ((e = hipMemcpy(g->d_rowmap, rowmap.data(), rowmap.size() * sizeof(u32), hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy(rowmap)", hipGetErrorString(e));
    std::vector<int> devs(n);
    for (u32 r = 0; r < n; ++r) devs[r] = (int)r;
    g->comms.assign(n, nullptr);
    const int nrc = g_rccl.CommInitAll(g->comms.data(), (int)n, devs.data());
    if (nrc != 0) return bail("ncclCommInitAll", g_rccl.GetErrorString(nrc));
    g->P = g->pk.P;
    memset(&g->stats, 0, sizeof g->stats);
    g->stats.block_threads = g->subs[0]->block_threads; g->stats.lds_bytes = g->subs[0]->stats.lds_bytes; g->stats.scene_bytes = g->subs[0]->stats.scene_bytes;
    g->stats.kernel_features = g->subs[0]->stats.kernel_features; g->stats.scene_in_lds = g->subs[0]->stats.scene_in_lds;
    ok();
    return g;
}

mrt_ctx *mrt_create(const mrt_render_desc *desc, const mrt_opts *opts)
{
    g_err.clear();
    if (!desc || !opts) { fail(MRT_ERR_ARG, "mrt_create: null argument"); return nullptr; }
    if (opts->abi_version != MRT_ABI_VERSION) { fail(MRT_ERR_ARG, "mrt_create: ABI version %u, library has %u", opts->abi_version, MRT_ABI_VERSION); return nullptr; }
    u32 n = opts->n_devices;
    if (n == 0) if (const char *e = getenv("MRT_GPUS")) n = (u32)atoi(e);       // the Rust shim's knob
    const bool force_group = getenv("MRT_FORCE_RCCL") != nullptr;               // tests: the group path on one device
    if (n > 1 || (n == 1 && force_group)) {
        if (opts->shard_count > 1) { fail(MRT_ERR_ARG, "mrt_create: n_devices and shard_count are mutually exclusive"); return nullptr; }
        return create_group(desc, opts, n);
    }
    return create_single(desc, opts);
}

void mrt_destroy(mrt_ctx *ctx) { free_ctx(ctx); }

// Lazy half of the per-execute statistics: HIP-event times and (MRT_FLAG_COUNT_SEGMENTS) the segment counter are read
// back when somebody asks (mrt_get_stats), not on every mrt_execute -- the reference's callers run one pass per call
// (src/cli.rs:162-170), so the per-call cost is what the drop-in binary pays 1024 times per frame.
static int resolve_stats(mrt_ctx *c)
{
    if (!c->stats_pending) return MRT_OK;
    c->stats_pending = false;
    int rc = set_device(c);
    if (rc) return rc;
    double k = 0, r = 0;
    for (u32 i = 0; i + 2u < c->ev_used; i += 3u) {
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, c->evs[i], c->evs[i + 1]));
        HIP_TRY(hipEventElapsedTime(&b, c->evs[i + 1], c->evs[i + 2]));
        k += a; r += b;
    }
    c->stats.kernel_ms = k;
    c->stats.reduce_ms = c->stats.k_split > 1u ? r : 0.0;
    if (c->count_segments) {
        unsigned long long seg = 0;
        HIP_TRY(hipMemcpy(&seg, c->d_segments, sizeof seg, hipMemcpyDeviceToHost));
        c->stats.segments = seg;
    }
    if (c->debug_fallbacks) {
        unsigned long long t[3] = {0, 0, 0};
        HIP_TRY(hipMemcpy(t, c->d_segments + 5, sizeof t, hipMemcpyDeviceToHost));
        fprintf(stderr, "[mrt fallbacks] since mrt_create: NaN directions (shortcut) %llu, walk area full %llu, rays the triangle BVH may not cull %llu\n", t[0], t[1], t[2]);
    }
#ifdef MRT_PHASE_TIMING
    {   // debug build: shader-clock ticks per phase, summed over wavefronts since the context was created
        unsigned long long t[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(t, c->d_segments + 2, sizeof t, hipMemcpyDeviceToHost));
        const double tot = (double)(t[0] + t[1] + t[2] + t[3]);
        fprintf(stderr, "[mrt phase ticks] closest-hit query %.3f  shading %.3f  shadow query %.3f  ray set-up / regeneration %.3f  (total %.4g wave-ticks)\n",
                t[0] / tot, t[1] / tot, t[2] / tot, t[3] / tot, tot);
    }
#endif
    return MRT_OK;
}

// asynchronous half of mrt_execute on one device: everything up to the closing event
static int exec_launch(mrt_ctx *c, uint32_t n_samples)
{
    int rc = set_device(c);
    if (rc) return rc;
    c->stats.kernel_ms = 0; c->stats.reduce_ms = 0; c->stats.gather_ms = 0; c->stats.launches = 0; c->stats.samples = 0; c->stats.segments = 0;
    c->stats_pending = false; c->ev_used = 0;
    if (!(n_samples && c->local_rows)) return MRT_OK;
    if (c->count_segments) HIP_TRY(hipMemsetAsync(c->d_segments, 0, sizeof(unsigned long long), c->stream));
    // sample split: spread a small frame over more wavefronts, one lane per (pixel, every k-th sample chunk)
    const u32 first_chunk = c->count / kChunk;
    const u32 end_chunk = (c->count + n_samples - 1u) / kChunk + 1u;
    const u32 n_chunks = end_chunk - first_chunk;
    const unsigned long long wave_tiles = (unsigned long long)((c->pk.nw + 7) / 8) * ((c->local_rows + 7) / 8);
    u32 k_split = 1;
    // (a frame of 100 000 wave tiles or more -- 4K -- has its 25 rounds without splitting: CornellBox2 at 3840x2160 loses 1.6 %
    // to a second lane per pixel, 985 -> 1001 ms)
    const unsigned long long split_target = wave_tiles >= 100000ull ? 0ull : kSplitTargetWaves;
    while (k_split * 2u <= n_chunks && k_split < 16u && wave_tiles * k_split < split_target) k_split *= 2u;
    if (c->knob_k_split) { k_split = c->knob_k_split; while (k_split > n_chunks) k_split /= 2u; }
    const size_t plane = (size_t)c->padded_rows * c->pk.nw * 3;
    u32 cap = kMaxChunksPerLaunch;                       // chunks per launch
    if (c->knob_max_chunks) cap = c->knob_max_chunks;
    size_t budget = kPartialBudgetBytes;
    if (c->knob_partial_budget) budget = c->knob_partial_budget;
    if (k_split > 1u) {
        if (cap < k_split) cap = k_split;
        while (cap > k_split && plane * cap * sizeof(float) > budget) cap /= 2u;
        const size_t need = plane * (n_chunks < cap ? n_chunks : cap);
        if (need * sizeof(float) > budget) k_split = 1u;                       // not even k_split planes fit: one lane per pixel
        else if (need > c->partial_floats) {
            if (c->d_partial) { (void)hipFree(c->d_partial); c->d_partial = nullptr; c->partial_floats = 0; }
            const size_t ask = c->knob_partial_fail ? ((size_t)1 << 60) : need * sizeof(float);
            if (!hip_tolerated(hipMalloc((void **)&c->d_partial, ask))) { c->d_partial = nullptr; k_split = 1u; }
            else c->partial_floats = need;
        }
    }
    c->P.partial = c->d_partial;
    c->P.partial_stride = plane;
    c->stats.k_split = k_split;
    const u32 s_end = c->count + n_samples;
    u32 base = c->count;
    while (base < s_end) {
        // this launch: samples [base, stop), stop on a chunk boundary (or the end); k_split == 1 needs no buffer: one launch
        u32 stop = s_end;
        if (k_split > 1u) {
            const unsigned long long lim = ((unsigned long long)(base / kChunk) + cap) * kChunk;
            if (lim < stop) stop = (u32)lim;
        }
        const u32 nc = (stop - 1u) / kChunk - base / kChunk + 1u;
        u32 ks = k_split;
        while (ks > nc) ks /= 2u;
        c->P.n_samples = stop - base;
        c->P.sample_base = base;
        c->P.k_split = ks;
        while (c->event_timing && c->evs.size() < (size_t)c->ev_used + 3u) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); c->evs.push_back(e); }
        hipEvent_t *ev = c->event_timing ? &c->evs[c->ev_used] : nullptr;
        const u32 bt = c->block_threads;
        c->P.persist_grid = (c->small_plain_grid && stop - base < kChunk) ? 0u : c->persist_grid;
        const bool persist = bt > 64u && c->P.persist_grid != 0u;
        if (persist) HIP_TRY(hipMemsetAsync(c->P.tile_counter, 0, sizeof(u32), c->stream));
        if (c->event_timing) HIP_TRY(hipEventRecord(ev[0], c->stream));
        HIP_TRY(launch_pt(c->P, bt, c->scene_in_lds, c->pk.features, c->stream));
        c->stats.block_threads = bt;
        c->stats.lds_bytes = (u32)pt_lds_bytes(c->P, bt, c->scene_in_lds, c->pk.features);
        c->stats.kernel_features = pt_instantiation(bt, c->scene_in_lds, c->pk.features);
        if (c->event_timing) HIP_TRY(hipEventRecord(ev[1], c->stream));
        if (ks > 1u) HIP_TRY(launch_reduce_chunks(c->d_accum, c->d_partial, (size_t)c->local_rows * c->pk.nw * 3, plane, nc, c->stream));
        if (c->event_timing) { HIP_TRY(hipEventRecord(ev[2], c->stream)); c->ev_used += 3u; }
        c->stats.launches += 1u;
        base = stop;
    }
    return MRT_OK;
}

static int exec_finish(mrt_ctx *c, uint32_t n_samples)
{
    int rc = set_device(c);
    if (rc) return rc;
    if (n_samples && c->local_rows) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->stats.samples = (uint64_t)c->local_rows * c->pk.nw * n_samples;
        c->stats_pending = true;
    }
    c->count += n_samples;                                        // src/sampler.rs:76
    return MRT_OK;
}

static int exec_group(mrt_ctx *g, uint32_t n_samples)
{
    const u32 n = (u32)g->subs.size();
    int rc = MRT_OK;
    u32 launched = 0;
    for (; launched < n; ++launched) if ((rc = exec_launch(g->subs[launched], n_samples))) break;
    // an error from here on still waits for everything that was launched, so no kernel is left running on a buffer
    // the caller may free; the first error is the one reported
    auto drain = [&](u32 upto) { for (u32 r = 0; r < upto; ++r) { if (hipSetDevice(g->subs[r]->device) == hipSuccess) (void)hipStreamSynchronize(g->subs[r]->stream); } (void)hipGetLastError(); };
    if (rc) { const std::string keep = g_err; drain(launched + (launched < n ? 1u : 0u)); g_err = keep; g_status = rc; return rc; }
    // one gather per batch: rank r sends its padded shard accumulator, device 0 receives rank i at offset i * plane.
    // gather_ms is device time, not host time around asynchronous launches: every sub-stream records an event after its
    // kernels (before its part of the gather) and one after it; the slowest stream's interval plus scatter_rows is the
    // exchange.  A rank whose kernels finish early waits inside the collective for the slowest one, so the figure is an
    // upper bound of the transfer itself; kernel_ms (the slowest rank's kernels) is reported next to it.
    const size_t plane = (size_t)g->subs[0]->padded_rows * g->pk.nw * 3;
    hipError_t he = hipSuccess;
    for (u32 r = 0; r < n && he == hipSuccess; ++r) {
        mrt_ctx *s = g->subs[r];
        if ((he = hipSetDevice(s->device)) != hipSuccess) break;
        if (!s->ev_g0 && (he = hipEventCreate(&s->ev_g0)) != hipSuccess) break;
        if (!s->ev_g1 && (he = hipEventCreate(&s->ev_g1)) != hipSuccess) break;
        he = hipEventRecord(s->ev_g0, s->stream);
    }
    int nrc = he == hipSuccess ? g_rccl.GroupStart() : 0;
    if (he == hipSuccess && nrc == 0) {
        for (u32 r = 0; r < n && nrc == 0 && he == hipSuccess; ++r) {
            if ((he = hipSetDevice(g->subs[r]->device)) != hipSuccess) break;
            nrc = g_rccl.Gather(g->subs[r]->d_accum, r == 0 ? g->d_gather : nullptr, plane, kNcclFloat, 0, g->comms[r], g->subs[r]->stream);
        }
        const int nrc2 = g_rccl.GroupEnd();                         // always closed, whatever happened inside the group
        if (nrc == 0) nrc = nrc2;
    }
    for (u32 r = 0; r < n && he == hipSuccess && nrc == 0; ++r) {
        if ((he = hipSetDevice(g->subs[r]->device)) != hipSuccess) break;
        he = hipEventRecord(g->subs[r]->ev_g1, g->subs[r]->stream);
    }
    if (he != hipSuccess || nrc != 0) {
        drain(n);
        if (he != hipSuccess) return fail(MRT_ERR_DEVICE, "HIP call failed around the gather group: %s", hipGetErrorString(he));
        return fail(MRT_ERR_DEVICE, "ncclGather: %s", g_rccl.GetErrorString(nrc));
    }
    for (mrt_ctx *s : g->subs) if ((rc = exec_finish(s, n_samples))) { const std::string keep = g_err; drain(n); g_err = keep; g_status = rc; return rc; }    // syncs every stream (kernel + gather)
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipEventRecord(g->ev0, g->stream));
    HIP_TRY(launch_scatter_rows(g->d_full, g->d_gather, g->d_rowmap, n * g->subs[0]->padded_rows, g->pk.nw * 3u, g->stream));
    HIP_TRY(hipEventRecord(g->ev1, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->count += n_samples;
    g->full_count = g->count;
    memset(&g->stats, 0, offsetof(mrt_stats, lds_bytes));
    g->stats.reduce_ms = 0;
    double gather_ms = 0;
    for (mrt_ctx *s : g->subs) {
        if ((rc = resolve_stats(s))) return rc;
        if (s->stats.kernel_ms > g->stats.kernel_ms) g->stats.kernel_ms = s->stats.kernel_ms;
        if (s->stats.reduce_ms > g->stats.reduce_ms) g->stats.reduce_ms = s->stats.reduce_ms;
        g->stats.samples += s->stats.samples; g->stats.segments += s->stats.segments; g->stats.launches += s->stats.launches;
        float ms = 0;
        HIP_TRY(hipSetDevice(s->device));
        // a sub-context without rows of its own (a frame of fewer row blocks than devices) or a call with n_samples == 0
        // still joined the gather, but exec_finish did not synchronise its stream: wait for its closing event here, or
        // hipEventElapsedTime answers hipErrorNotReady after the counts have been advanced
        HIP_TRY(hipEventSynchronize(s->ev_g1));
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_g0, s->ev_g1));
        if (ms > gather_ms) gather_ms = ms;
    }
    HIP_TRY(hipSetDevice(g->device));
    { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, g->ev0, g->ev1)); gather_ms += ms; }      // + placing the rows into the frame
    g->stats.k_split = g->subs[0]->stats.k_split;
    g->stats.gather_ms = gather_ms;
    return MRT_OK;
}

// ---- look-ahead of the eager per-call path -----------------------------------------------------------------------------------
// The reference's callers run ONE Sampler::execute per sample and wait for it (src/cli.rs:162-170, src/http.rs:141-144).  A
// one-sample launch cannot regenerate paths inside a lane (a wavefront lasts as long as its longest path: lane utilisation
// 0.65 against 0.70, 0.36 ms per 1080p pass against 0.25 inside a batched launch), and the image is a pure function of (scene,
// seed, sample index) -- so a context that sees one-sample calls arrive back to back traces AHEAD: samples [k, k + n) in one
// launch on a second stream, every sample into a plane of its own (Params.to_planes), and each call
// only folds its sample's plane into the accumulator (reduce_chunks, 0.02 ms) and waits for that.  The accumulator holds
// exactly the samples the caller has asked for at every return, added one by one in index order -- bit for bit what the plain
// per-call loop leaves there -- so observing it (mrt_accum, mrt_img, a bound or handed-out pointer) needs no special case.  Two
// plane sets: while one is folded call by call, the next launch already runs.  It starts at the third consecutive
// one-sample call with two samples per launch and doubles up to la_max (32), so a caller that stops early wastes at most about
// the work it has used; any other call (n != 1, reset, set_accum) drops what was traced ahead.
static void la_drop(mrt_ctx *c)
{
    if (c->la_stream && (c->la_n[0] || c->la_n[1])) (void)hipStreamSynchronize(c->la_stream);
    c->la_n[0] = c->la_n[1] = 0;
    c->la_streak = 0;
}

// launch the trace of samples [base, base + n) into plane set i (asynchronous, on la_stream)
static int la_launch(mrt_ctx *c, int i, u32 base, u32 n)
{
    const size_t plane = (size_t)c->padded_rows * c->pk.nw * 3;
    if (!c->la_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->la_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) { HIP_TRY(hipEventCreate(&c->la_ev0[k])); HIP_TRY(hipEventCreate(&c->la_ev1[k])); }
        HIP_TRY(hipMalloc((void **)&c->d_la_counter, sizeof(u32)));
    }
    if (plane * n > c->la_floats[i]) {
        if (c->d_la[i]) { (void)hipFree(c->d_la[i]); c->d_la[i] = nullptr; c->la_floats[i] = 0; }
        if (!hip_tolerated(hipMalloc((void **)&c->d_la[i], plane * n * sizeof(float)))) { c->d_la[i] = nullptr; return MRT_ERR_LIMIT; }     // the caller falls back
        c->la_floats[i] = plane * n;
    }
    Params P = c->P;
    P.n_samples = n; P.sample_base = base; P.k_split = 1u; P.to_planes = 1u;
    P.partial = c->d_la[i]; P.partial_stride = plane;
    P.count_segments = 0u;
    // small scenes: the plain grid (their persistent grid fills every wave slot of the chip and would keep the folds out until
    // the launch has ended); larger ones leave slots free and keep their persistent workgroups, with a tile counter of their own
    P.persist_grid = c->small_plain_grid ? 0u : c->persist_grid;
    P.tile_counter = c->d_la_counter;
    if (c->block_threads > 64u && P.persist_grid) HIP_TRY(hipMemsetAsync(c->d_la_counter, 0, sizeof(u32), c->la_stream));
    HIP_TRY(hipEventRecord(c->la_ev0[i], c->la_stream));
    HIP_TRY(launch_pt(P, c->block_threads, c->scene_in_lds, c->pk.features, c->la_stream));
    HIP_TRY(hipEventRecord(c->la_ev1[i], c->la_stream));
    c->la_base[i] = base; c->la_n[i] = n;
    return MRT_OK;
}

// one eager one-sample call served from the look-ahead planes; returns MRT_ERR_LIMIT when the planes cannot be had (the caller
// then runs the plain launch)
static int run_lookahead(mrt_ctx *c)
{
    int rc = set_device(c);
    if (rc) return rc;
    const u32 k = c->count;
    int s = -1;
    for (int i = 0; i < 2; ++i) if (c->la_n[i] && k >= c->la_base[i] && k < c->la_base[i] + c->la_n[i]) s = i;
    if (s < 0) {
        // nothing traced ahead for this sample (first use, or what was ahead has been dropped): start both sets
        la_drop(c);
        c->la_streak = 2;
        const u32 n0 = 2u < c->la_max ? 2u : c->la_max;
        if ((rc = la_launch(c, 0, k, n0))) return rc;
        const u32 n1 = 2u * n0 < c->la_max ? 2u * n0 : c->la_max;
        if ((rc = la_launch(c, 1, k + n0, n1))) { (void)hipStreamSynchronize(c->la_stream); c->la_n[0] = c->la_n[1] = 0; return rc; }
        s = 0;
    }
    const size_t plane = (size_t)c->padded_rows * c->pk.nw * 3;
    const size_t words = (size_t)c->local_rows * c->pk.nw * 3;
    HIP_TRY(hipStreamWaitEvent(c->stream, c->la_ev1[s], 0));
    HIP_TRY(launch_reduce_chunks(c->d_accum, c->d_la[s] + (size_t)(k - c->la_base[s]) * plane, words, plane, 1u, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                // the fold has run, so the set's launch has ended too
    c->count += 1u;
    c->stats.kernel_ms = 0; c->stats.reduce_ms = 0; c->stats.gather_ms = 0; c->stats.segments = 0;
    c->stats_pending = false; c->ev_used = 0;
    c->stats.launches = 1u; c->stats.k_split = 1u;
    c->stats.samples = (uint64_t)c->local_rows * c->pk.nw;
    c->stats.block_threads = c->block_threads;
    if (c->event_timing) {
        float ms = 0;
        if (hip_tolerated(hipEventElapsedTime(&ms, c->la_ev0[s], c->la_ev1[s]))) c->stats.kernel_ms = (double)ms / (double)c->la_n[s];   // this sample's share of its launch
    }
    if (k + 1u == c->la_base[s] + c->la_n[s]) {
        // the set is spent (its last plane has been folded): the next launch goes into it, behind the other set's, twice as long
        const int o = 1 - s;
        const u32 nb = c->la_base[o] + c->la_n[o];
        const u32 nn = 2u * c->la_n[o] < c->la_max ? 2u * c->la_n[o] : c->la_max;
        c->la_n[s] = 0;
        if (c->la_n[o] && (unsigned long long)nb + nn <= 0xffffffffull) (void)la_launch(c, s, nb, nn);      // (a failure here only means a later call starts over)
    }
    return MRT_OK;
}

static int run_samples(mrt_ctx *c, uint32_t n_samples)
{
    int rc;
    if (!c->subs.empty()) return exec_group(c, n_samples);
    if (c->la_enabled && n_samples == 1u && c->local_rows) {
        if (c->la_streak >= 2u) {
            rc = run_lookahead(c);
            if (rc != MRT_ERR_LIMIT) return rc;
            c->la_enabled = false;                       // no memory for the planes: the plain per-call launch from here on
        } else {
            ++c->la_streak;
        }
    } else if (c->la_n[0] || c->la_n[1] || c->la_streak) {
        la_drop(c);
    }
    if ((rc = exec_launch(c, n_samples))) return rc;
    return exec_finish(c, n_samples);
}

// Deferred execution (MRT_FLAG_DEFER): trace what earlier mrt_execute calls only booked.  Called by every entry point that
// observes or replaces the accumulator; the result is the same set of samples as if each call had run at once.
static int settle(mrt_ctx *c)
{
    if (!c->pending) return MRT_OK;
    const u32 n = c->pending;
    c->pending = 0;
    return run_samples(c, n);
}
constexpr u32 kDeferLimit = 1024u;        // booked samples that trigger a launch by themselves (one full-size batch)

int mrt_execute(mrt_ctx *c, uint32_t n_samples, double *seconds)
{
    if (!c) return fail(MRT_ERR_ARG, "mrt_execute: null context");
    if ((unsigned long long)c->count + c->pending + n_samples > 0xffffffffull) return fail(MRT_ERR_LIMIT, "mrt_execute: sample count overflows u32");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = MRT_OK;
    const bool deferred = c->defer && !c->exposed();
    if (deferred) {
        c->pending += n_samples;
        if (c->pending >= kDeferLimit) rc = settle(c);
    } else {
        rc = run_samples(c, n_samples);
    }
    if (rc) return rc;
    c->stats.deferred = deferred ? 1u : 0u;       // whether THIS call was booked (MRT_FLAG_DEFER honoured) or ran at once
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ok();
    return MRT_OK;
}

int mrt_dims(const mrt_ctx *c, uint32_t *nw, uint32_t *nh, uint32_t *local_rows)
{
    if (!c) return fail(MRT_ERR_ARG, "mrt_dims: null context");
    if (nw) *nw = c->pk.nw;
    if (nh) *nh = c->pk.nh;
    if (local_rows) *local_rows = c->local_rows;
    ok();
    return MRT_OK;
}

int mrt_accum_local(mrt_ctx *c, float *rgb, uint32_t *rows)
{
    if (!c) return fail(MRT_ERR_ARG, "mrt_accum_local: null context");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = settle(c))) return rc;
    if (!c->subs.empty()) {       // a multi-device context owns every row
        if (rows) memcpy(rows, c->row_of.data(), sizeof(u32) * c->local_rows);
        if (rgb) HIP_TRY(hipMemcpy(rgb, c->d_full, (size_t)c->pk.nh * c->pk.nw * 3 * sizeof(float), hipMemcpyDeviceToHost));
        ok();
        return MRT_OK;
    }
    if (rows) memcpy(rows, c->row_of.data(), sizeof(u32) * c->local_rows);
    if (rgb && c->local_rows) HIP_TRY(hipMemcpy(rgb, c->d_accum, (size_t)c->local_rows * c->pk.nw * 3 * sizeof(float), hipMemcpyDeviceToHost));
    ok();
    return MRT_OK;
}

int mrt_accum(mrt_ctx *c, float *rgb, uint32_t *count)
{
    if (!c) return fail(MRT_ERR_ARG, "mrt_accum: null context");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = settle(c))) return rc;
    const size_t row_bytes = (size_t)c->pk.nw * 3 * sizeof(float);
    if (rgb) {
        if (c->d_full) {
            HIP_TRY(hipMemcpy(rgb, c->d_full, row_bytes * c->pk.nh, hipMemcpyDeviceToHost));
        } else if (c->shard_count == 1) {
            HIP_TRY(hipMemcpy(rgb, c->d_accum, row_bytes * c->pk.nh, hipMemcpyDeviceToHost));
        } else {
            std::vector<float> tmp((size_t)c->local_rows * c->pk.nw * 3);
            if (c->local_rows) HIP_TRY(hipMemcpy(tmp.data(), c->d_accum, row_bytes * c->local_rows, hipMemcpyDeviceToHost));
            for (u32 r = 0; r < c->local_rows; ++r) memcpy((char *)rgb + row_bytes * c->row_of[r], (char *)tmp.data() + row_bytes * r, row_bytes);
        }
    }
    if (count) *count = c->d_full ? c->full_count : c->count;
    ok();
    return MRT_OK;
}

int mrt_accum_device_ptr(mrt_ctx *c, void **dev_ptr, size_t *bytes)
{
    if (!c) return fail(MRT_ERR_ARG, "mrt_accum_device_ptr: null context");
    { int rc = set_device(c); if (rc) return rc; if ((rc = settle(c))) return rc; }
    c->handed_out = true;                     // from now on the caller may read the accumulator behind the library's back (sticky)
    if (dev_ptr) *dev_ptr = c->subs.empty() ? c->d_accum : c->d_full;
    if (bytes) *bytes = (size_t)c->padded_rows * c->pk.nw * 3 * sizeof(float);
    ok();
    return MRT_OK;
}

int mrt_padded_rows(const mrt_ctx *c, uint32_t *rows)
{
    if (!c || !rows) return fail(MRT_ERR_ARG, "mrt_padded_rows: null argument");
    *rows = c->padded_rows;
    ok();
    return MRT_OK;
}

int mrt_bind_accum(mrt_ctx *c, void *dev_ptr, size_t bytes)
{
    if (!c) return fail(MRT_ERR_ARG, "mrt_bind_accum: null context");
    if (!c->subs.empty()) return fail(MRT_ERR_STATE, "mrt_bind_accum: not available on a multi-device context");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = settle(c))) return rc;
    const size_t need = (size_t)c->padded_rows * c->pk.nw * 3 * sizeof(float);
    float *dst = dev_ptr ? (float *)dev_ptr : c->d_accum_own;
    if (dev_ptr && bytes < need) return fail(MRT_ERR_ARG, "mrt_bind_accum: buffer of %zu bytes, need %zu", bytes, need);     // nothing changed
    if (dst != c->d_accum) {
        HIP_TRY(hipMemcpy(dst, c->d_accum, need, hipMemcpyDeviceToDevice));
        c->d_accum = dst;
        c->P.accum = dst;
    }
    c->bound = dev_ptr != nullptr;            // only a bind that succeeded: the caller reads this memory whenever it likes, so every execute runs at once
    ok();
    return MRT_OK;
}

int mrt_set_accum_device(mrt_ctx *c, const void *dev_rgb, uint32_t count)
{
    if (!c || !dev_rgb) return fail(MRT_ERR_ARG, "mrt_set_accum_device: null argument");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = settle(c))) return rc;
    la_drop(c);                                   // the sample count changes under what was traced ahead
    const size_t bytes = (size_t)c->pk.nw * c->pk.nh * 3 * sizeof(float);
    if (!c->subs.empty()) return fail(MRT_ERR_STATE, "mrt_set_accum_device: use mrt_set_accum on a multi-device context");
    if (c->shard_count == 1) {
        HIP_TRY(hipMemcpy(c->d_accum, dev_rgb, bytes, hipMemcpyDeviceToDevice));
        c->count = count;
    } else {
        if (!c->d_full) HIP_TRY(hipMalloc((void **)&c->d_full, bytes));
        HIP_TRY(hipMemcpy(c->d_full, dev_rgb, bytes, hipMemcpyDeviceToDevice));
        c->full_count = count;
    }
    ok();
    return MRT_OK;
}

int mrt_set_accum(mrt_ctx *c, const float *rgb, uint32_t count)
{
    if (!c || !rgb) return fail(MRT_ERR_ARG, "mrt_set_accum: null argument");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = settle(c))) return rc;
    la_drop(c);
    const size_t row_bytes = (size_t)c->pk.nw * 3 * sizeof(float);
    if (!c->subs.empty()) {
        HIP_TRY(hipMemcpy(c->d_full, rgb, row_bytes * c->pk.nh, hipMemcpyHostToDevice));
        for (mrt_ctx *s : c->subs) {
            HIP_TRY(hipSetDevice(s->device));
            for (u32 i = 0; i < s->local_rows; ++i)
                HIP_TRY(hipMemcpy((char *)s->d_accum + row_bytes * i, (const char *)rgb + row_bytes * s->row_of[i], row_bytes, hipMemcpyHostToDevice));
            s->count = count;
        }
        c->count = count; c->full_count = count;
    } else if (c->shard_count == 1) {
        HIP_TRY(hipMemcpy(c->d_accum, rgb, row_bytes * c->pk.nh, hipMemcpyHostToDevice));
        c->count = count;
    } else {
        // a sharded context keeps the gathered frame next to its own rows (it is only read by mrt_img / mrt_accum)
        if (!c->d_full) HIP_TRY(hipMalloc((void **)&c->d_full, row_bytes * c->pk.nh));
        HIP_TRY(hipMemcpy(c->d_full, rgb, row_bytes * c->pk.nh, hipMemcpyHostToDevice));
        c->full_count = count;
    }
    ok();
    return MRT_OK;
}

int mrt_reset(mrt_ctx *c)
{
    if (!c) return fail(MRT_ERR_ARG, "mrt_reset: null context");
    c->pending = 0;                           // booked samples of a deferred context are dropped with everything else
    int rc = set_device(c);
    if (rc) return rc;
    la_drop(c);
    if (!c->subs.empty()) {
        for (mrt_ctx *s : c->subs) if ((rc = mrt_reset(s))) return rc;
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipMemset(c->d_full, 0, (size_t)c->pk.nh * c->pk.nw * 3 * sizeof(float)));
        c->count = 0; c->full_count = 0;
        ok();
        return MRT_OK;
    }
    HIP_TRY(hipMemset(c->d_accum, 0, (size_t)c->padded_rows * c->pk.nw * 3 * sizeof(float)));
    if (c->d_full) { (void)hipFree(c->d_full); c->d_full = nullptr; }
    c->count = 0; c->full_count = 0;
    ok();
    return MRT_OK;
}

static int img_prepare(mrt_ctx *c)
{
    const u32 nw = c->pk.nw, nh = c->pk.nh, rw = c->pk.res_w, rh = c->pk.res_h;
    if (!c->d_ss) HIP_TRY(hipMalloc((void **)&c->d_ss, (size_t)nw * nh * 3));
    if (rw == nw && rh == nh) return MRT_OK;
    if (rw == 0 || rh == 0) return fail(MRT_ERR_SCENE, "mrt_img: zero output resolution");
    if (!c->d_out) {
        // everything is allocated and filled through locals and committed to the context only when all of it succeeded: a
        // failure half-way leaves the context as it was (d_out still null), so the next mrt_img starts over instead of
        // allocating over live pointers
        ResampleTaps v, h;
        lanczos3_taps(nh, rh, v);
        lanczos3_taps(nw, rw, h);
        u32 *vl = nullptr, *vc = nullptr, *hl = nullptr, *hc = nullptr;
        float *vw = nullptr, *hw = nullptr, *tmp = nullptr;
        unsigned char *out = nullptr;
        auto build = [&]() -> int {
            HIP_TRY(hipMalloc((void **)&vl, sizeof(u32) * rh));
            HIP_TRY(hipMalloc((void **)&vc, sizeof(u32) * rh));
            HIP_TRY(hipMalloc((void **)&vw, sizeof(float) * (size_t)rh * v.cap));
            HIP_TRY(hipMalloc((void **)&hl, sizeof(u32) * rw));
            HIP_TRY(hipMalloc((void **)&hc, sizeof(u32) * rw));
            HIP_TRY(hipMalloc((void **)&hw, sizeof(float) * (size_t)rw * h.cap));
            HIP_TRY(hipMemcpy(vl, v.left.data(), sizeof(u32) * rh, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(vc, v.count.data(), sizeof(u32) * rh, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(vw, v.weight.data(), sizeof(float) * (size_t)rh * v.cap, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(hl, h.left.data(), sizeof(u32) * rw, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(hc, h.count.data(), sizeof(u32) * rw, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(hw, h.weight.data(), sizeof(float) * (size_t)rw * h.cap, hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc((void **)&tmp, sizeof(float) * (size_t)nw * rh * 3));
            HIP_TRY(hipMalloc((void **)&out, (size_t)rw * rh * 3));
            return MRT_OK;
        };
        const int rc = build();
        if (rc) {
            void *ptrs[] = {vl, vc, vw, hl, hc, hw, tmp, out};
            for (void *q : ptrs) if (q) (void)hipFree(q);
            (void)hipGetLastError();
            return rc;
        }
        c->vcap = v.cap; c->hcap = h.cap;
        c->d_vl = vl; c->d_vc = vc; c->d_vw = vw; c->d_hl = hl; c->d_hc = hc; c->d_hw = hw; c->d_tmp = tmp; c->d_out = out;
    }
    return MRT_OK;
}

static int img_tonemap(mrt_ctx *c)
{
    const float *src = c->d_full ? c->d_full : c->d_accum;
    const u32 count = c->d_full ? c->full_count : c->count;
    if (!c->d_full && c->shard_count != 1) return fail(MRT_ERR_STATE, "mrt_img: this context holds only its own rows; gather and mrt_set_accum first");
    if (count == 0) return fail(MRT_ERR_STATE, "mrt_img: no samples accumulated (the reference would panic on an empty map, src/sampler.rs:85)");
    const float rc = 1.0f / (float)count;
    const float wexp = (1.0f - c->pk.exp) * (1.0f - c->pk.exp);
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(launch_tonemap(src, c->d_ss, c->pk.nw * c->pk.nh, rc, c->pk.gamma, wexp, c->stream));
    return MRT_OK;
}

int mrt_img_ss(mrt_ctx *c, uint8_t *rgb8)
{
    if (!c || !rgb8) return fail(MRT_ERR_ARG, "mrt_img_ss: null argument");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = settle(c))) return rc;
    if ((rc = img_prepare(c))) return rc;
    if ((rc = img_tonemap(c))) return rc;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1)); c->stats.img_ms = ms; }
    HIP_TRY(hipMemcpy(rgb8, c->d_ss, (size_t)c->pk.nw * c->pk.nh * 3, hipMemcpyDeviceToHost));
    ok();
    return MRT_OK;
}

int mrt_img(mrt_ctx *c, uint8_t *rgb8)
{
    if (!c || !rgb8) return fail(MRT_ERR_ARG, "mrt_img: null argument");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = settle(c))) return rc;
    if ((rc = img_prepare(c))) return rc;
    if ((rc = img_tonemap(c))) return rc;
    const u32 nw = c->pk.nw, nh = c->pk.nh, rw = c->pk.res_w, rh = c->pk.res_h;
    if (rw == nw && rh == nh) {    // image 0.24 resize copies when the dimensions match
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1)); c->stats.img_ms = ms; }
        HIP_TRY(hipMemcpy(rgb8, c->d_ss, (size_t)nw * nh * 3, hipMemcpyDeviceToHost));
        ok();
        return MRT_OK;
    }
    HIP_TRY(launch_lanczos_v(c->d_ss, c->d_tmp, nw, rh, c->d_vl, c->d_vc, c->d_vw, c->vcap, c->stream));
    HIP_TRY(launch_lanczos_h(c->d_tmp, c->d_out, nw, rw, rh, c->d_hl, c->d_hc, c->d_hw, c->hcap, c->stream));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1)); c->stats.img_ms = ms; }
    HIP_TRY(hipMemcpy(rgb8, c->d_out, (size_t)rw * rh * 3, hipMemcpyDeviceToHost));
    ok();
    return MRT_OK;
}

int mrt_get_stats(mrt_ctx *c, mrt_stats *out)
{
    if (!c || !out) return fail(MRT_ERR_ARG, "mrt_get_stats: null argument");
    if (c->pending) { int rc = set_device(c); if (rc) return rc; if ((rc = settle(c))) return rc; }      // an observation: booked samples are traced first
    if (c->stats_pending) { const int rc = resolve_stats(c); if (rc) return rc; }
    *out = c->stats;
    ok();
    return MRT_OK;
}

int mrt_plan_launch(const mrt_render_desc *desc, mrt_plan *out)
{
    if (!desc || !out) return fail(MRT_ERR_ARG, "mrt_plan_launch: null argument");
    Packed pk;
    std::string err;
    const int rc = pack_scene(desc, pk, err);
    if (rc != MRT_OK) return fail(rc, "mrt_plan_launch: %s", err.c_str());
    const u32 n_nodes = pk.n_tbvh_nodes;
    Plan pl;
    plan_launch(desc, pk, pl);
    memset(out, 0, sizeof *out);
    out->staging = !pl.in_lds ? 3u : ((pk.features & 128u) ? 2u : ((pk.features & 64u) ? 1u : 0u));
    out->block_threads = pl.block_threads;
    out->lds_bytes = (uint32_t)pt_lds_bytes(pk.P, pl.block_threads, pl.in_lds, pk.features);
    out->staged_bytes = pl.in_lds ? (uint32_t)pl.staged_bytes : 0u;
    out->scene_bytes = pk.P.lds_words * 4u;
    out->kernel_features = pt_instantiation(pl.block_threads, pl.in_lds, pk.features);
    out->tbvh_nodes = n_nodes;
    out->tbvh_hot_nodes = !pl.in_lds ? 0u : ((pk.features & 128u) ? pk.P.n_tbvh_hot : n_nodes);
    out->small_plain_grid = pl.small_plain_grid ? 1u : 0u;
    out->walk_cap = pk.P.walk_cap;
    ok();
    return MRT_OK;
}

int mrt_selftest_math(int device, int op, const float *a, const float *b, float *out, size_t n)
{
    if (!a || !out) return fail(MRT_ERR_ARG, "mrt_selftest_math: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MRT_ERR_DEVICE, "mrt_selftest_math: no HIP device");
    if (device < 0) device = 0;
    HIP_TRY(hipSetDevice(device));
    if (n == 0) { ok(); return MRT_OK; }
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    const size_t bytes = n * sizeof(float);
    auto run = [&]() -> int {                 // every exit goes through the frees below
        HIP_TRY(hipMalloc((void **)&da, bytes));
        HIP_TRY(hipMalloc((void **)&dout, bytes));
        HIP_TRY(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
        if (b) { HIP_TRY(hipMalloc((void **)&db, bytes)); HIP_TRY(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice)); }
        HIP_TRY(launch_math_selftest(op, da, db, dout, n, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
        return MRT_OK;
    };
    const int rc = run();
    if (da) (void)hipFree(da);
    if (dout) (void)hipFree(dout);
    if (db) (void)hipFree(db);
    if (rc) return rc;
    ok();
    return MRT_OK;
}

int mrt_selftest_sweep(int device, int op, uint64_t first, uint64_t count, uint32_t seed, uint64_t *mismatches, float *example)
{
    if (!mismatches) return fail(MRT_ERR_ARG, "mrt_selftest_sweep: null argument");
    if (op < 0 || op > 3) return fail(MRT_ERR_ARG, "mrt_selftest_sweep: op %d", op);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MRT_ERR_DEVICE, "mrt_selftest_sweep: no HIP device");
    if (device < 0) device = 0;
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d_mis = nullptr;
    float *d_ex = nullptr;
    unsigned long long mis = 0;
    float ex[4] = {0, 0, 0, 0};
    auto run = [&]() -> int {
        HIP_TRY(hipMalloc((void **)&d_mis, sizeof(unsigned long long)));
        HIP_TRY(hipMalloc((void **)&d_ex, 4 * sizeof(float)));
        HIP_TRY(hipMemset(d_mis, 0, sizeof(unsigned long long)));
        HIP_TRY(hipMemset(d_ex, 0, 4 * sizeof(float)));
        const uint64_t slice = 1ull << 28;                 // one launch per 2^28 elements
        for (uint64_t done = 0; done < count; done += slice) {
            const uint64_t n = count - done < slice ? count - done : slice;
            HIP_TRY(launch_math_sweep(op, first + done, n, seed, d_mis, d_ex, nullptr));
        }
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&mis, d_mis, sizeof mis, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ex, d_ex, sizeof ex, hipMemcpyDeviceToHost));
        return MRT_OK;
    };
    const int rc = run();
    if (d_mis) (void)hipFree(d_mis);
    if (d_ex) (void)hipFree(d_ex);
    if (rc) return rc;
    *mismatches = mis;
    if (example) memcpy(example, ex, sizeof ex);
    ok();
    return MRT_OK;
}

}  // extern "C"
