// harness.cpp — C++ drive of the C ABI through csrc/sampler.hpp, shaped like the reference's two callers:
// CLI::raytrace (src/cli.rs:155-177: per-sample execute, optional --update image, final img) and
// HttpServer::raytrace (src/http.rs:136-148: one Sampler per connection thread, concurrently).
// Prints FNV-1a checksums that tests/test_gpu_native.py compares with the Python path.
#include <stdio.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../micro_raytracer_amd/csrc/sampler.hpp"

static unsigned long long fnv(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// example/Default.json geometry at a small resolution: 1 sphere, 1 point light
static void default_scene(mrt_render_desc &d, mrt_renderer &r, mrt_instance &in, mrt_light &l, uint16_t w, uint16_t h, uint32_t spp)
{
    memset(&d, 0, sizeof d); memset(&r, 0, sizeof r); memset(&in, 0, sizeof in); memset(&l, 0, sizeof l);
    d.rt.bounce = 8; d.rt.sample = spp; d.rt.loss = 0.15f;
    d.frame.res_w = w; d.frame.res_h = h; d.frame.ssaa = 1.0f;
    mrt_camera &c = d.frame.cam;
    c.pos[0] = 0; c.pos[1] = -1; c.pos[2] = 0; c.dir[0] = 0; c.dir[1] = 0; c.dir[2] = 1; c.dir[3] = 0;
    c.fov = 70; c.gamma = 0.8f; c.exp = 0.2f; c.aprt = 0.001f; c.foc = 100;
    r.kind = MRT_KIND_SPHERE; r.param[0] = 0.5f;
    r.mat.albedo[0] = r.mat.albedo[1] = r.mat.albedo[2] = 1; r.mat.opacity = 1;
    r.mat.tex = r.mat.rmap = r.mat.mmap = r.mat.gmap = r.mat.omap = r.mat.emap = -1;
    in.dir[0] = -0.0f; in.dir[1] = -0.0f; in.dir[2] = -1.0f; in.dir[3] = -0.0f;
    r.inst = &in; r.n_inst = 1;
    l.kind = MRT_LIGHT_POINT; l.v[0] = -0.5f; l.v[1] = -1; l.v[2] = 0.5f; l.pwr = 0.5f; l.color[0] = l.color[1] = l.color[2] = 1;
    d.scene.renderer = &r; d.scene.n_renderer = 1; d.scene.light = &l; d.scene.n_light = 1;
    d.scene.sky.pwr = 0.5f;
}

int main()
{
    mrt_render_desc d; mrt_renderer r; mrt_instance in; mrt_light l;
    default_scene(d, r, in, l, 96, 54, 4);
    try {
        // CLI::raytrace with --update: img after every pass
        unsigned updates = 0;
        std::vector<uint8_t> a = mrt::raytrace(d, true, [&](uint32_t, const std::vector<uint8_t> &) { ++updates; }, 7);
        // CLI::raytrace without --update: all passes in one launch
        std::vector<uint8_t> b = mrt::raytrace(d, false, [](uint32_t, const std::vector<uint8_t> &) {}, 7);
        printf("updates %u\n", updates);
        printf("img_update %016llx\nimg_batched %016llx\n", fnv(a.data(), a.size()), fnv(b.data(), b.size()));
        // HttpServer: one Sampler per connection thread
        unsigned long long sums[3] = {0, 0, 0};
        std::vector<std::thread> th;
        for (int t = 0; t < 3; ++t) th.emplace_back([&, t]() {
            mrt::Sampler s(24, 64, 7);
            for (uint32_t i = 0; i < d.rt.sample; ++i) s.execute(d, 1);
            std::vector<uint8_t> im = s.img();
            sums[t] = fnv(im.data(), im.size());
        });
        for (auto &x : th) x.join();
        printf("img_threads %016llx %016llx %016llx\n", sums[0], sums[1], sums[2]);
        // the shim's default (MRT_FLAG_DEFER): per-sample calls are booked, observation traces them; the same loop run
        // eagerly gives the same image; a changed description rebuilds the context and keeps the sums
        {
            mrt::Sampler dfr(24, 64, 7);                                               // default flags: deferred
            mrt::Sampler eag(24, 64, 7, -1, MRT_FLAG_NO_EVENT_TIMING);
            for (uint32_t i = 0; i < d.rt.sample; ++i) { dfr.execute(d, 1); eag.execute(d, 1); }
            const mrt_stats sd = dfr.stats(), se = eag.stats();
            std::vector<uint8_t> x = dfr.img(), y = eag.img();
            printf("deferred %u eager %u same %d\n", sd.deferred, se.deferred, (int)(x == y));
            printf("img_deferred %016llx\n", fnv(x.data(), x.size()));
            uint32_t c0 = 0, c1 = 0;
            dfr.colors(&c0);
            mrt_render_desc d2 = d;
            d2.rt.bounce = 3;                                                           // another rt on the next call
            dfr.execute(d2, 1);
            dfr.colors(&c1);
            printf("rebuild contexts %u count %u -> %u\n", dfr.contexts_created(), c0, c1);
            // an in-place edit of bulk data the strided fingerprint may miss: the caller says so, the next call rebuilds
            uint32_t c2 = 0;
            dfr.invalidate();
            dfr.execute(d2, 1);
            dfr.colors(&c2);
            printf("invalidate contexts %u count %u\n", dfr.contexts_created(), c2);
        }
        // error path: emit outside [0,1] is what gen_bool would panic on (src/rt.rs:968)
        r.mat.emit = 2.0f;
        try { mrt::Sampler s; s.execute(d, 1); printf("error_path none\n"); }
        catch (const std::exception &e) { printf("error_path %s\n", e.what()); }
    } catch (const std::exception &e) {
        printf("FAILED %s\n", e.what());
        return 1;
    }
    return 0;
}
