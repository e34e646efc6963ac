// sampler.hpp — header-only C++ mirror of the reference's `Sampler` (src/sampler.rs:11-100) over the C ABI
// of include/mrt.h.  Same three entry points and error behaviour (std::runtime_error <-> Err(String)).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mrt.h"

namespace mrt {

class Sampler {
public:
    // Sampler::new(workers, n_dim), src/sampler.rs:19.  Thread-pool size and tile grid have no GPU meaning.
    // flags: MRT_FLAG_*; the default is what the Rust shim passes (shim/rust/sampler_hip.rs): no event timing, and
    // deferred execution -- a per-sample execute() only books its sample, img() / colors() trace what is booked in
    // 1024-sample batches (same samples, same image; the returned Duration of a booking call is ~0).
    // MRT_DEFER=0 in the environment keeps the default eager (real per-sample Durations), as in the shim.
    static uint32_t default_flags()
    {
        const char *v = std::getenv("MRT_DEFER");
        const bool eager = v && v[0] == '0' && v[1] == '\0';
        return MRT_FLAG_NO_EVENT_TIMING | (eager ? 0u : MRT_FLAG_DEFER);
    }
    explicit Sampler(uint32_t /*workers*/ = 24, size_t /*n_dim*/ = 64, uint64_t seed = 1, int device = -1,
                     uint32_t flags = default_flags())
        : seed_(seed), device_(device), flags_(flags) {}
    Sampler(const Sampler &) = delete;
    Sampler &operator=(const Sampler &) = delete;
    ~Sampler() { if (ctx_) mrt_destroy(ctx_); }

    // Sampler::execute(&mut self, scene, frame, rt) -> Duration, src/sampler.rs:28: one sample pass.
    // The context is created on the first call (scene and frame only arrive here) and rebuilt when a later call passes
    // a description with other contents; sums accumulated so far are carried over when the supersampled frame keeps its
    // size (the reference's map keeps adding whatever the scene, src/sampler.rs:60-70).
    double execute(const mrt_render_desc &render, uint32_t n_samples = 1)
    {
        const uint64_t print = fingerprint(render);
        if (!ctx_ || print != print_ || stale_) {
            stale_ = false;
            mrt_opts o{};
            o.abi_version = MRT_ABI_VERSION;
            o.seed = seed_;
            o.device = device_;
            o.flags = flags_;
            mrt_ctx *fresh = mrt_create(&render, &o);
            if (!fresh) throw std::runtime_error(mrt_last_error());
            uint32_t nw = 0, nh = 0;
            mrt_dims(fresh, &nw, &nh, nullptr);
            if (ctx_) {
                if (nw == nw_ && nh == nh_) {
                    uint32_t count = 0;
                    std::vector<float> sums((size_t)nw * nh * 3);
                    if (mrt_accum(ctx_, sums.data(), &count) != MRT_OK || (count && mrt_set_accum(fresh, sums.data(), count) != MRT_OK)) {
                        const std::string why = mrt_last_error();
                        mrt_destroy(fresh);
                        throw std::runtime_error(why);
                    }
                }
                mrt_destroy(ctx_);
            }
            ctx_ = fresh; print_ = print; nw_ = nw; nh_ = nh; ++created_;
            res_w_ = render.frame.res_w; res_h_ = render.frame.res_h;
        }
        double secs = 0;
        if (mrt_execute(ctx_, n_samples, &secs) != MRT_OK) throw std::runtime_error(mrt_last_error());
        return secs;
    }

    // Sampler::img(&self, frame) -> Result<RgbImage, String>, src/sampler.rs:80: rgb8[res_h][res_w][3]
    std::vector<uint8_t> img() const
    {
        if (!ctx_) throw std::runtime_error("img before execute");
        std::vector<uint8_t> out((size_t)res_w_ * res_h_ * 3);
        if (mrt_img(ctx_, out.data()) != MRT_OK) throw std::runtime_error(mrt_last_error());
        return out;
    }

    // colors / last_count, src/sampler.rs:14-15
    std::vector<float> colors(uint32_t *count = nullptr) const
    {
        if (!ctx_) throw std::runtime_error("colors before execute");
        std::vector<float> out((size_t)nw_ * nh_ * 3);
        if (mrt_accum(ctx_, out.data(), count) != MRT_OK) throw std::runtime_error(mrt_last_error());
        return out;
    }

    mrt_stats stats() const
    {
        mrt_stats st{};
        if (!ctx_ || mrt_get_stats(ctx_, &st) != MRT_OK) throw std::runtime_error(ctx_ ? mrt_last_error() : "stats before execute");
        return st;
    }

    // The fingerprint hashes bulk data (triangles, texels) by address, length and a strided sample of its words: an edit IN
    // PLACE that falls between the sample points is not seen.  A caller that edits bulk arrays in place says so here; the
    // next execute() rebuilds the context from the description it is given (sums carried over as for any other change).
    void invalidate() { stale_ = true; }

    uint32_t width() const { return res_w_; }
    uint32_t height() const { return res_h_; }
    uint32_t contexts_created() const { return created_; }

private:
    // FNV-1a over what the context is built from: every scalar and instance by value; bulk arrays (triangles, texels) by
    // address, size and a strided sample of their words.
    struct Fnv {
        uint64_t h = 1469598103934665603ull;
        void bytes(const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } }
        template <class T> void pod(const T &v) { bytes(&v, sizeof v); }
        void bulk(const float *p, size_t n) { pod(p); pod(n); if (!p) return; const size_t step = 1 + n / 1024; for (size_t i = 0; i < n; i += step) pod(p[i]); }
    };
    uint64_t fingerprint(const mrt_render_desc &d)
    {
        Fnv f;
        f.pod(d.rt.bounce); f.pod(d.rt.loss);
        f.pod(d.frame.res_w); f.pod(d.frame.res_h); f.pod(d.frame.ssaa); f.pod(d.frame.cam.pos); f.pod(d.frame.cam.dir);
        f.pod(d.frame.cam.fov); f.pod(d.frame.cam.gamma); f.pod(d.frame.cam.exp); f.pod(d.frame.cam.aprt); f.pod(d.frame.cam.foc);
        f.pod(d.scene.sky.color); f.pod(d.scene.sky.pwr);
        f.pod(d.scene.n_renderer); f.pod(d.scene.n_light); f.pod(d.scene.n_textures);
        for (uint32_t i = 0; i < d.scene.n_renderer; ++i) {
            const mrt_renderer &r = d.scene.renderer[i];
            f.pod(r.kind); f.pod(r.param); f.pod(r.n_tris); f.pod(r.n_inst);
            f.pod(r.mat.albedo); f.pod(r.mat.rough); f.pod(r.mat.metal); f.pod(r.mat.glass); f.pod(r.mat.opacity); f.pod(r.mat.emit);
            f.pod(r.mat.tex); f.pod(r.mat.rmap); f.pod(r.mat.mmap); f.pod(r.mat.gmap); f.pod(r.mat.omap); f.pod(r.mat.emap);
            if (r.kind == MRT_KIND_MESH) f.bulk(r.tris, (size_t)r.n_tris * 9);
            for (uint32_t k = 0; k < r.n_inst; ++k) { f.pod(r.inst[k].pos); f.pod(r.inst[k].dir); }
        }
        for (uint32_t i = 0; i < d.scene.n_light; ++i) { const mrt_light &l = d.scene.light[i]; f.pod(l.kind); f.pod(l.v); f.pod(l.pwr); f.pod(l.color); }
        for (uint32_t i = 0; i < d.scene.n_textures; ++i) { const mrt_texture &t = d.scene.textures[i]; f.pod(t.w); f.pod(t.h); f.bulk(t.dat, t.dat ? (size_t)t.w * t.h * 3 : 0); }
        return f.h;
    }

    mrt_ctx *ctx_ = nullptr;
    uint64_t seed_;
    int device_;
    uint32_t flags_;
    uint64_t print_ = 0;
    uint32_t nw_ = 0, nh_ = 0, res_w_ = 0, res_h_ = 0;
    uint32_t created_ = 0;
    bool stale_ = false;
};

// CLI::raytrace / HttpServer::raytrace, src/cli.rs:155-177, src/http.rs:136-148
template <class OnUpdate>
inline std::vector<uint8_t> raytrace(const mrt_render_desc &render, bool update, OnUpdate on_update, uint64_t seed = 1)
{
    Sampler s(24, 64, seed);
    if (update) {
        for (uint32_t i = 0; i < render.rt.sample; ++i) { s.execute(render, 1); on_update(i, s.img()); }
    } else {
        s.execute(render, render.rt.sample);     // no per-sample image needed: one launch for all passes
    }
    return s.img();
}

}  // namespace mrt
