// sampler.hpp — header-only C++ mirror of the reference's `Sampler` (src/sampler.rs:11-100) over the C ABI
// of include/mrt.h.  Same three entry points and error behaviour (std::runtime_error <-> Err(String)).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mrt.h"

namespace mrt {

class Sampler {
public:
    // Sampler::new(workers, n_dim), src/sampler.rs:19.  Thread-pool size and tile grid have no GPU meaning.
    Sampler(uint32_t /*workers*/ = 24, size_t /*n_dim*/ = 64, uint64_t seed = 1, int device = -1) : seed_(seed), device_(device) {}
    Sampler(const Sampler &) = delete;
    Sampler &operator=(const Sampler &) = delete;
    ~Sampler() { if (ctx_) mrt_destroy(ctx_); }

    // Sampler::execute(&mut self, scene, frame, rt) -> Duration, src/sampler.rs:28: one sample pass.
    // The context is created on the first call (scene and frame only arrive here), like the Rust shim.
    double execute(const mrt_render_desc &render, uint32_t n_samples = 1)
    {
        if (!ctx_) {
            mrt_opts o{};
            o.abi_version = MRT_ABI_VERSION;
            o.seed = seed_;
            o.device = device_;
            ctx_ = mrt_create(&render, &o);
            if (!ctx_) throw std::runtime_error(mrt_last_error());
            mrt_dims(ctx_, &nw_, &nh_, nullptr);
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

    uint32_t width() const { return res_w_; }
    uint32_t height() const { return res_h_; }

private:
    mrt_ctx *ctx_ = nullptr;
    uint64_t seed_;
    int device_;
    uint32_t nw_ = 0, nh_ = 0, res_w_ = 0, res_h_ = 0;
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
