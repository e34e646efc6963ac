// mrt_post.h — per-element bodies of Sampler::img (reference src/sampler.rs:80-99): tone map and the
// two passes of image 0.24's Lanczos3 resize.  Shared by the kernels in mrt_kernels.hip.
#pragma once
#include "mrt_math.h"

namespace mrt {

// `f32 as u8`: saturating, NaN -> 0
MRT_HD unsigned char f32_to_u8(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (unsigned char)(u32)v;
}

// one channel of src/sampler.rs:85-94: sum * (1/count), ^gamma, extended Reinhard, * 255, truncate
MRT_HD unsigned char tonemap_channel(float sum, float rc, float gamma, float wexp)
{
    const float col = sum * rc;                              // Vec3f / f32 == * recip, src/lin.rs:296-301
    const float g = pow_(col, gamma);                        // src/sampler.rs:88
    const float f = g * (1.0f + g / wexp) / (1.0f + g);      // src/sampler.rs:91, wexp = (1 - exp)^2
    return f32_to_u8(255.0f * f);                            // src/sampler.rs:94
}

// horizontal_sample's output conversion: clamp(t, 0, 255), f32::round (half away from zero), to u8
MRT_HD unsigned char resample_to_u8(float t)
{
    const float c = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
    const float fl = floor_(c);
    const float r = (c - fl >= 0.5f) ? fl + 1.0f : fl;      // c >= 0; a NaN falls through to 0
    return f32_to_u8(r);
}

}  // namespace mrt
