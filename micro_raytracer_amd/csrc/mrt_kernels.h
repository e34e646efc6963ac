// mrt_kernels.h — host-callable launchers of the kernels in mrt_kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include "mrt_scene.h"

namespace mrt {

hipError_t configure_pt(size_t max_lds_bytes);
u32 pt_instantiation(u32 block_threads, bool scene_in_lds, u32 features);     // FEAT template argument of the kernel launch_pt picks
size_t pt_lds_bytes(const Params &P, u32 block_threads, bool scene_in_lds, u32 features);
hipError_t launch_pt(const Params &P, u32 block_threads, bool scene_in_lds, u32 features, hipStream_t stream);
hipError_t launch_reduce_chunks(float *accum, const float *partial, size_t n_words, size_t stride, u32 n_chunks, hipStream_t stream);
hipError_t launch_scatter_rows(float *frame, const float *gathered, const u32 *rowmap, u32 n_rows, u32 row_words, hipStream_t stream);
hipError_t launch_tonemap(const float *accum, unsigned char *out, u32 n_px, float rc, float gamma, float wexp, hipStream_t stream);
hipError_t launch_lanczos_v(const unsigned char *src, float *dst, u32 sw, u32 dh, const u32 *left, const u32 *count,
                            const float *weight, u32 cap, hipStream_t stream);
hipError_t launch_lanczos_h(const float *src, unsigned char *dst, u32 sw, u32 dw, u32 dh, const u32 *left, const u32 *count,
                            const float *weight, u32 cap, hipStream_t stream);
hipError_t launch_math_selftest(int op, const float *a, const float *b, float *out, size_t n, hipStream_t stream);
hipError_t launch_math_sweep(int op, unsigned long long first, unsigned long long n, u32 seed, unsigned long long *mismatches, float *example, hipStream_t stream);

}  // namespace mrt
