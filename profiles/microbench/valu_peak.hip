// micro-benchmark: peak VALU issue rate of non-packed f32 mul/add on gfx950 (for the roofline denominator); mode 2: v_pk_fma_f32
// (two f32 FMAs per instruction on an even-aligned register pair), mode 3: v_pk_add_f32; modes 4 / 5 / 6: eight v_fma_f32 with 2 / 4 / 8 s_add_u32 between them (does scalar issue cost vector issue?)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned sc = (unsigned)iters;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a0 + 8, a1 + 8}, p5 = {a2 + 8, a3 + 8}, p6 = {a4 + 8, a5 + 8}, p7 = {a6 + 8, a7 + 8};
    const f2 s2 = {s, s};
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) { a0 = a0 * s; a1 = a1 + s; a2 = a2 * s; a3 = a3 + s; a4 = a4 * s; a5 = a5 + s; a6 = a6 * s; a7 = a7 + s; }
            if (MODE == 2) { p0 = __builtin_elementwise_fma(p0, s2, s2); p1 = __builtin_elementwise_fma(p1, s2, s2); p2 = __builtin_elementwise_fma(p2, s2, s2); p3 = __builtin_elementwise_fma(p3, s2, s2); p4 = __builtin_elementwise_fma(p4, s2, s2); p5 = __builtin_elementwise_fma(p5, s2, s2); p6 = __builtin_elementwise_fma(p6, s2, s2); p7 = __builtin_elementwise_fma(p7, s2, s2); }
            if (MODE == 3) { p0 = p0 + s2; p1 = p1 - s2; p2 = p2 + s2; p3 = p3 - s2; p4 = p4 + s2; p5 = p5 - s2; p6 = p6 + s2; p7 = p7 - s2; }
            if (MODE >= 4) {
                a0 = __builtin_fmaf(a0, s, s); a1 = __builtin_fmaf(a1, s, s); a2 = __builtin_fmaf(a2, s, s); a3 = __builtin_fmaf(a3, s, s); a4 = __builtin_fmaf(a4, s, s); a5 = __builtin_fmaf(a5, s, s); a6 = __builtin_fmaf(a6, s, s); a7 = __builtin_fmaf(a7, s, s);
                constexpr int NS = MODE == 4 ? 2 : (MODE == 5 ? 4 : 8);
#pragma unroll
                for (int q = 0; q < NS; ++q) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sc));
            }
            if (MODE == 1) { a0 = __builtin_fmaf(a0, s, s); a1 = __builtin_fmaf(a1, s, s); a2 = __builtin_fmaf(a2, s, s); a3 = __builtin_fmaf(a3, s, s); a4 = __builtin_fmaf(a4, s, s); a5 = __builtin_fmaf(a5, s, s); a6 = __builtin_fmaf(a6, s, s); a7 = __builtin_fmaf(a7, s, s); }
        }
    }
    const f2 ps = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (MODE >= 2 ? ps.x + ps.y : 0.0f) + (MODE >= 4 ? (float)sc : 0.0f);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 7; ++mode)
        for (int blocks : {256, 512, 1024, 2048}) {
            const int iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256>>>(d, iters, 1.0001f); else if (mode == 1) k<1><<<blocks, 256>>>(d, iters, 1.0001f); else if (mode == 2) k<2><<<blocks, 256>>>(d, iters, 1.0001f); else if (mode == 3) k<3><<<blocks, 256>>>(d, iters, 1.0001f); else if (mode == 4) k<4><<<blocks, 256>>>(d, iters, 1.0001f); else if (mode == 5) k<5><<<blocks, 256>>>(d, iters, 1.0001f); else k<6><<<blocks, 256>>>(d, iters, 1.0001f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double winstr = (double)blocks * 4 * iters * 64.0;   // wave-instructions
            printf("mode %d blocks %d (%.1f waves/SIMD): %.3f ms  %.1f G wave-instr/s  = %.3f instr/cycle/SIMD @2.4GHz\n", mode, blocks, blocks * 4 / 1024.0, ms,
                   winstr / ms / 1e6, winstr / (ms * 1e-3) / 1024 / 2.4e9);
        }
    return 0;
}
