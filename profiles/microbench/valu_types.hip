// micro-benchmark: issue rate per VALU instruction type on gfx950 (hipcc --offload-arch=gfx950 -O3 -o valu_types valu_types.hip).
// Each kernel runs one instruction form on eight independent registers (inline asm, nothing for the compiler to pack or fold);
// 8 waves per SIMD.  Prints wave-instructions per cycle per SIMD at the 2.4 GHz nominal clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters, float s)                              \
    {                                                                                                        \
        float a[8], b = s, c = s + 1.0f;                                                                     \
        for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k;                                                  \
        for (int i = 0; i < iters; ++i) {                                                                    \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                  \
                _Pragma("unroll") for (int k = 0; k < 8; ++k) asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c)); \
            }                                                                                                \
        }                                                                                                    \
        float t = 0; for (int k = 0; k < 8; ++k) t += a[k];                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = t;                                                      \
    }
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_addu, "v_add_u32 %0, %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp, "v_cmp_gt_f32 vcc, %0, %1")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
KERNEL(k_fma_sgpr, "v_fma_f32 %0, %0, s4, s4")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
KERNEL(k_min, "v_min_f32 %0, %0, %1")
KERNEL(k_add_sgpr, "v_add_f32 %0, s4, %0")
KERNEL(k_mul_sgpr, "v_mul_f32 %0, s4, %0")
KERNEL(k_mul_lit, "v_mul_f32 %0, 0x40490fdb, %0")
KERNEL(k_add_inl, "v_add_f32 %0, 1.0, %0")
KERNEL(k_fma_inl, "v_fma_f32 %0, %0, %1, 1.0")
KERNEL(k_fma_abs, "v_fma_f32 %0, %0, |%1|, %2")
KERNEL(k_fma_neg, "v_fma_f32 %0, %0, %1, -%2")
KERNEL(k_cnd_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL(k_cnd_vcc2, "v_cndmask_b32_e32 %0, %1, %2, vcc")
KERNEL(k_cmp_e64, "v_cmp_gt_f32_e64 s[10:11], %0, %1")
KERNEL(k_cmp_u32, "v_cmp_eq_u32_e32 vcc, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 5")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_rsq, "v_rsq_f32 %0, %0")
KERNEL(k_frexp, "v_frexp_mant_f32 %0, %0")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1")
KERNEL(k_divscale, "v_div_scale_f32 %0, vcc, %0, %1, %2")
KERNEL(k_divfmas, "v_div_fmas_f32 %0, %0, %1, %2")
KERNEL(k_divfixup, "v_div_fixup_f32 %0, %0, %1, %2")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2")
KERNEL(k_class, "v_cmp_class_f32 vcc, %0, %1")
KERNEL(k_floor, "v_floor_f32 %0, %0")
KERNEL(k_fract, "v_fract_f32 %0, %0")
KERNEL(k_cvtu, "v_cvt_u32_f32 %0, %0")
KERNEL(k_cmp_cnd, "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %2, vcc")
KERNEL(k_cmp_cnd4, "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_cndmask_b32_e32 %0, %2, %0, vcc\n v_cndmask_b32_e32 %0, %0, %1, vcc")
KERNEL(k_cmp64_cnd, "v_cmp_gt_f32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[10:11]")
// packed: register pairs
__global__ void __launch_bounds__(256) k_pk_fma(float *out, int iters, float s)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8], b = {s, s}, c = {s + 1.0f, s};
    for (int k = 0; k < 8; ++k) a[k] = f2{(float)threadIdx.x + k, (float)k};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        }
    }
    f2 t = {0, 0}; for (int k = 0; k < 8; ++k) t += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t.x + t.y;
}
__global__ void __launch_bounds__(256) k_pk_add(float *out, int iters, float s)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8], b = {s, s};
    for (int k = 0; k < 8; ++k) a[k] = f2{(float)threadIdx.x + k, (float)k};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        }
    }
    f2 t = {0, 0}; for (int k = 0; k < 8; ++k) t += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t.x + t.y;
}
template <class K> static void run(const char *name, K kern, float *d)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 8000, blocks = 2048;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        kern<<<blocks, 256>>>(d, iters, 1.0001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double winstr = (double)blocks * 4 * iters * 64.0;
    printf("%-12s %.3f ms  %7.1f G wave-instr/s  %.3f instr/cycle/SIMD  (%.2f cycles per instruction)\n", name, ms, winstr / ms / 1e6,
           winstr / (ms * 1e-3) / 1024 / 2.4e9, 1024 * 2.4e9 * (ms * 1e-3) / winstr);
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 2048 * 4);
    run("v_fma_f32", k_fma, d); run("v_mul_f32", k_mul, d); run("v_add_f32", k_add, d); run("v_max_f32", k_max, d); run("v_max3_f32", k_max3, d);
    run("v_mov_b32", k_mov, d); run("v_add_u32", k_addu, d); run("v_and_b32", k_and, d); run("v_lshl_add", k_lshl_add, d); run("v_cndmask", k_cndmask, d);
    run("v_cmp_gt_f32", k_cmp, d); run("v_mul_lo_u32", k_mul_lo, d); run("v_rcp_f32", k_rcp, d); run("v_sqrt_f32", k_sqrt, d); run("v_cvt_f32_u32", k_cvt, d);
    run("v_fma (sgpr)", k_fma_sgpr, d); run("v_pk_fma_f32", k_pk_fma, d); run("v_pk_add_f32", k_pk_add, d);
    run("v_fmac_f32", k_fmac, d); run("v_sub_f32", k_sub, d); run("v_min_f32", k_min, d); run("v_add_f32 sgpr", k_add_sgpr, d); run("v_mul_f32 sgpr", k_mul_sgpr, d);
    run("v_mul literal", k_mul_lit, d); run("v_add inline", k_add_inl, d); run("v_fma inline", k_fma_inl, d); run("v_fma |abs|", k_fma_abs, d); run("v_fma -neg", k_fma_neg, d);
    run("cndmask e64", k_cnd_e64, d); run("cndmask vcc", k_cnd_vcc2, d); run("v_cmp e64", k_cmp_e64, d); run("v_cmp_eq_u32", k_cmp_u32, d); run("v_xor_b32", k_xor, d);
    run("v_lshrrev", k_lshr, d); run("v_lshlrev", k_lshl, d); run("v_mul_u32_u24", k_mul24, d); run("v_mad_u32_u24", k_mad24, d); run("v_bfe_u32", k_bfe, d);
    run("v_add3_u32", k_add3, d); run("v_or3_b32", k_or3, d); run("v_and_or_b32", k_and_or, d); run("v_mul_hi_u32", k_mulhi, d); run("v_rsq_f32", k_rsq, d);
    run("v_frexp_mant", k_frexp, d); run("v_ldexp_f32", k_ldexp, d); run("v_div_scale", k_divscale, d); run("v_div_fmas", k_divfmas, d); run("v_div_fixup", k_divfixup, d);
    run("v_med3_f32", k_med3, d); run("v_min3_f32", k_min3, d); run("v_cmp_class", k_class, d); run("v_floor_f32", k_floor, d); run("v_fract_f32", k_fract, d); run("v_cvt_u32_f32", k_cvtu, d);
    run("cmp+cnd (2)", k_cmp_cnd, d); run("cmp+3cnd (4)", k_cmp_cnd4, d); run("cmp64+cnd64(2)", k_cmp64_cnd, d);
    return 0;
}
