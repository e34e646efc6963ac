#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../micro_raytracer_amd/csrc/mrt_trace.h"
using namespace mrt;
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, u32 seed)
{
    u32 pk = seed + threadIdx.x * 977u + blockIdx.x;
    V3 n = v3(0.3f, 0.4f, 0.8f);
    float accf = 0.0f; u32 accu = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { accu += draw_u32(pk, i) ^ draw_u32(pk, i + 7); }
        if (MODE == 1) { accf += acos_(1.0f - 2.0f * u32_to_unit(pk + i * 0x9E3779B9u)); }
        if (MODE == 2) { float s, c, s2, c2; float x = u32_to_unit(pk + i * 0x9E3779B9u) * 6.28f; sincos_(x, s, c); sincos_(x * 0.5f, s2, c2); accf += s * c2 + c * s2; }
        if (MODE == 3) { V3 v = norm(v3(accf + 1.0f, (float)i, 0.5f)); accf += v.x; }
        if (MODE == 4) { V3 v = rand_normal(n, 1.0f, u32_to_unit(draw_u32(pk, i)), u32_to_unit(draw_u32(pk, i + 7))); accf += v.x; n = v; }
        if (MODE == 5) { accu += mix32(pk + i); }
        if (MODE == 6) { accf += sqrt_(accf + (float)i); }
        if (MODE == 7) { accf += 1.0f / (accf + (float)i); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = accf + (float)accu;
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *names[] = {"2 draws", "acos", "2 sincos", "norm", "rand_normal+2draws", "mix32", "sqrt", "div"};
    for (int mode = 0; mode < 8; ++mode) {
        const int iters = 4000, blocks = 2048;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            switch (mode) { case 0: k<0><<<blocks, 256>>>(d, iters, 1); break; case 1: k<1><<<blocks, 256>>>(d, iters, 1); break; case 2: k<2><<<blocks, 256>>>(d, iters, 1); break;
                case 3: k<3><<<blocks, 256>>>(d, iters, 1); break; case 4: k<4><<<blocks, 256>>>(d, iters, 1); break; case 5: k<5><<<blocks, 256>>>(d, iters, 1); break;
                case 6: k<6><<<blocks, 256>>>(d, iters, 1); break; case 7: k<7><<<blocks, 256>>>(d, iters, 1); break; }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double calls = (double)blocks * 4 * iters;    // wave-level calls
        printf("%-20s %.3f ms  -> %.1f cycles per wave-call per SIMD (8 waves/SIMD)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / (calls / 1024.0));
    }
    return 0;
}
