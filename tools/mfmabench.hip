// mfmabench.hip -- cycles per matrix instruction on one SIMD: v_mfma_f32_16x16x16_f16 against v_mfma_f32_16x16x32_f16
// (gfx950), four independent accumulators, 1, 2 or 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o
// tools/mfmabench tools/mfmabench.hip; run on the GPU box: ./tools/mfmabench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K32>
__global__ __launch_bounds__(1024) void k(float* out, int iters)
{
    f4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    h4 a4 = {(_Float16)threadIdx.x, 1, 2, 3}, b4 = {1, (_Float16)0.5f, 0, 1};
    h8 a8 = {(_Float16)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b8 = {1, (_Float16)0.5f, 0, 1, 0, 0, 1, 0};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                acc[q] = K32 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[q], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[q], 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

int main()
{
    float* out;
    hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpb = 256; wpb <= 1024; wpb *= 2)          // 1, 2, 4 waves per SIMD (one block per CU)
        for (int k32 = 0; k32 < 2; k32++) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (k32) hipLaunchKernelGGL(k<1>, dim3(256), dim3(wpb), 0, 0, out, iters);
                else hipLaunchKernelGGL(k<0>, dim3(256), dim3(wpb), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double per_simd = (double)iters * 32 * (wpb / 256);
            printf("%s, %d wave(s) per SIMD: %.3f ms, %.1f ns per instruction per SIMD = %.1f cycles at 2.4 GHz\n",
                   k32 ? "16x16x32_f16" : "16x16x16_f16", wpb / 256, best, best * 1e6 / per_simd, best * 1e6 / per_simd * 2.4);
        }
    return 0;
}
