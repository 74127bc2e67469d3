// valubench.hip -- issue rate of a few VALU instruction kinds on gfx950, one to eight waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valubench tools/valubench.hip ; run: tools/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32;

template <int KIND>
__global__ __launch_bounds__(256) void bench(float* out, int iters, float seed)
{
    f32x2 a[8];
    for (int k = 0; k < 8; k++) a[k] = f32x2{seed + k + threadIdx.x, seed - k};
    const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (KIND == 0) {            // 2 scalar fma
                a[k].x = __builtin_fmaf(a[k].x, m.x, c.x);
                a[k].y = __builtin_fmaf(a[k].y, m.y, c.y);
            } else if (KIND == 1) {     // 1 packed fma
                a[k] = __builtin_elementwise_fma(a[k], m, c);
            } else if (KIND == 2) {     // 1 packed add
                a[k] = a[k] + c;
            } else if (KIND == 3) {     // 2 pk_max_u16
                u32 x = __builtin_bit_cast(u32, a[k].x), y = __builtin_bit_cast(u32, a[k].y);
                x = __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, y)));
                y = __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(u16x2, y), __builtin_bit_cast(u16x2, x + 1)));
                a[k].x = __builtin_bit_cast(float, x); a[k].y = __builtin_bit_cast(float, y);
            } else if (KIND == 4) {     // 2 dpp movs
                u32 x = __builtin_bit_cast(u32, a[k].x), y = __builtin_bit_cast(u32, a[k].y);
                x = __builtin_amdgcn_update_dpp(x, y, 0x138, 0xf, 0xf, false);
                y = __builtin_amdgcn_update_dpp(y, x, 0x130, 0xf, 0xf, false);
                a[k].x = __builtin_bit_cast(float, x); a[k].y = __builtin_bit_cast(float, y);
            } else if (KIND == 5) {     // 2 cvt_f32_ubyte + ...
                u32 x = __builtin_bit_cast(u32, a[k].x);
                a[k].x = (float)(x & 0xffu) ; a[k].y = (float)((x >> 8) & 0xffu) + a[k].y;
            }
        }
    }
    float s = 0;
    for (int k = 0; k < 8; k++) s += a[k].x + a[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int instr_per_iter, float* d)
{
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;   // 4 waves per block -> wps waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        bench<KIND><<<blocks, 256>>>(d, 100, 1.0f);
        hipEventRecord(e0);
        bench<KIND><<<blocks, 256>>>(d, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // cycles per instruction per SIMD at 2.4 GHz
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * instr_per_iter * wps);
        printf("%-28s waves/SIMD %d  %.3f ms  %.2f cycles/instr/SIMD (at 2.4 GHz)\n", name, wps, ms, cyc);
    }
}

int main()
{
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32 x2", 16, d);
    run<1>("v_pk_fma_f32", 8, d);
    run<2>("v_pk_add_f32", 8, d);
    run<3>("v_pk_max/min_u16 (+add)", 24, d);
    run<4>("v_mov_b32_dpp x2", 16, d);
    run<5>("cvt_f32_ubyte x2 + add", 24, d);
    hipFree(d);
    return 0;
}
