// valubench.hip -- issue rate of a few VALU instruction kinds on gfx950, one to eight waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valubench tools/valubench.hip ; run: tools/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32;
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

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
            } else if (KIND == 6) {     // 2 v_dot2_u32_u16
                u32 x = __builtin_bit_cast(u32, a[k].x), y = __builtin_bit_cast(u32, a[k].y);
                x = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, y), x, false);
                y = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, y), __builtin_bit_cast(u16x2, x), y, false);
                a[k].x = __builtin_bit_cast(float, x); a[k].y = __builtin_bit_cast(float, y);
            } else if (KIND == 7) {     // 2 v_dot2_f32_f16
                u32 x = __builtin_bit_cast(u32, a[k].x), y = __builtin_bit_cast(u32, a[k].y);
                a[k].x = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, x), __builtin_bit_cast(h16x2, y), a[k].x, false);
                a[k].y = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, y), __builtin_bit_cast(h16x2, x), a[k].y, false);
            } else if (KIND == 8) {     // 2 v_sqrt_f32
                a[k].x = __builtin_amdgcn_sqrtf(a[k].x);
                a[k].y = __builtin_amdgcn_sqrtf(a[k].y);
            } else if (KIND == 9) {     // 2 v_perm_b32
                u32 x = __builtin_bit_cast(u32, a[k].x), y = __builtin_bit_cast(u32, a[k].y);
                x = __builtin_amdgcn_perm(x, y, 0x05040100u);
                y = __builtin_amdgcn_perm(y, x, 0x07060302u);
                a[k].x = __builtin_bit_cast(float, x); a[k].y = __builtin_bit_cast(float, y);
            } else if (KIND == 10) {    // 2 v_alignbit_b32
                u32 x = __builtin_bit_cast(u32, a[k].x), y = __builtin_bit_cast(u32, a[k].y);
                x = __builtin_amdgcn_alignbit(x, y, 16);
                y = __builtin_amdgcn_alignbit(y, x, 8);
                a[k].x = __builtin_bit_cast(float, x); a[k].y = __builtin_bit_cast(float, y);
            } else if (KIND == 11) {    // v_cvt_u32_f32 + v_cvt_f32_u32
                u32 x = (u32)a[k].x;
                a[k].x = (float)(x + 1u) ; 
            } else if (KIND == 13) {    // 2 v_pk_mad_u16
                u16x2 x = __builtin_bit_cast(u16x2, a[k].x), y = __builtin_bit_cast(u16x2, a[k].y);
                x = x * y + x;
                y = y * x + y;
                a[k].x = __builtin_bit_cast(float, x); a[k].y = __builtin_bit_cast(float, y);
            } else if (KIND == 14) {    // 2 v_mad_u32_u24
                u32 x = __builtin_bit_cast(u32, a[k].x), y = __builtin_bit_cast(u32, a[k].y);
                x = (x & 0xffffffu) * (y & 0xffffffu) + x;
                y = (y & 0xffffffu) * (x & 0xffffffu) + y;
                a[k].x = __builtin_bit_cast(float, x); a[k].y = __builtin_bit_cast(float, y);
            } else if (KIND == 15) {    // 1 v_cvt_pkrtz_f16_f32 + 1 v_pk_add_f16
                h16x2 h = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(a[k].x, a[k].y));
                h = h + __builtin_bit_cast(h16x2, a[k].y);
                a[k].x = __builtin_bit_cast(float, h);
            } else if (KIND == 17) {    // 2 v_fma_mix_f32 (f16 low / high half of src0 widened inside the fma)
#if defined(__HIP_DEVICE_COMPILE__)
                const u32 h = __builtin_bit_cast(u32, c.x) | 0x3c003c00u;
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a[k].x) : "v"(h), "v"(m.x));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[k].y) : "v"(h), "v"(m.y));
#endif
            } else if (KIND == 16) {    // 2 v_floor_f32
                a[k].x = __builtin_floorf(a[k].x + 0.5f);   // folds? keep the add out of the count if it stays
                a[k].y = __builtin_floorf(a[k].y);
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

// sqrt-table shape: every lane reads a byte of a 64 KiB LDS table at a data-dependent address,
// 4 lookups per iteration beside 8 fma (LDS issue beside VALU work)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void lds_bench(float* out, int iters, u32 seed, int spread)
{
    __shared__ unsigned char tab[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += BLOCK) ((u32*)tab)[i] = (u32)i * 2654435761u;
    __syncthreads();
    u32 x = seed + threadIdx.x * 7919u + blockIdx.x;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        u32 s = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            x = x * 1664525u + 1013904223u;
            s += tab[(x >> 8) & (u32)spread];
        }
        acc = __builtin_fmaf(acc, 0.999f, (float)s);
    }
    out[blockIdx.x * BLOCK + threadIdx.x] = acc;
}

void run_lds(float* d)
{
    const int iters = 20000;
    for (int spread : {0xffff, 0x3ff}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        lds_bench<768><<<512, 768>>>(d, 100, 1u, spread);
        hipEventRecord(e0);
        lds_bench<768><<<512, 768>>>(d, iters, 1u, spread);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // 6 waves per SIMD (2 blocks of 12 waves per CU); cycles per lookup instruction per CU
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 4 * 24);
        printf("ds_read_u8 random (spread %5d) 6 waves/SIMD  %.3f ms  %.2f cycles/lookup-instr/CU\n", spread + 1, ms, cyc);
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
    run<6>("v_dot2_u32_u16 x2", 16, d);
    run<7>("v_dot2_f32_f16 x2", 16, d);
    run<8>("v_sqrt_f32 x2", 16, d);
    run<9>("v_perm_b32 x2", 16, d);
    run<10>("v_alignbit_b32 x2", 16, d);
    run<11>("cvt_u32_f32+add+cvt_f32_u32", 24, d);
    run<13>("v_pk_mul_lo_u16+v_pk_add_u16 x2", 32, d);
    run<14>("v_mad_u32_u24 x2 (+and)", 16, d);
    run<15>("cvt_pkrtz + pk_add_f16", 16, d);
    run<16>("v_floor_f32 x2 (+add)", 24, d);
    run<17>("v_fma_mix_f32 x2", 16, d);
    run_lds(d);
    hipFree(d);
    return 0;
}
