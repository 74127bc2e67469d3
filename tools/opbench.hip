// opbench.hip -- what a scalar-register operand costs a vector instruction on gfx950: the same instruction, 16
// independent accumulators per lane, operand 1 from an SGPR or from a VGPR, two to four waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/opbench tools/opbench.hip ; run: tools/opbench
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPS(X) \
    X(0, "v_fmac_f32_e32 %0, %1, %2") \
    X(16, "v_cvt_f32_ubyte1_e32 %0, %2") \
    X(17, "v_min_u16_e32 %0, %1, %2") \
    X(18, "v_pk_min_u16 %0, %1, %2") \
    X(19, "v_med3_u16 %0, %1, %2, %0") \
    X(1, "v_fma_f32 %0, %1, %2, %0") \
    X(2, "v_add_u32_e32 %0, %1, %2") \
    X(3, "v_dot2_u32_u16 %0, %1, %2, %0") \
    X(4, "v_pk_mul_lo_u16 %0, %1, %2") \
    X(5, "v_pk_add_u16 %0, %1, %2") \
    X(6, "v_mad_u32_u24 %0, %1, %2, %0") \
    X(7, "v_and_or_b32 %0, %2, %1, %0") \
    X(8, "v_pk_fma_f16 %0, %1, %2, %0") \
    X(9, "v_dot2_f32_f16 %0, %1, %2, %0") \
    X(10, "v_mul_f32_e32 %0, %1, %2") \
    X(11, "v_perm_b32 %0, %0, %2, %1") \
    X(12, "v_pk_max_u16 %0, %1, %2") \
    X(13, "v_mad_u16 %0, %1, %2, %0") \
    X(14, "v_pk_mad_u16 %0, %1, %2, %0")

template <int KIND, bool SCALAR>
__global__ __launch_bounds__(256) void opbench(float* out, int iters, float seed)
{
    float a[16];
    for (int k = 0; k < 16; k++) a[k] = seed + k + threadIdx.x;
    float vm = seed * 1.0001f, vc = 0.25f;
    asm volatile("" : "+v"(vm));
    asm volatile("" : "+v"(vc));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a2[16], vm2 = {vm, vm}, vc2 = {vc, vc};
    for (int k = 0; k < 16; k++) a2[k] = f32x2{a[k], a[k]};
    asm volatile("" : "+v"(vm2));
    asm volatile("" : "+v"(vc2));
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (KIND == 100) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[k]) : "v"(vm2), "v"(vc2));   // (both columns: vector operands)
#define X(N, TEXT) \
            if (KIND == N) { if (SCALAR) asm volatile(TEXT : "+v"(a[k]) : "s"(seed), "v"(vc)); else asm volatile(TEXT : "+v"(a[k]) : "v"(vm), "v"(vc)); }
            OPS(X)
#undef X
#endif
        }
    }
    float s = 0;
    for (int k = 0; k < 16; k++) s += a[k] + a2[k].x + a2[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, bool SCALAR>
double run(float* d, int wps)
{
    // long enough for the clocks to settle under this instruction's load: about 30 ms of warm-up, 15 ms timed
    const int iters = 200000, blocks = 256 * wps;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 6; r++) opbench<KIND, SCALAR><<<blocks, 256>>>(d, iters, 1.0f);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) opbench<KIND, SCALAR><<<blocks, 256>>>(d, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms * 1e-3 * 2.4e9 / (3.0 * iters * 16 * wps);
}

int main()
{
    float* d;
    (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    printf("cycles per instruction per SIMD at 2.4 GHz; waves per SIMD: 2, 3, 4\n");
    printf("%-36s %-22s %-22s\n", "instruction", "operand 1 in a VGPR", "operand 1 in an SGPR");
#define X(N, TEXT) \
    printf("%-36s %5.2f %5.2f %5.2f      %5.2f %5.2f %5.2f\n", TEXT, run<N, false>(d, 2), run<N, false>(d, 3), run<N, false>(d, 4), \
           run<N, true>(d, 2), run<N, true>(d, 3), run<N, true>(d, 4));
    OPS(X)
#undef X
    printf("%-36s %5.2f %5.2f %5.2f      %5.2f %5.2f %5.2f\n", "v_pk_fma_f32 (2 fma per instruction)", run<100, false>(d, 2), run<100, false>(d, 3), run<100, false>(d, 4),
           run<100, true>(d, 2), run<100, true>(d, 3), run<100, true>(d, 4));
    (void)hipFree(d);
    return 0;
}
