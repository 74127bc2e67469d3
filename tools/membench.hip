// membench.hip -- streaming-copy micro-benchmark used to choose the access pattern of the
// HBM-bound kernels (grid size, bytes in flight per lane, grid-stride vs block-contiguous,
// non-temporal loads/stores).  Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef long long i64;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool NT_LD, bool NT_ST, bool CONTIG>
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ in, u32x4* __restrict__ out, i64 n)
{
    // CONTIG: each block walks chunks of 256*U consecutive vectors; else classic grid-stride
    const i64 nthreads = (i64)gridDim.x * 256;
    if (CONTIG) {
        const i64 chunk = 256 * U;
        for (i64 base = (i64)blockIdx.x * chunk; base < n; base += (i64)gridDim.x * chunk) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const i64 i = base + u * 256 + threadIdx.x;
                if (i < n) v[u] = NT_LD ? __builtin_nontemporal_load(in + i) : in[i];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const i64 i = base + u * 256 + threadIdx.x;
                if (i < n) { if (NT_ST) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; }
            }
        }
    } else {
        for (i64 i0 = (i64)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += nthreads * U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const i64 i = i0 + u * nthreads;
                if (i < n) v[u] = NT_LD ? __builtin_nontemporal_load(in + i) : in[i];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const i64 i = i0 + u * nthreads;
                if (i < n) { if (NT_ST) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; }
            }
        }
    }
}

template <int U, bool NT_LD, bool NT_ST, bool CONTIG>
static double run(const u32x4* in, u32x4* out, i64 n, int grid, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) copy_k<U, NT_LD, NT_ST, CONTIG><<<grid, 256>>>(in, out, n);
    CK(hipDeviceSynchronize());
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(a));
        copy_k<U, NT_LD, NT_ST, CONTIG><<<grid, 256>>>(in, out, n);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms[i], a, b));
    }
    std::sort(ms.begin(), ms.end());
    return 2.0 * n * 16 / (ms[reps / 2] * 1e-3) / 1e9;   // GB/s read+write, median
}

int main(int argc, char** argv)
{
    const i64 mb = argc > 1 ? atoll(argv[1]) : 512;
    const i64 n = mb * 1024 * 1024 / 16;
    u32x4 *in, *out;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16));
    CK(hipMemset(in, 1, n * 16)); CK(hipMemset(out, 0, n * 16));
    const int reps = 15;
    printf("buffer %lld MiB each way; GB/s = (read+write)/median time\n", mb);
    // hipMemcpyAsync D2D for reference
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        std::vector<float> ms(reps);
        for (int i = 0; i < reps; i++) {
            CK(hipEventRecord(a)); CK(hipMemcpyAsync(out, in, n * 16, hipMemcpyDeviceToDevice, 0));
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[i], a, b));
        }
        std::sort(ms.begin(), ms.end());
        printf("%-44s %8.0f GB/s\n", "hipMemcpyAsync D2D", 2.0 * n * 16 / (ms[reps / 2] * 1e-3) / 1e9);
    }
    const int grids[] = {512, 1024, 2048, 4096, 8192, 16384, 65536};
    for (int g : grids) {
        printf("grid %6d  stride U1 %6.0f | U2 %6.0f | U4 %6.0f | U8 %6.0f || contig U1 %6.0f | U2 %6.0f | U4 %6.0f | U8 %6.0f\n", g,
               run<1, false, false, false>(in, out, n, g, reps), run<2, false, false, false>(in, out, n, g, reps),
               run<4, false, false, false>(in, out, n, g, reps), run<8, false, false, false>(in, out, n, g, reps),
               run<1, false, false, true>(in, out, n, g, reps), run<2, false, false, true>(in, out, n, g, reps),
               run<4, false, false, true>(in, out, n, g, reps), run<8, false, false, true>(in, out, n, g, reps));
    }
    for (int g : grids) {
        printf("grid %6d  contig U4: nt-ld %6.0f | nt-st %6.0f | nt-both %6.0f || stride U4: nt-ld %6.0f | nt-st %6.0f | nt-both %6.0f\n", g,
               run<4, true, false, true>(in, out, n, g, reps), run<4, false, true, true>(in, out, n, g, reps),
               run<4, true, true, true>(in, out, n, g, reps),
               run<4, true, false, false>(in, out, n, g, reps), run<4, false, true, false>(in, out, n, g, reps),
               run<4, true, true, false>(in, out, n, g, reps));
    }
    // one vector per thread, no loop
    {
        const i64 g = (n + 255) / 256;
        printf("grid full (%lld blocks) U1 plain %6.0f | nt-both %6.0f\n", g,
               run<1, false, false, false>(in, out, n, (int)g, reps), run<1, true, true, false>(in, out, n, (int)g, reps));
    }
    return 0;
}
