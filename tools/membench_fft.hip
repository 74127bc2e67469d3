// membench_fft.hip -- what the memory system delivers for the access patterns of the FFT passes,
// without any arithmetic: every block copies "lines" the way fft_lines_kernel reads and writes them.
//   row pattern   : a line is 32 KiB contiguous; lane t moves points t + 256 k, k = 0..15
//   column pattern: 64 adjacent columns x 64 points, point stride 2 MiB (row stride 32 KiB x 64)
// each with 8-byte or 16-byte accesses per lane, persistent (3 blocks per CU, register prefetch of
// the next line) or one block per line.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/membench_fft tools/membench_fft.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef long long i64;

// V = floats per access (2: b64, 4: b128).  A line = 4096 points of 8 bytes.
template <int V, bool COL, bool PERSIST>
__global__ __launch_bounds__(256) void lines_k(const float* __restrict__ in, float* __restrict__ out, int total_lines)
{
    typedef float vec __attribute__((ext_vector_type(V)));
    constexpr int PTS = 16 * 2 / V;          // accesses per lane: 16 (b64) or 8 (b128)
    const int t = threadIdx.x;
    const int step = PERSIST ? gridDim.x : total_lines;
    vec cur[PTS], nxt[PTS];
    auto addr = [&](int line, int k) -> i64 {      // in floats
        if (!COL) return (i64)line * 8192 + ((i64)t + 256 * k) * V;
        // column pattern: "line" = block of 64 columns x 64 rows-with-stride; 4096 points per block
        // line -> (plane-local) column group cg = line % 64, row group rg = line / 64
        const i64 plane = line / 4096, l = line % 4096;
        const i64 cg = l % 64, rg = l / 64;              // 64 column groups x 64 row residues per plane
        // lanes: 64/ (V/2) lanes cover 64 columns (8 B each) => 512 B runs
        constexpr int LPR = 64 * 2 / V;              // lanes per 512-byte run
        const int row_in_blk = (t / LPR) + (256 / LPR) * k;          // 0..63
        const i64 row = plane * 4096 + rg + 64 * (i64)row_in_blk;     // stride 64 rows
        return (row * 4096 + cg * 64) * 2 + (t % LPR) * V;
    };
    int line = blockIdx.x;
    if (line >= total_lines) return;
#pragma unroll
    for (int k = 0; k < PTS; k++) nxt[k] = *(const vec*)(in + addr(line, k));
    for (; line < total_lines; line += step) {
#pragma unroll
        for (int k = 0; k < PTS; k++) cur[k] = nxt[k];
        if (line + step < total_lines) {
#pragma unroll
            for (int k = 0; k < PTS; k++) nxt[k] = *(const vec*)(in + addr(line + step, k));
        }
#pragma unroll
        for (int k = 0; k < PTS; k++) *(vec*)(out + addr(line, k)) = cur[k] * 2.0f;
    }
}

// Whole-column tile: a block of 1024 lanes owns all 4096 rows of C adjacent columns (C = 4: 128 KiB of LDS in
// the real kernel, so one block per CU; global accesses are runs of C * 8 bytes at a 32 KiB stride) and moves
// them with 16-byte accesses, every load of the tile issued before the first store -- the shape a single-pass
// column transform would have.  REMAP: the 128 / (8 C) tiles that share a 128-byte line get consecutive slots
// on ONE XCD (blocks b, b+8, b+16, ... share an XCD) so that they meet in that XCD's L2.
template <int C, bool REMAP>
__global__ __launch_bounds__(1024) void coltile_k(const float* __restrict__ in, float* __restrict__ out, int total_tiles)
{
    extern __shared__ float occupancy_pad[];
    constexpr int LPR = C * 8 / 16;                 // lanes per row run
    constexpr int PTS = 4096 * LPR / 1024;          // 16-byte accesses per lane
    const unsigned nblk = gridDim.x, b = blockIdx.x;
    const unsigned q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    const unsigned tile = REMAP ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx : b;
    if ((int)tile >= total_tiles) return;
    constexpr int TPP = 4096 / C;                   // tiles per plane
    const i64 plane = tile / TPP, cg = tile % TPP;
    const int t = threadIdx.x;
    f4 v[PTS];
#pragma unroll
    for (int k = 0; k < PTS; k++) {
        const i64 row = plane * 4096 + t / LPR + (1024 / LPR) * k;
        v[k] = *(const f4*)(in + (row * 4096 + cg * C) * 2 + (t % LPR) * 4);
    }
#pragma unroll
    for (int k = 0; k < PTS; k++) {
        const i64 row = plane * 4096 + t / LPR + (1024 / LPR) * k;
        *(f4*)(out + (row * 4096 + cg * C) * 2 + (t % LPR) * 4) = v[k] * 2.0f;
    }
}

template <int C, bool REMAP>
void run_coltile(const char* name, const float* in, float* out, int planes, int lds_bytes)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = planes * 4096 / C;
    hipFuncSetAttribute((const void*)coltile_k<C, REMAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 3; i++) coltile_k<C, REMAP><<<tiles, 1024, lds_bytes>>>(in, out, tiles);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) coltile_k<C, REMAP><<<tiles, 1024, lds_bytes>>>(in, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-52s lds %6d  %.3f ms  %.0f GB/s  (%s)\n", name, lds_bytes, ms, (double)planes * 4096 * 65536.0 / ms / 1e6,
           hipGetErrorString(hipGetLastError()));
}

// The two passes of a vector-radix (2-D four-step) plan for 4096 x 4096 = (32 * 128)^2, each reading and writing
// the plane ONCE (32 instead of 48 B/px per direction), as pure copies of 128-KiB tiles, one block of 1024 lanes
// per CU, every load issued before the first store:
//   MODE 1  the 32 x 32 points (128 a + r, 128 b + c), c in a run of 16 contiguous columns (128-byte runs,
//           1-KiB apart within a row, rows 128 apart)
//   MODE 2  a contiguous tile of 128 rows x 128 columns (1-KiB runs)
template <int MODE>
__global__ __launch_bounds__(1024) void tile2_k(const float* __restrict__ in, float* __restrict__ out, int total_tiles)
{
    extern __shared__ float occupancy_pad[];
    const unsigned nblk = gridDim.x, b = blockIdx.x;
    const unsigned q = nblk >> 3, r8 = nblk & 7, xcd = b & 7, idx = b >> 3;
    const unsigned tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    if ((int)tile >= total_tiles) return;
    const i64 plane = tile / 1024;
    const int tt = tile % 1024;
    const int t = threadIdx.x;
    i64 off[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int i = t + 1024 * k;                 // 16-byte access index within the tile
        i64 row, col;                               // col in points
        if (MODE == 1) {
            const int r = tt / 8, c0 = (tt % 8) * 16;
            const int run = i / 8, a = run / 32, bb = run % 32;
            row = 128 * a + r; col = 128 * bb + c0 + (i % 8) * 2;
        } else {
            const int ty = tt / 32, tx = tt % 32;
            row = 128 * ty + i / 64; col = 128 * tx + (i % 64) * 2;
        }
        off[k] = ((plane * 4096 + row) * 4096 + col) * 2;
    }
    f4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = *(const f4*)(in + off[k]);
#pragma unroll
    for (int k = 0; k < 8; k++) *(f4*)(out + off[k]) = v[k] * 2.0f;
}

template <int MODE>
void run_tile2(const char* name, const float* in, float* out, int planes, int lds_bytes)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = planes * 1024;
    hipFuncSetAttribute((const void*)tile2_k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 3; i++) tile2_k<MODE><<<tiles, 1024, lds_bytes>>>(in, out, tiles);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) tile2_k<MODE><<<tiles, 1024, lds_bytes>>>(in, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-60s lds %6d  %.3f ms  %.0f GB/s  (%s)\n", name, lds_bytes, ms, (double)planes * 4096 * 65536.0 / ms / 1e6,
           hipGetErrorString(hipGetLastError()));
}

template <int V, bool COL, bool PERSIST>
void run(const char* name, const float* in, float* out, int total_lines, int grid)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int g = PERSIST ? grid : total_lines;
    for (int i = 0; i < 3; i++) lines_k<V, COL, PERSIST><<<g, 256>>>(in, out, total_lines);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) lines_k<V, COL, PERSIST><<<g, 256>>>(in, out, total_lines);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-44s grid %6d  %.3f ms  %.0f GB/s\n", name, g, ms, (double)total_lines * 65536.0 / ms / 1e6);
}

int main()
{
    const int planes = 8;                       // 8 x 128 MiB in + out
    const int total_lines = planes * 4096;
    float *in, *out;
    hipMalloc(&in, (size_t)total_lines * 32768);
    hipMalloc(&out, (size_t)total_lines * 32768);
    hipMemset(in, 0, (size_t)total_lines * 32768);
    for (int grid : {512, 768, 1024, 2048}) {
        run<2, false, true>("row b64 persistent", in, out, total_lines, grid);
        run<4, false, true>("row b128 persistent", in, out, total_lines, grid);
        run<2, true, true>("col b64 persistent", in, out, total_lines, grid);
        run<4, true, true>("col b128 (1 KiB runs of 2 lanes/col) persistent", in, out, total_lines, grid);
    }
    run_coltile<4, true>("whole-column tile, 4 columns (32-B runs), XCD remap", in, out, planes, 128 * 1024);
    run_coltile<4, false>("whole-column tile, 4 columns (32-B runs), no remap", in, out, planes, 128 * 1024);
    run_coltile<4, true>("whole-column tile, 4 columns, 2 blocks/CU (unreal)", in, out, planes, 64 * 1024);
    run_coltile<2, true>("whole-column tile, 2 columns (16-B runs), XCD remap", in, out, planes, 64 * 1024);
    run_coltile<2, false>("whole-column tile, 2 columns (16-B runs), no remap", in, out, planes, 64 * 1024);
    run_coltile<8, true>("whole-column tile, 8 columns (64-B runs; 256 KiB: unreal)", in, out, planes, 128 * 1024);
    run_tile2<1>("2-pass plan, pass 1: 32x32 strided points x 128-B runs, 1 block/CU", in, out, planes, 128 * 1024);
    run_tile2<2>("2-pass plan, pass 2: 128x128 contiguous tile, 1 block/CU", in, out, planes, 128 * 1024);
    run_tile2<1>("2-pass plan, pass 1, 2 blocks/CU (unreal: 64 KiB of LDS)", in, out, planes, 64 * 1024);
    run_tile2<2>("2-pass plan, pass 2, 2 blocks/CU (unreal: 64 KiB of LDS)", in, out, planes, 64 * 1024);
    run<2, false, false>("row b64 one block per line", in, out, total_lines, 0);
    run<4, false, false>("row b128 one block per line", in, out, total_lines, 0);
    run<2, true, false>("col b64 one block per line", in, out, total_lines, 0);
    return 0;
}
