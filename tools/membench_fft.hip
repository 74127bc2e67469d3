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

// The cols pass of the two-pass plan (csrc/fft_twopass.h) as a pure copy: a tile of NCOLS adjacent columns x
// R2 = 16384 / NCOLS rows of one residue class (DIT: rows RX n2 + n1 in, rows R2 n1 + k2 out; DIF: rows
// R2 k1 + n2 in, rows RX k2 + k1 out), 512 lanes x 32 points, every load issued before the first store,
// persistent blocks (2 per CU) or one block per tile, tiles in launch order or remapped so that an XCD owns
// a contiguous run of tiles, 8- or 16-byte accesses, optional non-temporal stores.
template <int V, int NCOLS, bool DIF, int REMAP, bool NT>
__global__ __launch_bounds__(512, 4) void cols2p_k(const float* __restrict__ in, float* __restrict__ out, int total_tiles)
{
    extern __shared__ float occupancy_pad[];
    typedef float vec __attribute__((ext_vector_type(V)));
    constexpr int R2 = 16384 / NCOLS, RX = 4096 / R2;
    constexpr int LPR = NCOLS * 2 / V;              // lanes per run
    constexpr int TT = 512 / LPR;                   // rows per access round
    constexpr int PTS = R2 / TT;                    // accesses per lane
    const int tid = threadIdx.x, cl = tid % LPR, t = tid / LPR;
    const unsigned nblk = gridDim.x, b = blockIdx.x;
    const unsigned q = nblk >> 3, r8 = nblk & 7, xcd = b & 7, idx = b >> 3;
    const unsigned logical = REMAP ? (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx : b;
    const int per = (total_tiles + nblk - 1) / nblk;
    for (int it = 0; it < per; it++) {
        // REMAP 2: a block owns `per` consecutive tiles (an XCD: a contiguous range of the image)
        const int tile = REMAP == 2 ? (int)logical * per + it : (int)logical + it * (int)nblk;
        if (tile >= total_tiles) break;
        constexpr int CG = 4096 / NCOLS;
        const int cg = tile % CG, u = tile / CG, n1 = u % RX;
        const i64 plane = u / RX;
        const i64 in_base = DIF ? (i64)R2 * n1 : n1, out_base = DIF ? n1 : (i64)R2 * n1;
        constexpr int in_stride = DIF ? 1 : RX, out_stride = DIF ? RX : 1;
        vec v[PTS];
#pragma unroll
        for (int j = 0; j < PTS; j++) {
            const i64 row = plane * 4096 + in_base + (i64)(t + TT * j) * in_stride;
            v[j] = *(const vec*)(in + (row * 4096 + cg * NCOLS) * 2 + cl * V);
        }
#pragma unroll
        for (int j = 0; j < PTS; j++) {
            const i64 row = plane * 4096 + out_base + (i64)(t + TT * j) * out_stride;
            vec* p = (vec*)(out + (row * 4096 + cg * NCOLS) * 2 + cl * V);
            if (NT) __builtin_nontemporal_store(v[j] * 2.0f, p); else *p = v[j] * 2.0f;
        }
    }
}

template <int V, int NCOLS, bool DIF, int REMAP, bool NT>
void run_cols2p(const char* name, const float* in, float* out, int planes, int grid, int lds_bytes)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = planes * 1024;
    const int g = grid > 0 ? grid : tiles;
    hipFuncSetAttribute((const void*)cols2p_k<V, NCOLS, DIF, REMAP, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 3; i++) cols2p_k<V, NCOLS, DIF, REMAP, NT><<<g, 512, lds_bytes>>>(in, out, tiles);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) cols2p_k<V, NCOLS, DIF, REMAP, NT><<<g, 512, lds_bytes>>>(in, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-64s grid %5d lds %6d  %.3f ms  %.0f GB/s  (%s)\n", name, g, lds_bytes, ms, (double)planes * 4096 * 65536.0 / ms / 1e6,
           hipGetErrorString(hipGetLastError()));
}

typedef unsigned int u32b;
typedef u32b u32x2bb __attribute__((ext_vector_type(2)));
// the same with the library kernel's addressing: buffer descriptors, 32-bit lane offset + scalar offset, 8-byte accesses
template <bool DIF, bool PROLOGUE, int GLOBAL = 0>
__global__ __launch_bounds__(512, 4) void cols2p_buf_k(const float* __restrict__ in, float* __restrict__ out, int total_tiles, const float* __restrict__ table)
{
    extern __shared__ float lds_pad_[];
    constexpr int R2 = 1024, RX = 4, NCOL = 16, T = 32;
    const int tid = threadIdx.x, c = tid % NCOL, t = tid / NCOL;
    float twr0 = 0.f, twr1 = 0.f;
    if (PROLOGUE) { twr0 = table[tid * 16]; twr1 = table[(tid + 512) * 16]; }
    const unsigned nblk = gridDim.x, b = blockIdx.x;
    const unsigned q = nblk >> 3, r8 = nblk & 7, xcd = b & 7, idx = b >> 3;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    if (tile >= total_tiles) return;
    const int cg = tile & 255, u = tile >> 8, n1 = u & 3, plane = u >> 2;
    constexpr u32b plane_bytes = 4096u * 4096u * 8u, row_bytes = 4096 * 8;
    constexpr u32b in_stride = DIF ? 1u : RX, out_stride = DIF ? RX : 1u;
    constexpr u32b in_step = T * in_stride * row_bytes, out_step = T * out_stride * row_bytes;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (i64)plane * 4096 * 8192), 0, (int)plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (i64)plane * 4096 * 8192), 0, (int)plane_bytes, 0x00020000);
    const u32b in_base = DIF ? R2 * n1 : n1, out_base = DIF ? n1 : R2 * n1;
    const u32b col_bytes = (u32b)(cg * NCOL + c) * 8u;
    const u32b voff_in = (in_base + (u32b)t * in_stride) * row_bytes + col_bytes;
    const u32b voff_out = (out_base + (u32b)t * out_stride) * row_bytes + col_bytes;
    u32x2bb v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        if (GLOBAL & 1) v[j] = *(const u32x2bb*)((const char*)(in + (i64)plane * 4096 * 8192) + voff_in + j * in_step);
        else v[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, voff_in, j * in_step, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (PROLOGUE) { lds_pad_[tid] = twr0; lds_pad_[tid + 512] = twr1; __syncthreads(); }
#pragma unroll
    for (int j = 0; j < 32; j++) {
        if (GLOBAL & 2) *(u32x2bb*)((char*)(out + (i64)plane * 4096 * 8192) + voff_out + j * out_step) = v[j];
        else __builtin_amdgcn_raw_buffer_store_b64(v[j], rout, voff_out, j * out_step, 0);
    }
}

template <bool DIF, bool PROLOGUE, int GLOBAL = 0>
void run_cols2p_buf(const char* name, const float* in, float* out, int planes, int lds_bytes)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = planes * 1024;
    hipFuncSetAttribute((const void*)cols2p_buf_k<DIF, PROLOGUE, GLOBAL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 3; i++) cols2p_buf_k<DIF, PROLOGUE, GLOBAL><<<tiles, 512, lds_bytes>>>(in, out, tiles, in);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) cols2p_buf_k<DIF, PROLOGUE, GLOBAL><<<tiles, 512, lds_bytes>>>(in, out, tiles, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-64s grid %5d lds %6d  %.3f ms  %.0f GB/s  (%s)\n", name, tiles, lds_bytes, ms, (double)planes * 4096 * 65536.0 / ms / 1e6,
           hipGetErrorString(hipGetLastError()));
}

// persistent form of cols2p_buf_k: a block owns `per` consecutive tiles (XCD-contiguous blocks), bulk loads, bulk stores
template <int SB>
__global__ __launch_bounds__(512, 4) void cols2p_persist_k(const float* __restrict__ in, float* __restrict__ out, int total_tiles)
{
    extern __shared__ float lds_pad_[];
    constexpr int R2 = 1024, RX = 4, NCOL = 16, T = 32;
    const int tid = threadIdx.x, c = tid % NCOL, t = tid / NCOL;
    const unsigned nblk = gridDim.x, b = blockIdx.x;
    const unsigned q = nblk >> 3, r8 = nblk & 7, xcd = b & 7, idx = b >> 3;
    const int logical = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int per = (total_tiles + nblk - 1) / nblk;
    constexpr u32b plane_bytes = 4096u * 4096u * 8u, row_bytes = 4096 * 8;
    constexpr u32b in_step = T * RX * row_bytes, out_step = T * row_bytes;
    for (int it = 0; it < per; it++) {
        const int tile = logical * per + it;
        if (tile >= total_tiles) break;
        const int cg = tile & 255, u = tile >> 8, n1 = u & 3, plane = u >> 2;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (i64)plane * 4096 * 8192), 0, (int)plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (i64)plane * 4096 * 8192), 0, (int)plane_bytes, 0x00020000);
        const u32b col_bytes = (u32b)(cg * NCOL + c) * 8u;
        const u32b voff_in = ((u32b)n1 + (u32b)t * RX) * row_bytes + col_bytes;
        const u32b voff_out = ((u32b)R2 * n1 + (u32b)t) * row_bytes + col_bytes;
        u32x2bb v[32];
#pragma unroll
        for (int j = 0; j < 32; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, voff_in, j * in_step, 0);
        if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 32; j++) __builtin_amdgcn_raw_buffer_store_b64(v[j], rout, voff_out, j * out_step, 0);
        if (SB) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int SB>
void run_cols2p_persist(const char* name, const float* in, float* out, int planes, int grid, int lds_bytes)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = planes * 1024;
    hipFuncSetAttribute((const void*)cols2p_persist_k<SB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 3; i++) cols2p_persist_k<SB><<<grid, 512, lds_bytes>>>(in, out, tiles);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) cols2p_persist_k<SB><<<grid, 512, lds_bytes>>>(in, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-64s grid %5d lds %6d  %.3f ms  %.0f GB/s  (%s)\n", name, grid, lds_bytes, ms, (double)planes * 4096 * 65536.0 / ms / 1e6,
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
    const int planes = getenv("PLANES") ? atoi(getenv("PLANES")) : 8;                       // 8 x 128 MiB in + out
    const int total_lines = planes * 4096;
    float *in, *out;
    hipMalloc(&in, (size_t)total_lines * 32768);
    hipMalloc(&out, (size_t)total_lines * 32768);
    hipMemset(in, 0, (size_t)total_lines * 32768);
    if (getenv("COLS2P")) {
        const int L = 76 * 1024;
        const int P = getenv("PLANES") ? atoi(getenv("PLANES")) : planes;
        run_cols2p_persist<1>("persistent, block-contiguous tiles, bulk loads then bulk stores", in, out, P, 512, L);
        run_cols2p_persist<0>("persistent, block-contiguous tiles, compiler's order", in, out, P, 512, L);
        run_cols2p_persist<1>("persistent bulk, 1024 blocks", in, out, P, 1024, L);
        run_cols2p_persist<1>("persistent bulk, 2048 blocks", in, out, P, 2048, L);
        run_cols2p_buf<false, false>("buffer ops, b64 16 cols DIT, one block per tile, XCD-contiguous", in, out, P, L);
        run_cols2p_buf<false, true>("buffer ops + table prologue + barrier, same", in, out, P, L);
        run_cols2p_buf<false, true, 1>("global loads, buffer stores, same", in, out, P, L);
        run_cols2p_buf<false, true, 2>("buffer loads, global stores, same", in, out, P, L);
        run_cols2p_buf<false, true, 3>("global loads, global stores, same", in, out, P, L);
        run_cols2p_buf<false, false, 3>("global loads, global stores, no prologue", in, out, P, L);
        run_cols2p_buf<true, true>("buffer ops + table prologue + barrier, DIF", in, out, P, L);
        run_cols2p<2, 16, false, 0, false>("b64 16 cols DIT persistent (the kernel's shape)", in, out, P, 512, L);
        run_cols2p<2, 16, false, 1, false>("b64 16 cols DIT persistent, XCD-contiguous", in, out, P, 512, L);
        run_cols2p<2, 16, false, 2, false>("b64 16 cols DIT persistent, block-contiguous tiles", in, out, P, 512, L);
        run_cols2p<4, 16, false, 2, false>("b128 16 cols DIT persistent, block-contiguous tiles", in, out, P, 512, L);
        run_cols2p<2, 16, false, 0, false>("b64 16 cols DIT, one block per tile", in, out, P, 0, L);
        run_cols2p<2, 16, false, 1, false>("b64 16 cols DIT, one block per tile, XCD-contiguous", in, out, P, 0, L);
        run_cols2p<2, 16, false, 1, true>("b64 16 cols DIT, one block per tile, XCD-contiguous, nt", in, out, P, 0, L);
        run_cols2p<4, 16, false, 1, false>("b128 16 cols DIT, one block per tile, XCD-contiguous", in, out, P, 0, L);
        run_cols2p<4, 16, false, 1, true>("b128 16 cols DIT, one block per tile, XCD-contiguous, nt", in, out, P, 0, L);
        run_cols2p<4, 16, true, 1, false>("b128 16 cols DIF, one block per tile, XCD-contiguous", in, out, P, 0, L);
        run_cols2p<2, 16, true, 1, false>("b64 16 cols DIF, one block per tile, XCD-contiguous", in, out, P, 0, L);
        run_cols2p<4, 32, false, 1, false>("b128 32 cols x 512 DIT, one block per tile, XCD-contiguous", in, out, P, 0, L);
        run_cols2p<2, 32, false, 1, false>("b64 32 cols x 512 DIT, one block per tile, XCD-contiguous", in, out, P, 0, L);
        run_cols2p<4, 64, false, 1, false>("b128 64 cols x 256 DIT, one block per tile, XCD-contiguous", in, out, P, 0, L);
        return 0;
    }
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
