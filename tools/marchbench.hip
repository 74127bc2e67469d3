// marchbench.hip -- the fused chain's memory access shape WITHOUT its arithmetic: waves march over
// bands of rows, one 16-byte buffer load per lane and row, stores `ov` rows behind the loads (the
// chain's stages have a combined reach of 4 rows each way, so a band loads 8 rows it does not
// store).  Sweeps prefetch depth, band height, margin lanes, occupancy and the load/store cache
// policy to find what this shape can reach as a pure copy.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/marchbench tools/marchbench.hip ; run: tools/marchbench
#include "../cudaimageprocessing_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// PF rows in flight per wave; MRG margin lanes per side; LDNT / STNT: aux bits of the loads / stores
// VEC = 2: a lane owns two adjacent 16-byte groups (8 pixels), the wave 2 KiB of a row
template <int PF, int MRG, int LDAUX, int STAUX, int VEC>
__global__ __launch_bounds__(256) void march_copy(const u32* __restrict__ in, u32* __restrict__ out, const Geo g, int ov)
{
    extern __shared__ u32 occupancy_pad[];
    const WaveJob j = wave_job(g);
    if (!j.valid) return;
    constexpr int OUTW = 4 * VEC * (64 - 2 * MRG);
    const March mch = make_march(g, j);
    const int cx0 = j.strip * OUTW - 4 * VEC * MRG;
    const RowIO<1> io(in, out, g, j, mch, cx0, 0);
    const int lane_off = 16 * (VEC - 1) * j.lane;       // RowIO places lane l at 16 l; VEC groups per lane: 16 VEC l
    const int cx = cx0 + 4 * VEC * j.lane;
    const bool productive = j.lane >= MRG && j.lane < 64 - MRG && cx < g.W;
    const u32 ooff = productive ? (u32)(cx * 4) : 0x80000000u;
    auto load = [&](int row, int v) { return __builtin_amdgcn_raw_buffer_load_b128(io.rin, (row - io.row0) * io.pitch + io.in_off + lane_off + 16 * v, 0, LDAUX); };
    auto store = [&](int y, int v, u32x4 x) { __builtin_amdgcn_raw_buffer_store_b128(x, io.rout, (u32)((y - io.ylo) * io.pitch) + ooff + 16 * v, 0, STAUX); };
    u32x4 q[PF][VEC];
#pragma unroll
    for (int p = 0; p < PF; p++)
#pragma unroll
        for (int v = 0; v < VEC; v++) q[p][v] = load(mch.row(p - ov), v);
    const int steps = mch.n + ov;
    for (int base = 0; base < steps; base += PF) {
#pragma unroll
        for (int s = 0; s < PF; s++) {
            const int i = base + s;
            if (i < steps) {
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    const u32x4 cur = q[s][v];
                    q[s][v] = load(mch.row(i + PF - ov), v);
                    store(mch.row(i - ov), v, cur ^ u32x4{1u, 2u, 3u, 4u});
                }
            }
        }
    }
}

// Burst form: every K steps the wave issues the loads of the NEXT K rows together (two register sets), so
// K consecutive rows of its strip are requested at one time; with STB the K stores go out together too.
template <int K, int MRG, int STB>
__global__ __launch_bounds__(256) void march_burst(const u32* __restrict__ in, u32* __restrict__ out, const Geo g, int ov)
{
    extern __shared__ u32 occupancy_pad[];
    const WaveJob j = wave_job(g);
    if (!j.valid) return;
    constexpr int OUTW = 4 * (64 - 2 * MRG);
    const March mch = make_march(g, j);
    const RowIO<1> io(in, out, g, j, mch, j.strip * OUTW - 4 * MRG, MRG);
    auto load = [&](int row) { return __builtin_amdgcn_raw_buffer_load_b128(io.rin, (row - io.row0) * io.pitch + io.in_off, 0, 0); };
    auto store = [&](int y, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, io.rout, (u32)((y - io.ylo) * io.pitch) + io.out_off, 0, 2); };
    u32x4 q[2][K], o[K];
#pragma unroll
    for (int p = 0; p < K; p++) q[0][p] = load(mch.row(p - ov));
    const int steps = mch.n + ov;
    for (int base = 0; base < steps; base += 2 * K) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int b0 = base + h * K;
            if (b0 < steps) {
#pragma unroll
                for (int p = 0; p < K; p++) q[h ^ 1][p] = load(mch.row(b0 + K + p - ov));
#pragma unroll
                for (int p = 0; p < K; p++) {
                    const u32x4 v = q[h][p] ^ u32x4{1u, 2u, 3u, 4u};
                    if (STB) o[p] = v; else if (b0 + p < steps) store(mch.row(b0 + p - ov), v);
                }
                if (STB) {
#pragma unroll
                    for (int p = 0; p < K; p++) if (b0 + p < steps) store(mch.row(b0 + p - ov), o[p]);
                }
            }
        }
    }
}

// Row-synchronous form (round 3): one workgroup of NW waves spans the WHOLE image row (NW x 1 KiB = the row:
// 3840 px = 15 waves), owns a band of rows, and issues its bursts in lockstep -- one s_barrier per burst -- so
// that a burst of K rows is one contiguous run of K x W x 4 bytes (full-width rows are adjacent in memory)
// instead of NW 1-KiB pieces at independent phases.  Blocks are dealt to XCDs so that neighbouring bands share one.
template <int K, int NW, int SYNC>
__global__ __launch_bounds__(64 * NW) void march_rowsync(const u32* __restrict__ in, u32* __restrict__ out, const Geo g, int ov)
{
    extern __shared__ u32 occupancy_pad[];
    const u32 nblk = gridDim.x, b = blockIdx.x;
    const u32 q8 = nblk >> 3, r8 = nblk & 7, xcd = b & 7, idx = b >> 3;
    const u32 logical = g.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx : b;
    WaveJob j;
    j.strip = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    j.band = logical % g.bands;
    j.frame = logical / g.bands;
    j.lane = threadIdx.x & 63;
    j.valid = true;
    const March mch = make_march(g, j);
    const RowIO<1> io(in, out, g, j, mch, j.strip * 256, 0);
    auto load = [&](int row) { return __builtin_amdgcn_raw_buffer_load_b128(io.rin, (row - io.row0) * io.pitch + io.in_off, 0, 0); };
    auto store = [&](int y, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, io.rout, (u32)((y - io.ylo) * io.pitch) + io.out_off, 0, 2); };
    u32x4 q[2][K];
#pragma unroll
    for (int p = 0; p < K; p++) q[0][p] = load(mch.row(p - ov));
    const int steps = mch.n + ov;          // block-uniform
    for (int base = 0; base < steps; base += 2 * K) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int b0 = base + h * K;
            if (b0 < steps) {
                if (SYNC) __syncthreads();
#pragma unroll
                for (int p = 0; p < K; p++) q[h ^ 1][p] = load(mch.row(b0 + K + p - ov));
#pragma unroll
                for (int p = 0; p < K; p++) {
                    const u32x4 v = q[h][p] ^ u32x4{1u, 2u, 3u, 4u};
                    if (b0 + p < steps) store(mch.row(b0 + p - ov), v);
                }
            }
        }
    }
}

struct Cfg { int pf, mrg, ldaux, staux, rpb, ov, blocks_per_cu, remap, vec, W, H, B; };

template <int PF, int MRG, int LDAUX, int STAUX, int VEC>
static float run(const u32* in, u32* out, Geo g, const Cfg& c, int reps)
{
    g.W = c.W; g.H = c.H; g.batch = c.B; g.y_end = c.H;
    g.N = (i64)c.W * c.H; g.in_hi = g.N; g.in_frame_stride = g.N; g.out_frame_stride = g.N;
    const int outw = 4 * VEC * (64 - 2 * MRG);
    g.strips = (g.W + outw - 1) / outw;
    g.rows_per_band = c.rpb;
    g.bands = (g.H + c.rpb - 1) / c.rpb;
    g.total_waves = g.batch * g.bands * g.strips;
    g.xcd_remap = c.remap;
    const int grid = (g.total_waves + 3) / 4;
    const size_t lds = c.blocks_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / c.blocks_per_cu) & ~(size_t)255;
    static bool attr_set = false;
    if (!attr_set) { CK(hipFuncSetAttribute((const void*)march_copy<PF, MRG, LDAUX, STAUX, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 10; i++) march_copy<PF, MRG, LDAUX, STAUX, VEC><<<grid, 256, lds>>>(in, out, g, c.ov);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) march_copy<PF, MRG, LDAUX, STAUX, VEC><<<grid, 256, lds>>>(in, out, g, c.ov);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps;
}

template <int PF, int MRG>
static float run_aux(const u32* in, u32* out, const Geo& g, const Cfg& c, int reps)
{
    if (c.vec == 2) return (c.ldaux == 0 && c.staux == 2) ? run<PF, MRG, 0, 2, 2>(in, out, g, c, reps) : -1.f;
    if (c.ldaux == 0 && c.staux == 2) return run<PF, MRG, 0, 2, 1>(in, out, g, c, reps);
    if (c.ldaux == 2 && c.staux == 2) return run<PF, MRG, 2, 2, 1>(in, out, g, c, reps);
    if (c.ldaux == 0 && c.staux == 0) return run<PF, MRG, 0, 0, 1>(in, out, g, c, reps);
    return -1.f;
}

template <int K, int MRG, int STB>
static float run_burst(const u32* in, u32* out, Geo g, const Cfg& c, int reps)
{
    g.W = c.W; g.H = c.H; g.batch = c.B; g.y_end = c.H;
    g.N = (i64)c.W * c.H; g.in_hi = g.N; g.in_frame_stride = g.N; g.out_frame_stride = g.N;
    const int outw = 4 * (64 - 2 * MRG);
    g.strips = (g.W + outw - 1) / outw;
    g.rows_per_band = c.rpb;
    g.bands = (g.H + c.rpb - 1) / c.rpb;
    g.total_waves = g.batch * g.bands * g.strips;
    g.xcd_remap = c.remap;
    const int grid = (g.total_waves + 3) / 4;
    const size_t lds = c.blocks_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / c.blocks_per_cu) & ~(size_t)255;
    static bool attr_set = false;
    if (!attr_set) { CK(hipFuncSetAttribute((const void*)march_burst<K, MRG, STB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 10; i++) march_burst<K, MRG, STB><<<grid, 256, lds>>>(in, out, g, c.ov);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) march_burst<K, MRG, STB><<<grid, 256, lds>>>(in, out, g, c.ov);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps;
}

template <int K, int NW, int SYNC>
static float run_rowsync(const u32* in, u32* out, Geo g, const Cfg& c, int reps)
{
    g.W = c.W; g.H = c.H; g.batch = c.B; g.y_end = c.H;
    g.N = (i64)c.W * c.H; g.in_hi = g.N; g.in_frame_stride = g.N; g.out_frame_stride = g.N;
    if (c.W != NW * 256) return -1.f;
    g.strips = NW;
    g.rows_per_band = c.rpb;
    g.bands = (g.H + c.rpb - 1) / c.rpb;
    g.total_waves = g.batch * g.bands * g.strips;
    g.xcd_remap = c.remap;
    const int grid = g.batch * g.bands;
    const size_t lds = c.blocks_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / c.blocks_per_cu) & ~(size_t)255;
    CK(hipFuncSetAttribute((const void*)march_rowsync<K, NW, SYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 10; i++) march_rowsync<K, NW, SYNC><<<grid, 64 * NW, lds>>>(in, out, g, c.ov);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) march_rowsync<K, NW, SYNC><<<grid, 64 * NW, lds>>>(in, out, g, c.ov);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps;
}

static float dispatch(const u32* in, u32* out, const Geo& g, const Cfg& c, int reps)
{
    // vec 20 / 21: the row-synchronous form (pf = burst size), with / without the barrier; W = 3840 (15 waves) or 4096 (16)
    if (c.vec == 20 || c.vec == 21) {
#define RC(KK) if (c.pf == KK) { if (c.W == 3840) return c.vec == 20 ? run_rowsync<KK, 15, 1>(in, out, g, c, reps) : run_rowsync<KK, 15, 0>(in, out, g, c, reps); \
                                 if (c.W == 4096) return c.vec == 20 ? run_rowsync<KK, 16, 1>(in, out, g, c, reps) : run_rowsync<KK, 16, 0>(in, out, g, c, reps); }
        RC(1) RC(2) RC(3) RC(4) RC(6) RC(8)
#undef RC
        return -1.f;
    }
    // vec 10 / 11: the burst form (pf = burst size), stores each step / stores in bursts too
    if (c.vec >= 10 && c.mrg == 2) {
#define BC(KK) if (c.pf == KK) return c.vec == 11 ? run_burst<KK, 2, 1>(in, out, g, c, reps) : run_burst<KK, 2, 0>(in, out, g, c, reps);
        BC(2) BC(3) BC(4) BC(6) BC(8)
#undef BC
        return -1.f;
    }
    if (c.vec >= 10 && c.mrg == 0) {
#define BC(KK) if (c.pf == KK) return run_burst<KK, 0, 0>(in, out, g, c, reps);
        BC(2) BC(3) BC(4) BC(6) BC(8)
#undef BC
        return -1.f;
    }
#define CASE(P, M) if (c.pf == P && c.mrg == M) return run_aux<P, M>(in, out, g, c, reps);
    CASE(1, 0) CASE(2, 0) CASE(3, 0) CASE(4, 0) CASE(6, 0)
    CASE(1, 1) CASE(2, 1) CASE(3, 1) CASE(4, 1)
    CASE(1, 2) CASE(2, 2) CASE(3, 2) CASE(4, 2) CASE(6, 2)
#undef CASE
    return -1.f;
}

// Every argument is one configuration: pf,mrg,ldaux,staux,rpb,ov,blocks_per_cu,remap,vec,W,H,B
int main(int argc, char** argv)
{
    const size_t n = (size_t)3840 * 2160 * 16;      // every geometry must hold this many pixels at most
    u32 *in, *out;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipMemset(in, 1, n * 4)); CK(hipMemset(out, 0, n * 4));
    Geo g;
    g.y_begin = 0; g.in_origin = 0; g.in_lo = 0; g.out_origin = 0;
    Cfg warm{1, 2, 0, 2, 30, 8, 6, 1, 1, 3840, 2160, 16};
    dispatch(in, out, g, warm, 600);
    printf("%3s %3s %5s %5s %4s %3s %4s %5s %3s %6s %6s %4s   %8s %8s\n", "pf", "mrg", "ldaux", "staux", "rpb", "ov", "b/cu", "remap", "vec", "W", "H", "B", "ms", "TB/s");
    for (int a = 1; a < argc; a++) {
        Cfg c;
        if (sscanf(argv[a], "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d", &c.pf, &c.mrg, &c.ldaux, &c.staux, &c.rpb, &c.ov, &c.blocks_per_cu,
                   &c.remap, &c.vec, &c.W, &c.H, &c.B) != 12 || (size_t)c.W * c.H * c.B > n) { fprintf(stderr, "bad config %s\n", argv[a]); return 1; }
        const float ms = std::min(dispatch(in, out, g, c, 150), dispatch(in, out, g, c, 150));
        printf("%3d %3d %5d %5d %4d %3d %4d %5d %3d %6d %6d %4d   %8.4f %8.2f\n", c.pf, c.mrg, c.ldaux, c.staux, c.rpb, c.ov, c.blocks_per_cu,
               c.remap, c.vec, c.W, c.H, c.B, ms, (double)c.W * c.H * c.B * 8.0 / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
    return 0;
}
