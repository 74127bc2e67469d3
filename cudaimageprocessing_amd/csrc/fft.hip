// fft.hip -- 2-D complex fp32 DFT, unnormalised, forward (-1) or inverse (+1): what the
// reference asks of cuFFT (cufftPlan2d + cufftExecC2C, kernels/imageTools.cu:687-701, 741-751).
//
// A 2-D transform is a pass over the rows then a pass over the columns.  Each pass transforms
// whole lines inside LDS: a line of N = 2^L points is owned by N/16 lanes, 16 points per lane,
// and runs a Stockham autosort FFT whose stages are radix 16 (the last one radix 2, 4 or 8 when
// L is not a multiple of 4).  A stage reads points t + (N/16)*k (k = 0..15) -- the same
// positions in every stage, lane-contiguous -- and scatters its results; the first stage reads
// from global memory and the last one writes to global memory, so a 4096-point line crosses LDS
// only twice.  The column pass works on LPB adjacent columns per block so that its global
// accesses are LPB*8-byte runs; a column of 1024 or more points, of which only a few fit LDS
// side by side, is split (out of place) into two passes of short transforms over 64 adjacent
// columns each -- see launch_fft2d.  Blocks are persistent and ask for their next lines before
// transforming the current ones.  Twiddles come from a table in device memory computed once
// per device in double precision.  The inverse is conj(FFT(conj(x))).
#include <mutex>

#include "common.h"
#include "internal.h"

namespace mi {

struct c2 { float x, y; };

MI_DEV c2 cadd(c2 a, c2 b) { return c2{a.x + b.x, a.y + b.y}; }
MI_DEV c2 csub(c2 a, c2 b) { return c2{a.x - b.x, a.y - b.y}; }
MI_DEV c2 cmul(c2 a, c2 b) { return c2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
MI_DEV c2 mul_neg_i(c2 a) { return c2{a.y, -a.x}; }   // a * (-i)

constexpr int TW_N = 16384;
__device__ c2 g_twiddle[TW_N];   // exp(-2*pi*i*n/16384)

__global__ void twiddle_init_kernel()
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < TW_N) {
        double s, c;
        sincospi(-2.0 * (double)n / (double)TW_N, &s, &c);
        g_twiddle[n] = c2{(float)c, (float)s};
    }
}

// ---------------------------------------------------------------- small forward DFTs, in place
MI_DEV void dft2(c2& a, c2& b)
{
    const c2 s = cadd(a, b), d = csub(a, b);
    a = s; b = d;
}

MI_DEV void dft4(c2& x0, c2& x1, c2& x2, c2& x3)
{
    const c2 s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = mul_neg_i(csub(x1, x3));
    x0 = cadd(s02, s13); x1 = cadd(d02, d13); x2 = csub(s02, s13); x3 = csub(d02, d13);
}

// X[q] for q = 0..R-1 from x[0..R-1]; both arrays are indexed with compile-time constants.
template <int R>
MI_DEV void dft(c2 (&x)[R]);

template <>
MI_DEV void dft<2>(c2 (&x)[2]) { dft2(x[0], x[1]); }

template <>
MI_DEV void dft<4>(c2 (&x)[4]) { dft4(x[0], x[1], x[2], x[3]); }

template <>
MI_DEV void dft<8>(c2 (&x)[8])
{   // n = 4*n1 + n2, k = k1 + 2*k2
    constexpr float h = 0.70710678118654752440f;
    c2 a[4][2];
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
        a[n2][0] = cadd(x[n2], x[n2 + 4]);
        a[n2][1] = csub(x[n2], x[n2 + 4]);
    }
    a[1][1] = cmul(a[1][1], c2{h, -h});
    a[2][1] = mul_neg_i(a[2][1]);
    a[3][1] = cmul(a[3][1], c2{-h, -h});
#pragma unroll
    for (int k1 = 0; k1 < 2; k1++) {
        dft4(a[0][k1], a[1][k1], a[2][k1], a[3][k1]);
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) x[k1 + 2 * k2] = a[k2][k1];
    }
}

template <>
MI_DEV void dft<16>(c2 (&x)[16])
{   // n = 4*n1 + n2, k = k1 + 4*k2;  W16^m = (cos(pi m/8), -sin(pi m/8))
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    c2 a[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
        a[n2][0] = x[n2]; a[n2][1] = x[n2 + 4]; a[n2][2] = x[n2 + 8]; a[n2][3] = x[n2 + 12];
        dft4(a[n2][0], a[n2][1], a[n2][2], a[n2][3]);   // over n1 -> k1
    }
    // twiddles W16^(n2*k1)
    a[1][1] = cmul(a[1][1], c2{c1, -s1}); a[1][2] = cmul(a[1][2], c2{h, -h});   a[1][3] = cmul(a[1][3], c2{s1, -c1});
    a[2][1] = cmul(a[2][1], c2{h, -h});   a[2][2] = mul_neg_i(a[2][2]);         a[2][3] = cmul(a[2][3], c2{-h, -h});
    a[3][1] = cmul(a[3][1], c2{s1, -c1}); a[3][2] = cmul(a[3][2], c2{-h, -h});  a[3][3] = cmul(a[3][3], c2{-c1, s1});
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
        dft4(a[0][k1], a[1][k1], a[2][k1], a[3][k1]);   // over n2 -> k2
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) x[k1 + 4 * k2] = a[k2][k1];
    }
}

// ---------------------------------------------------------------- line FFT in LDS
// A pass transforms `total_lines` lines.  Line L = (q, c) with c = L % inner, q = L / inner,
// q = (plane, g) with g = q % groups.  Its points sit at
//   in :  plane*plane_stride + g*in_group_stride  + c*line_stride + i*in_elem_stride
//   out:  plane*plane_stride + g*out_group_stride + c*line_stride + k*out_elem_stride
// and, when tw_step != 0, output k is multiplied by exp(-2 pi i * g*k*tw_step / 16384) (the
// twiddle between the two halves of a split column transform).
//   row pass:        inner = 1 line per "c", groups = rows: g*cols, elem stride 1
//   column pass:     inner = cols, groups = 1, elem stride cols
//   split column pass A/B: inner = cols, groups = R2 or R1 (see launch_fft2d)
struct FftGeo {
    int total_lines;       // < 2^31 per launch (launch_fft2d transforms planes in groups)
    int inner_log2;        // inner and groups are powers of two
    int groups_log2;
    i64 plane_stride;
    i64 line_stride;
    i64 in_group_stride, out_group_stride;
    int in_elem_stride, out_elem_stride;   // a line spans < 2^31 elements
    int tw_step;
    int tw_uniform;        // all lines of a block iteration share their group index (inner >= lines per block)
    int inverse;
};

struct LineAddr { i64 in, out; int g; bool active; };

MI_DEV LineAddr line_addr(const FftGeo& g, int L)
{
    LineAddr a;
    a.active = L < g.total_lines;
    const u32 Lc = a.active ? (u32)L : 0u;
    const u32 c = Lc & ((1u << g.inner_log2) - 1u), q = Lc >> g.inner_log2;
    const i64 plane = q >> g.groups_log2;
    a.g = (int)(q & ((1u << g.groups_log2) - 1u));
    const i64 common = plane * g.plane_stride + (i64)c * g.line_stride;
    a.in = common + a.g * g.in_group_stride;
    a.out = common + a.g * g.out_group_stride;
    return a;
}

MI_DEV int lds_pad(int i) { return i + (i >> 4); }   // one spare slot per 16: breaks the 128-byte stride

// Stage plan of an N-point line: radix-16 stages, then one radix-2/4/8 stage when log2 N is not a
// multiple of 4.  The twiddles of stage s (all but the first stage have some) sit in LDS as a
// table [(q-1) * NS + ktw] = exp(-2 pi i q ktw / (NS R)), q = 1..R-1, ktw = 0..NS-1, filled once
// per block: inside the line loop the only vector-memory traffic is then the lines themselves, so
// the in-order vmcnt counter never makes a stage wait for the prefetch of the next line.
// A 16384-point line fills 136 KiB of LDS by itself: only its first table (240 entries) joins it
// there, the later stages read the device table (L2 hits; such blocks do not prefetch).
template <int N>
struct Plan {
    static constexpr int L = __builtin_ctz(N);
    static constexpr int A = L / 4;                  // radix-16 stages
    static constexpr int REM = 1 << (L % 4);         // trailing radix (1 = none)
    static constexpr int n_stages = A + (REM > 1 ? 1 : 0);
    static constexpr int radix(int s) { return s < A ? 16 : REM; }
    static constexpr int ns(int s) { int p = 1; for (int i = 0; i < s; i++) p *= radix(i); return p; }
    static constexpr int tw_off(int s) { int o = 0; for (int i = 1; i < s; i++) o += (radix(i) - 1) * ns(i); return o; }
    static constexpr int lds_tw_stages = N > 8192 ? 2 : n_stages;   // stages 1..lds_tw_stages-1 have their table in LDS
    static constexpr int tw_total = tw_off(lds_tw_stages);
};

// One Stockham stage of radix R over this lane's 16 points (N/16 lanes per line).
// NS = product of the radices of the earlier stages; tws = this stage's twiddle table.
template <int N, int R, int NS, bool LAST, bool TW_LDS>
MI_DEV void fft_stage(c2 (&v)[16], c2* __restrict__ line, const c2* __restrict__ tws, int t)
{
    constexpr int T = N / 16, G = 16 / R;
#pragma unroll
    for (int m = 0; m < G; m++) {
        const int j = t + T * m;
        const int ktw = j % NS;
        c2 x[R];
#pragma unroll
        for (int q = 0; q < R; q++) x[q] = v[m + q * G];
        if (NS > 1) {
#pragma unroll
            for (int q = 1; q < R; q++) x[q] = cmul(x[q], TW_LDS ? tws[(q - 1) * NS + ktw] : g_twiddle[q * ktw * (TW_N / (NS * R))]);
        }
        dft<R>(x);
        if (LAST) {
#pragma unroll
            for (int q = 0; q < R; q++) v[m + q * G] = x[q];
        } else {
            const int d = (j / NS) * (NS * R) + ktw;
#pragma unroll
            for (int q = 0; q < R; q++) line[lds_pad(d + q * NS)] = x[q];
        }
    }
    if (!LAST) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = line[lds_pad(t + T * k)];
        __syncthreads();
    }
}

template <int N, int S = 0>
MI_DEV void fft_line(c2 (&v)[16], c2* __restrict__ line, const c2* __restrict__ tw, int t)
{
    using P = Plan<N>;
    if constexpr (S < P::n_stages) {
        fft_stage<N, P::radix(S), P::ns(S), S == P::n_stages - 1, (S < P::lds_tw_stages)>(v, line, tw + (S < P::lds_tw_stages ? P::tw_off(S) : 0), t);
        fft_line<N, S + 1>(v, line, tw, t);
    }
}

// Fills the block's twiddle tables from the device table (all threads; caller synchronises).
template <int N, int S = 1>
MI_DEV void fill_twiddles(c2* __restrict__ tw, int tid, int nthreads)
{
    using P = Plan<N>;
    if constexpr (S < P::lds_tw_stages) {
        constexpr int R = P::radix(S), NS = P::ns(S), step = TW_N / (NS * R);
        for (int idx = tid; idx < (R - 1) * NS; idx += nthreads) {
            const int q = idx / NS + 1, k = idx % NS;
            tw[P::tw_off(S) + idx] = g_twiddle[q * k * step];
        }
        fill_twiddles<N, S + 1>(tw, tid, nthreads);
    }
}

// COL = false: lanes of a line are adjacent (lane-contiguous points of a row).
// COL = true : LPB adjacent lines (columns) are adjacent lanes (runs of LPB*8 bytes per point).
// A block walks lines blockIdx.x*LPB, += gridDim.x*LPB ... and asks for the next group's points
// before it transforms the current ones, so its HBM loads stay in flight during the LDS phases.
typedef float f32x2b __attribute__((ext_vector_type(2)));
typedef u32 u32x2b __attribute__((ext_vector_type(2)));

// Global accesses go through buffer descriptors: the per-lane part of an address is one 32-bit
// byte offset and the per-point part (k * T * stride) a scalar offset, so no lane holds sixteen
// 64-bit addresses; lanes of an inactive line get an out-of-range offset (loads return 0,
// stores are dropped).  A launch therefore covers less than 2 GiB (launch_fft2d groups planes).
// PREFETCH: hold the next group's points in registers while transforming the current ones
// (costs 32 registers: 2 blocks of 256 lanes per SIMD set instead of 4).
template <int N, int LPB, bool COL, bool TW, bool PREFETCH>
__global__ __launch_bounds__((N / 16) * LPB, ((N / 16) * LPB >= 1024) ? 4 : (PREFETCH ? 2 : 3))
void fft_lines_kernel(const c2* __restrict__ in, c2* __restrict__ out, const FftGeo g, u32 bytes)
{
    constexpr int T = N / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    c2* lds = reinterpret_cast<c2*>(lds_raw);
    const int tid = threadIdx.x;
    const int lin = COL ? tid % LPB : tid / T;
    const int t = COL ? tid / LPB : tid % T;
    c2* line = lds + lin * (N + N / 16);
    c2* tw = lds + LPB * (N + N / 16);
    fill_twiddles<N>(tw, tid, T * LPB);
    __syncthreads();
    const int step = (int)gridDim.x * LPB;
    const int is = COL ? g.in_elem_stride : 1, os = COL ? g.out_elem_stride : 1;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)bytes, 0x00020000);

    auto in_off = [&](const LineAddr& a) { return a.active ? (u32)((a.in + (i64)t * is) * 8) : 0x80000000u; };
    auto load16 = [&](u32 voff, c2 (&d)[16]) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const f32x2b r = __builtin_bit_cast(f32x2b, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(rin, voff, (T * k) * is * 8, 0));
            d[k] = c2{r.x, r.y};
        }
    };

    // Inter-pass twiddles of a split column transform: output kk of a line of group gi is
    // multiplied by exp(-2 pi i gi kk tw_step / 16384).  When all lines of a block iteration share
    // gi (cols >= LPB: the usual case) the N factors go through a double-buffered LDS table that
    // lanes tid < N fill one iteration ahead -- their single load is issued before the prefetch
    // of that iteration's lines, so nothing in the loop waits for anything younger than itself.
    c2* twg = tw + Plan<N>::tw_total;                // 2 * N entries (TW kernels only)
    // the prefetching variant is only launched for the shared-group case (launch_lines_tw)
    const bool tw_shared = TW && (PREFETCH || g.tw_uniform);
    auto tw_fetch = [&](const LineAddr& a) {
        c2 w{1.f, 0.f};
        if (TW && tw_shared && tid < N && a.active) w = g_twiddle[(a.g * tid * g.tw_step) & (TW_N - 1)];
        return w;
    };

    // transform the 16 points in v (loaded from line `a`) and store them; `par` = table parity
    auto process = [&](c2 (&v)[16], const LineAddr& a, c2 wnext, int par) {
        if (g.inverse) {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k].y = -v[k].y;
        }
        if (TW && tw_shared && tid < N) twg[par * N + tid] = wnext;
        if (TW && Plan<N>::n_stages == 1) __syncthreads();     // otherwise the stage barriers order it
        fft_line<N>(v, line, tw, t);
        const u32 ooff = a.active ? (u32)((a.out + (i64)t * os) * 8) : 0x80000000u;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            c2 r = v[k];
            if (TW) r = cmul(r, tw_shared ? twg[par * N + t + T * k] : g_twiddle[(a.g * (t + T * k) * g.tw_step) & (TW_N - 1)]);
            if (g.inverse) r.y = -r.y;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2b, f32x2b{r.x, r.y}), rout, ooff, (T * k) * os * 8, 0);
        }
    };

    int B = (int)blockIdx.x * LPB;                 // first line of this block's current group
    if (B >= g.total_lines) return;
    if constexpr (PREFETCH) {
        // Two register sets used alternately: while one line is transformed in place in `va`, the
        // next one arrives in `vb`, and vice versa -- no copy ever reads a register that a load in
        // flight is still to write (a copy would make the wave wait for its own prefetch).
        // All trip counts are block-uniform, so the barriers inside are safe.
        c2 va[16], vb[16];
        LineAddr a0 = line_addr(g, B + lin);
        c2 w0 = tw_fetch(a0);
        load16(in_off(a0), va);
        int par = 0;
        while (true) {
            const bool more1 = B + step < g.total_lines && B + step > 0;
            const LineAddr a1 = line_addr(g, B + step + lin);
            const c2 w1 = tw_fetch(a1);
            if (more1) load16(in_off(a1), vb);
            process(va, a0, w0, par);
            if (!more1) break;
            B += step;
            const bool more2 = B + step < g.total_lines && B + step > 0;
            const LineAddr a2 = line_addr(g, B + step + lin);
            w0 = tw_fetch(a2);
            if (more2) load16(in_off(a2), va);
            process(vb, a1, w1, par ^ 1);
            if (!more2) break;
            B += step;
            a0 = a2;
        }
    } else {
        int par = 0;
        for (; B < g.total_lines && B >= 0; B += step, par ^= 1) {
            const LineAddr cur = line_addr(g, B + lin);
            c2 v[16];
            const c2 w = tw_fetch(cur);
            load16(in_off(cur), v);
            process(v, cur, w, par);
        }
    }
}

// Lines of 2, 4 or 8 points: one lane per line, all in registers.
template <int N>
__global__ __launch_bounds__(256) void fft_small_kernel(const c2* __restrict__ in,
                                                        c2* __restrict__ out, const FftGeo g)
{
    const i64 L64 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (L64 >= g.total_lines) return;
    const LineAddr a = line_addr(g, (int)L64);
    if (!a.active) return;
    c2 x[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        x[k] = in[a.in + k * g.in_elem_stride];
        if (g.inverse) x[k].y = -x[k].y;
    }
    dft<N>(x);
#pragma unroll
    for (int k = 0; k < N; k++) {
        c2 r = x[k];
        if (g.tw_step) r = cmul(r, g_twiddle[(a.g * k * g.tw_step) & (TW_N - 1)]);
        if (g.inverse) r.y = -r.y;
        out[a.out + k * g.out_elem_stride] = r;
    }
}

}  // namespace mi

#include "fft_twopass.h"

namespace mi {

// ---------------------------------------------------------------- host side
// The twiddle table of the current device is filled once, by the first FFT call on that device, and
// that call blocks until the fill has completed before any stream is told the table exists: a first
// call on another stream or thread either waits on the mutex or finds a finished table (it used to
// be marked ready as soon as the fill was enqueued, and a second stream could read it half written).
// Under stream capture nothing may block and the fill would only be recorded, so it is recorded
// with that graph (every replay fills the table again, harmlessly) and the table stays "not ready".
static int ensure_twiddles(hipStream_t s)
{
    static std::mutex mu;
    static bool ready[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_status(e, "hipGetDevice");
    if (dev < 0 || dev >= 64) return fail(MI_ERR_UNSUPPORTED, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (ready[dev]) return MI_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    twiddle_init_kernel<<<TW_N / 256, 256, 0, s>>>();
    e = hipGetLastError();
    if (e != hipSuccess) return hip_status(e, "twiddle table init");
    if (cap != hipStreamCaptureStatusNone) return MI_OK;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_status(e, "twiddle table init");
    ready[dev] = true;
    return MI_OK;
}

// hipFuncSetAttribute and the occupancy query are per device: once per (kernel, device).
struct PerDeviceAttr {
    std::mutex mu;
    bool done[64] = {};
    hipError_t err[64] = {};
    int blocks_per_cu[64] = {};
    template <class Init>
    hipError_t get(int* blocks, Init init)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lock(mu);
        if (!done[dev]) {
            blocks_per_cu[dev] = 1;
            err[dev] = init(&blocks_per_cu[dev]);
            done[dev] = true;
        }
        if (blocks) *blocks = blocks_per_cu[dev];
        return err[dev];
    }
};

template <int N, int LPB, bool COL, bool TW, bool PREFETCH>
static int launch_lines_pf(const c2* in, c2* out, const FftGeo& g, u32 bytes, hipStream_t s)
{
    constexpr int threads = (N / 16) * LPB;
    constexpr size_t lds = ((size_t)LPB * (N + N / 16) + Plan<N>::tw_total + (TW ? 2 * N : 0)) * sizeof(c2);
    static PerDeviceAttr attr;
    static_assert(lds <= 160 * 1024, "a block's lines and twiddle tables must fit the CU's LDS");
    int blocks_per_cu = 1;
    const hipError_t attr_err = attr.get(&blocks_per_cu, [](int* blocks) {
        hipError_t e = hipSuccess;
        if (lds > 64 * 1024)
            e = hipFuncSetAttribute((const void*)fft_lines_kernel<N, LPB, COL, TW, PREFETCH>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int n = 0;
        if (e == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fft_lines_kernel<N, LPB, COL, TW, PREFETCH>, threads, lds) == hipSuccess && n > 0)
            *blocks = n;
        return e;
    });
    if (attr_err != hipSuccess) return hip_status(attr_err, "hipFuncSetAttribute(max dynamic LDS)");
    const i64 blocks = ((i64)g.total_lines + LPB - 1) / LPB;
    // persistent blocks: every CU keeps its full complement, each walks several line groups
    const i64 resident = (i64)256 * blocks_per_cu * std::max(1, tuning().fft_waves);
    const int grid = (int)std::min<i64>(blocks, tuning().fft_persistent ? resident : blocks);
    if (blocks > 0x7fffffff) return fail(MI_ERR_UNSUPPORTED, "fft launch too large");
    fft_lines_kernel<N, LPB, COL, TW, PREFETCH><<<grid, threads, lds, s>>>(in, out, g, bytes);
    return hip_status(hipGetLastError(), "fft kernel launch");
}

template <int N, int LPB, bool COL, bool TW>
static int launch_lines_tw(const c2* in, c2* out, const FftGeo& g, u32 bytes, hipStream_t s)
{
    // 1024-lane blocks have 128 registers per lane: no room to hold the next group's points
    if constexpr ((N / 16) * LPB <= 512) {
        if (tuning().fft_prefetch && (!TW || g.tw_uniform)) return launch_lines_pf<N, LPB, COL, TW, true>(in, out, g, bytes, s);
    }
    return launch_lines_pf<N, LPB, COL, TW, false>(in, out, g, bytes, s);
}

template <int N, int LPB, bool COL>
static int launch_lines(const c2* in, c2* out, const FftGeo& g, u32 bytes, hipStream_t s)
{
    // the inter-pass twiddle is only ever applied by the short column transforms of a split pass
    if constexpr (COL && N <= 256) {     // R1 = rows / 64 <= 256
        if (g.tw_step) return launch_lines_tw<N, LPB, COL, true>(in, out, g, bytes, s);
    }
    return launch_lines_tw<N, LPB, COL, false>(in, out, g, bytes, s);
}

template <int N>
static int launch_small(const c2* in, c2* out, const FftGeo& g, hipStream_t s)
{
    const i64 blocks = ((i64)g.total_lines + 255) / 256;
    if (blocks > 0x7fffffff) return fail(MI_ERR_UNSUPPORTED, "fft launch too large");
    fft_small_kernel<N><<<(int)blocks, 256, 0, s>>>(in, out, g);
    return hip_status(hipGetLastError(), "fft kernel launch");
}

// Row pass: as many whole lines per 256-lane block as fit.  Column pass: as many adjacent
// columns per block as 1024 lanes and 160 KiB of LDS allow, for the widest global runs.
template <bool COL>
static int launch_pass(int n, const c2* in, c2* out, const FftGeo& g, u32 bytes, hipStream_t s)
{
    switch (n) {
    case 2: return launch_small<2>(in, out, g, s);
    case 4: return launch_small<4>(in, out, g, s);
    case 8: return launch_small<8>(in, out, g, s);
    case 16: return launch_lines<16, 64, COL>(in, out, g, bytes, s);
    case 32: return launch_lines<32, 64, COL>(in, out, g, bytes, s);
    case 64: return launch_lines<64, 64, COL>(in, out, g, bytes, s);
    case 128: return launch_lines<128, 32, COL>(in, out, g, bytes, s);
    case 256: return launch_lines<256, COL ? 32 : 16, COL>(in, out, g, bytes, s);
    case 512: return launch_lines<512, COL ? 32 : 8, COL>(in, out, g, bytes, s);
    case 1024: return launch_lines<1024, COL ? 16 : 4, COL>(in, out, g, bytes, s);
    case 2048: return launch_lines<2048, COL ? 8 : 2, COL>(in, out, g, bytes, s);
    case 4096: return launch_lines<4096, COL ? 2 : 1, COL>(in, out, g, bytes, s);   // 2 lines + 32 KiB of twiddles
    case 8192: return launch_lines<8192, 1, COL>(in, out, g, bytes, s);              // 1 line + 64 KiB of twiddles
    case 16384: return launch_lines<16384, 1, COL>(in, out, g, bytes, s);            // 1 line of 136 KiB
    default: return fail(MI_ERR_UNSUPPORTED, "fft line length %d", n);
    }
}

// ---- the two-pass plan (fft_twopass.h)
template <auto kernel>
static int twopass_launch(int threads, size_t lds, const c2* in, c2* out, const TwoPassGeo& g, i64 ntiles, hipStream_t s)
{
    static PerDeviceAttr attr;          // per kernel (and device): the LDS attribute and the occupancy
    int blocks_per_cu = 1;
    const hipError_t attr_err = attr.get(&blocks_per_cu, [&](int* blocks) {
        hipError_t e = hipSuccess;
        if (lds > 64 * 1024) e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int n = 0;
        if (e == hipSuccess && hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) == hipSuccess && n > 0) *blocks = n;
        return e;
    });
    if (attr_err != hipSuccess) return hip_status(attr_err, "hipFuncSetAttribute(max dynamic LDS)");
    if (ntiles > 0x7fffffff) return fail(MI_ERR_UNSUPPORTED, "fft launch too large");
    kernel<<<(int)ntiles, threads, lds, s>>>(in, out, g);       // one tile per block
    return hip_status(hipGetLastError(), "fft kernel launch");
}

static TwoPassGeo twopass_geo(int rows, int cols, int planes, int dir)
{
    TwoPassGeo g{};
    g.planes = planes; g.plane_stride = (i64)rows * cols; g.inverse = dir > 0;
    g.xcd_remap = tuning().fft_twopass_remap; g.nt_stores = 0;
    return g;
}

template <int NC, int RX, bool DIF, int IO = IO_COMPLEX>
static int twopass_rows(const c2* in, c2* out, const TwoPassGeo& g, hipStream_t s)
{
    constexpr int R2 = 1024;
    constexpr size_t lds = ((size_t)2 * (NC + NC / 16) + Plan8<NC>::tw_total) * sizeof(c2);
    static_assert(lds <= 80 * 1024, "at least two blocks per CU");
    const i64 units = IO == IO_U8_OUT ? g.planes / 3 : g.planes;
    return twopass_launch<fft2p_rows_kernel<NC, RX, R2, DIF, false, IO>>(NC / 8, lds, in, out, g, units * R2, s);
}

template <int RX, int NC, bool DIF, bool GAIN = false>
static int twopass_cols(const c2* in, c2* out, const TwoPassGeo& g, hipStream_t s)
{
    constexpr int R2 = 1024, NCOL = 512 / (R2 / 32);
    constexpr size_t lds = (size_t)(R2 + R2 / 32) * NCOL * 4 + (size_t)Plan32<R2>::tw_total * sizeof(c2);
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    return twopass_launch<fft2p_cols_kernel<R2, RX, NC, DIF, false, GAIN>>(512, lds, in, out, g, (i64)g.planes * RX * (NC / NCOL), s);
}

// Sizes the two-pass plan is built for: 1024-point column tiles, so rows = RX * 1024 with RX = 1, 2 or 4 (RX rows of
// `cols` points = at most 16 Ki points a rows-pass tile), and the row lengths the rows pass is instantiated for.
// rows = 1024 needs no rows pass of this kind (nothing to combine across rows): its cols pass works on its own rows,
// in place if asked, and the ordinary row transform follows.
static bool twopass_supported(int rows, int cols)
{
    return (rows == 1024 || rows == 2048 || rows == 4096) && (cols == 1024 || cols == 2048 || cols == 4096);
}

// the ordinary row transform of `planes` planes (fft_lines_kernel), in launches of less than 2 GiB
static int plain_rows_pass(const c2* in, c2* out, int rows, int cols, int planes, int dir, hipStream_t s);

// order: 0 = cols pass then rows pass (DIT), 1 = rows pass then cols pass (DIF).  The cols pass is out of place
// (src != dst) unless RX == 1, the rows pass works on its tile's own rows (src may equal dst).
template <int NC, int RX>
static int twopass_run_t(const c2* in, c2* mid, c2* out, int rows, int planes, int dir, int order, hipStream_t s)
{
    const TwoPassGeo g = twopass_geo(rows, NC, planes, dir);
    int rc;
    if constexpr (RX == 1) {
        if (order == 0) {
            rc = twopass_cols<1, NC, false>(in, mid, g, s);
            if (rc == MI_OK) rc = plain_rows_pass(mid, out, rows, NC, planes, dir, s);
        } else {
            rc = plain_rows_pass(in, mid, rows, NC, planes, dir, s);
            if (rc == MI_OK) rc = twopass_cols<1, NC, false>(mid, out, g, s);
        }
    } else if (order == 0) {
        rc = twopass_cols<RX, NC, false>(in, mid, g, s);
        if (rc == MI_OK) rc = twopass_rows<NC, RX, false>(mid, out, g, s);
    } else {
        rc = twopass_rows<NC, RX, true>(in, mid, g, s);
        if (rc == MI_OK) rc = twopass_cols<RX, NC, true>(mid, out, g, s);
    }
    return rc;
}

static int twopass_run(const c2* in, c2* mid, c2* out, int rows, int cols, int planes, int dir, int order, hipStream_t s)
{
#define MI_TP(NC_, RX_) if (cols == NC_ && rows == RX_ * 1024) return twopass_run_t<NC_, RX_>(in, mid, out, rows, planes, dir, order, s);
    MI_TP(1024, 1) MI_TP(1024, 2) MI_TP(1024, 4) MI_TP(2048, 1) MI_TP(2048, 2) MI_TP(2048, 4) MI_TP(4096, 1) MI_TP(4096, 2) MI_TP(4096, 4)
#undef MI_TP
    return fail(MI_ERR_UNSUPPORTED, "two-pass plan: size %dx%d", rows, cols);
}

// imageFFTImageGenerate's device core (imageTools.cu:633-791) with every pointwise step riding on a pass of the
// two-pass plan: u8 -> complex on the loads of the forward rows pass, shift x gain x shift on the stores of the
// forward cols pass, /N + saturate + pack on the stores of the inverse rows pass -- four passes over HBM per
// image where the step-by-step form makes nine.  rows x cols = the geometry the transforms see (the
// reference's: rows = w, cols = h); a, b: scratch of 3 * batch planes each.  Square images of 2048 or 4096 pixels
// a side (1024 has no rows pass of this kind to carry the conversions).
bool fft_highpass_fused_supported(int rows, int cols)
{
    return tuning().fft_twopass && tuning().fft_fused_spectral && rows == cols && (rows == 2048 || rows == 4096);
}

template <int N>
static int spectral_fused_forward_t(const void* d_in, int batch, float cutoff, void* a, void* b, hipStream_t s)
{
    constexpr int RX = N / 1024;
    TwoPassGeo f = twopass_geo(N, N, 3 * batch, -1);
    f.image_in = (const u32*)d_in;
    int rc = twopass_rows<N, RX, true, IO_U8_IN>((c2*)a, (c2*)a, f, s);                 // images -> a (rows pass of the DIF order)
    if (rc) return rc;
    if (cutoff > 0.0f) {
        f.gain_h = N; f.cutoff = cutoff;
        return twopass_cols<RX, N, true, true>((c2*)a, (c2*)b, f, s);                 // a -> b: the filtered spectra
    }
    return twopass_cols<RX, N, true>((c2*)a, (c2*)b, f, s);
}

template <int N>
static int spectral_fused_inverse_t(void* d_out, int batch, void* a, void* b, hipStream_t s)
{
    constexpr int RX = N / 1024;
    TwoPassGeo i = twopass_geo(N, N, 3 * batch, +1);
    int rc = twopass_cols<RX, N, false>((c2*)b, (c2*)a, i, s);                         // b -> a (cols pass of the DIT order)
    if (rc) return rc;
    i.image_out = (u32*)d_out; i.div = (float)(N * N);
    return twopass_rows<N, RX, false, IO_U8_OUT>((c2*)a, (c2*)a, i, s);                // a -> images
}

// forward half: the images' colour planes -> their spectra in b (natural order), times the high-pass factor
// when cutoff > 0; a: scratch of the same size
int launch_spectral_fused_forward(const void* d_in, int w, int h, int batch, int rows, int cols, float cutoff,
                                  void* a, void* b, hipStream_t s)
{
    // (the filter indexes the plane as h rows of w points: the plane's own geometry only when w == cols)
    if (!fft_highpass_fused_supported(rows, cols) || w != cols || h != rows) return fail(MI_ERR_UNSUPPORTED, "fused spectral path: size %dx%d", rows, cols);
    const int rc = ensure_twiddles(s);
    if (rc) return rc;
    return rows == 4096 ? spectral_fused_forward_t<4096>(d_in, batch, cutoff, a, b, s) : spectral_fused_forward_t<2048>(d_in, batch, cutoff, a, b, s);
}

// inverse half: the spectra in b -> images (real part / (w h), saturated, alpha 255); a: scratch
int launch_spectral_fused_inverse(void* d_out, int w, int h, int batch, int rows, int cols, void* a, void* b, hipStream_t s)
{
    if (!fft_highpass_fused_supported(rows, cols) || w != cols || h != rows) return fail(MI_ERR_UNSUPPORTED, "fused spectral path: size %dx%d", rows, cols);
    return rows == 4096 ? spectral_fused_inverse_t<4096>(d_out, batch, a, b, s) : spectral_fused_inverse_t<2048>(d_out, batch, a, b, s);
}

int launch_fft_highpass_fused(const void* d_in, void* d_out, int w, int h, int batch, int rows, int cols, float cutoff,
                              void* a, void* b, hipStream_t s)
{
    const int rc = launch_spectral_fused_forward(d_in, w, h, batch, rows, cols, cutoff, a, b, s);
    return rc ? rc : launch_spectral_fused_inverse(d_out, w, h, batch, rows, cols, a, b, s);
}

static int plain_rows_pass(const c2* in, c2* out, int rows, int cols, int planes, int dir, hipStream_t s)
{
    const i64 plane = (i64)rows * cols;
    const int group = (int)std::max<i64>(1, std::min<i64>((((i64)1 << 31) - 1) / (plane * 8), ((i64)1 << 30) / rows));
    for (int b0 = 0; b0 < planes; b0 += group) {
        const int nb = std::min(group, planes - b0);
        FftGeo row;
        row.inverse = dir > 0; row.plane_stride = plane; row.tw_step = 0; row.tw_uniform = 0;
        row.inner_log2 = 0; row.groups_log2 = __builtin_ctz(rows); row.total_lines = rows * nb; row.line_stride = 0;
        row.in_group_stride = row.out_group_stride = cols;
        row.in_elem_stride = row.out_elem_stride = 1;
        const int rc = launch_pass<false>(cols, in + (i64)b0 * plane, out + (i64)b0 * plane, row, (u32)((i64)nb * plane * 8), s);
        if (rc) return rc;
    }
    return MI_OK;
}

int launch_fft2d(const void* in, void* out, int rows, int cols, int batch, int dir, hipStream_t s, void* tmp)
{
    int rc = ensure_twiddles(s);
    if (rc) return rc;
    if (tuning().fft_twopass && twopass_supported(rows, cols) && (in != out || tmp != nullptr || rows == 1024)) {
        const i64 plane = (i64)rows * cols;
        const int order = tuning().fft_twopass == 2 ? 1 : 0;
        if (in != out) {
            // Both passes over up to MI_FFT_TWOPASS_GROUP planes at a time: a launch of fewer than a few
            // thousand tiles pays for its ramp-up and tail (2 planes per launch: 4.0 ms per round trip of
            // 16 planes; 8 or 16: 3.5 ms), which is worth more than finding the first pass's output
            // in the Infinity Cache.
            // (the knob counts 4096 x 4096 planes: the same bytes per launch for smaller ones)
            const int group = std::max(1, tuning().fft_twopass_group * (int)(((i64)4096 * 4096) / plane));
            for (int b0 = 0; b0 < batch; b0 += group) {
                const int nb = std::min(group, batch - b0);
                const c2* pin = (const c2*)in + (i64)b0 * plane;
                c2* pout = (c2*)out + (i64)b0 * plane;
                // DIT: cols pass in -> out, rows pass in place;  DIF: rows pass in -> out, cols pass needs a third buffer
                rc = order == 0 || !tmp ? twopass_run(pin, pout, pout, rows, cols, nb, dir, 0, s) : MI_OK;
                if (order == 1 && tmp) {
                    for (int b = 0; b < nb && rc == MI_OK; b++)
                        rc = twopass_run(pin + b * plane, (c2*)tmp, pout + b * plane, rows, cols, 1, dir, 1, s);
                }
                if (rc) return rc;
            }
        } else if (rows == 1024) {                 // the cols pass of a 1024-row plane works on its own rows: in place
            const int group = std::max(1, tuning().fft_twopass_group * (int)(((i64)4096 * 4096) / plane));
            for (int b0 = 0; b0 < batch; b0 += group) {
                c2* p = (c2*)out + (i64)b0 * plane;
                rc = twopass_run(p, p, p, rows, cols, std::min(group, batch - b0), dir, 0, s);
                if (rc) return rc;
            }
        } else {
            for (int b = 0; b < batch; b++) {      // one plane of scratch: a plane at a time
                c2* p = (c2*)out + (i64)b * plane;
                rc = twopass_run(p, (c2*)tmp, p, rows, cols, 1, dir, 0, s);
                if (rc) return rc;
            }
        }
        return MI_OK;
    }
    // Planes are transformed in groups that fit the 256 MiB Infinity Cache (all passes on the
    // same group back to back), so a pass finds the previous pass's output on die instead of
    // re-reading it from HBM.  MI_FFT_GROUP_MIB=0 makes every pass sweep the whole batch.
    const i64 plane = (i64)rows * cols;
    const i64 budget = (i64)tuning().fft_group_mib << 20;
    int group = budget <= 0 ? batch : (int)std::max<i64>(1, std::min<i64>(batch, budget / (plane * 8)));
    group = (int)std::max<i64>(1, std::min<i64>(group, ((i64)1 << 30) / ((i64)std::max(rows, 64 * cols))));
    group = (int)std::max<i64>(1, std::min<i64>(group, (((i64)1 << 31) - 1) / (plane * 8)));   // buffer offsets stay below 2^31
    // A column of >= 1024 points fits LDS only a few columns at a time (32-byte runs in global
    // memory).  Out of place, such a column transform is split R = R1*R2 into two passes of short
    // transforms over 64 adjacent columns each (512-byte runs):
    //   A: for every n2, FFT_R1 over n1 of x[R2*n1+n2], times W_R^(n2*k1), stored at row R1*n2+k1
    //   B: for every k1, FFT_R2 over n2 of rows R1*n2+k1, stored in place at rows k1+R1*k2
    // and the row transform runs last, in place (row and column transforms commute).
    // In place, the split needs one plane of scratch (`tmp`): A writes there, B works there, the
    // row pass brings the plane back; without it the single long-column pass is used (half speed).
    const bool via_tmp = in == out && tmp != nullptr;
    const bool split = (in != out || via_tmp) && rows >= 1024 && tuning().fft_split;
    if (split && via_tmp) group = 1;
    for (int b0 = 0; b0 < batch; b0 += group) {
        const int nb = std::min(group, batch - b0);
        const c2* pin = (const c2*)in + (i64)b0 * plane;
        c2* pout = (c2*)out + (i64)b0 * plane;
        const u32 bytes = (u32)((i64)nb * plane * 8);
        FftGeo g;
        g.inverse = dir > 0;
        g.plane_stride = plane;
        g.tw_step = 0;
        g.tw_uniform = 0;
        FftGeo row = g;
        row.inner_log2 = 0; row.groups_log2 = __builtin_ctz(rows); row.total_lines = rows * nb; row.line_stride = 0;
        row.in_group_stride = row.out_group_stride = cols;
        row.in_elem_stride = row.out_elem_stride = 1;
        if (split) {
            const int R2 = 64, R1 = rows / R2;
            FftGeo a = g;
            a.inner_log2 = __builtin_ctz(cols); a.groups_log2 = __builtin_ctz(R2); a.total_lines = cols * R2 * nb; a.line_stride = 1;
            a.in_group_stride = cols; a.in_elem_stride = R2 * cols;           // rows R2*n1 + n2
            a.out_group_stride = (i64)R1 * cols; a.out_elem_stride = cols;          // rows R1*n2 + k1
            a.tw_step = TW_N / rows;
            a.tw_uniform = cols >= 64;     // launch_pass<true> puts at most 64 adjacent columns in a block
            c2* mid = via_tmp ? (c2*)tmp : pout;
            rc = launch_pass<true>(R1, pin, mid, a, bytes, s);
            if (rc) return rc;
            FftGeo b = g;
            b.inner_log2 = __builtin_ctz(cols); b.groups_log2 = __builtin_ctz(R1); b.total_lines = cols * R1 * nb; b.line_stride = 1;
            b.in_group_stride = b.out_group_stride = cols;                          // rows k1 + R1*(.)
            b.in_elem_stride = b.out_elem_stride = R1 * cols;
            rc = launch_pass<true>(R2, mid, mid, b, bytes, s);
            if (rc) return rc;
            rc = launch_pass<false>(cols, mid, pout, row, bytes, s);
            if (rc) return rc;
        } else {
            rc = launch_pass<false>(cols, pin, pout, row, bytes, s);
            if (rc) return rc;
            FftGeo col = g;
            col.inner_log2 = __builtin_ctz(cols); col.groups_log2 = 0; col.total_lines = cols * nb; col.line_stride = 1;
            col.in_group_stride = col.out_group_stride = 0;
            col.in_elem_stride = col.out_elem_stride = cols;
            rc = launch_pass<true>(rows, pout, pout, col, bytes, s);
            if (rc) return rc;
        }
    }
    return MI_OK;
}


// ================================================================ lengths that are not 2^k
// The reference hands cuFFT whatever size the image has (its demo input is 675x900,
// main.cpp:34-45), so the library does too: a line of n points, n not a power of two, is
// transformed with Bluestein's chirp-z identity
//     X[k] = w[k] * sum_j (x[j] w[j]) * conj(w)[k-j],      w[j] = exp(-i pi j^2 / n),
// i.e. a cyclic convolution of length m = 2^p >= 2n-1 done with the power-of-two line
// transform above: multiply by the chirp and zero-pad, FFT_m, multiply by FFT_m(conj chirp),
// inverse FFT_m, multiply by the chirp / m.  j^2 is reduced mod 2n in integers and the chirp is
// evaluated in double, so the only fp32 rounding is that of the two m-point transforms.
// The column transform of such a plane goes through a transpose in the caller's workspace.

static bool is_pow2(int n) { return n >= 2 && (n & (n - 1)) == 0; }
static int blue_m(int n)            // 0: n is a power of two and needs no chirp
{
    if (is_pow2(n)) return 0;
    int m = 2;
    while (m < 2 * n - 1) m <<= 1;
    return m;
}

// chirp[j] = exp(-i pi j^2 / n), j < n;  b[q] = conj(chirp[min(q, m-q)]) or 0, q < m
__global__ void blue_tables_kernel(c2* __restrict__ chirp, c2* __restrict__ b, int n, int m)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const int d = q < m - q ? q : m - q;
    c2 w{0.f, 0.f};
    if (d < n) {
        const unsigned long long r = ((unsigned long long)d * (unsigned long long)d) % (2ull * (unsigned long long)n);
        double sn, cs;
        sincospi((double)r / (double)n, &sn, &cs);
        w = c2{(float)cs, (float)sn};                  // conj(chirp[d])
    }
    b[q] = w;
    if (q < n) chirp[q] = c2{w.x, -w.y};
}

// pad[l][q] = q < n ? x[l][q] (conjugated for the inverse) * chirp[q] : 0
__global__ __launch_bounds__(256) void blue_pre_kernel(const c2* __restrict__ src, i64 src_stride,
                                                       c2* __restrict__ pad, const c2* __restrict__ chirp,
                                                       int n, int m_log2, i64 total, int inverse)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i & ((1 << m_log2) - 1));
    const i64 l = i >> m_log2;
    c2 v{0.f, 0.f};
    if (q < n) {
        v = src[l * src_stride + q];
        if (inverse) v.y = -v.y;
        v = cmul(v, chirp[q]);
    }
    pad[i] = v;
}

__global__ __launch_bounds__(256) void blue_mul_kernel(c2* __restrict__ pad, const c2* __restrict__ bhat,
                                                       int m_log2, i64 total)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    pad[i] = cmul(pad[i], bhat[(int)(i & ((1 << m_log2) - 1))]);
}

// dst[l][k] = pad[l][k] * chirp[k] / m  (conjugated for the inverse), k < n
__global__ __launch_bounds__(256) void blue_post_kernel(const c2* __restrict__ pad, c2* __restrict__ dst,
                                                        i64 dst_stride, const c2* __restrict__ chirp,
                                                        int n, int m_log2, i64 total, float inv_m, int inverse)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % n);
    const i64 l = i / n;
    c2 v = cmul(pad[(l << m_log2) + k], chirp[k]);
    v.x *= inv_m; v.y *= inv_m;
    if (inverse) v.y = -v.y;
    dst[l * dst_stride + k] = v;
}

// The whole chirp-z transform of a line in one kernel: the line never leaves LDS between the chirp
// multiply, the m-point transform, the multiplication by the transformed chirp, the inverse
// m-point transform and the final chirp multiply, so a pass reads n points and writes n points per
// line where the five separate kernels above move about 8 m.  Same operations in the same order
// as those kernels (which remain for m < 16): the results are bit-identical.
// LPB lines per block iteration, M/16 lanes per line, 16 points per lane (point i = t + (M/16) k).
template <int M, int LPB>
__global__ __launch_bounds__((M / 16) * LPB) void blue_lines_kernel(const c2* __restrict__ src, i64 src_stride,
                                                                    c2* __restrict__ dst, i64 dst_stride, int nlines,
                                                                    int n, const c2* __restrict__ chirp,
                                                                    const c2* __restrict__ bhat, int inverse)
{
    constexpr int T = M / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    c2* lds = reinterpret_cast<c2*>(lds_raw);
    const int tid = threadIdx.x, lin = tid / T, t = tid % T;
    c2* line = lds + lin * (M + M / 16);
    c2* tw = lds + LPB * (M + M / 16);
    fill_twiddles<M>(tw, tid, T * LPB);
    __syncthreads();
    const float inv_m = 1.0f / (float)M;
    // the trip count is block-uniform: the barriers inside fft_line are safe
    for (i64 L0 = (i64)blockIdx.x * LPB; L0 < nlines; L0 += (i64)gridDim.x * LPB) {
        const i64 l = L0 + lin;
        const bool active = l < nlines;
        const c2* x = src + l * src_stride;
        c2 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = t + T * k;
            c2 a{0.f, 0.f};
            if (active && i < n) {
                a = x[i];
                if (inverse) a.y = -a.y;
                a = cmul(a, chirp[i]);
            }
            v[k] = a;
        }
        fft_line<M>(v, line, tw, t);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            v[k] = cmul(v[k], bhat[t + T * k]);
            v[k].y = -v[k].y;                      // inverse transform = conj, forward transform, conj
        }
        fft_line<M>(v, line, tw, t);
        c2* y = dst + l * dst_stride;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = t + T * k;
            if (active && i < n) {
                c2 r = v[k];
                r.y = -r.y;
                r = cmul(r, chirp[i]);
                r.x *= inv_m; r.y *= inv_m;
                if (inverse) r.y = -r.y;
                y[i] = r;
            }
        }
    }
}

// dst (cols x rows) = transpose of src (rows x cols); 32x32 tiles through LDS
// (blockIdx.z = plane of a batch of back-to-back planes)
__global__ __launch_bounds__(256) void transpose_c2_kernel(const c2* __restrict__ src, c2* __restrict__ dst,
                                                           int rows, int cols)
{
    __shared__ c2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    src += (i64)blockIdx.z * rows * cols;
    dst += (i64)blockIdx.z * rows * cols;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int x = x0 + tx, y = y0 + ty + j;
        if (x < cols && y < rows) tile[ty + j][tx] = src[(i64)y * cols + x];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int y = y0 + tx, x = x0 + ty + j;
        if (x < cols && y < rows) dst[(i64)x * rows + y] = tile[tx][ty + j];
    }
}

static int grid_for(i64 total, int* grid)
{
    const i64 blocks = (total + 255) / 256;
    if (blocks > 0x7fffffff) return fail(MI_ERR_UNSUPPORTED, "fft launch too large");
    *grid = (int)blocks;
    return MI_OK;
}

// `nlines` lines of n = 2^k contiguous points each, line l at src + l*src_stride -> dst + l*dst_stride
static int pow2_lines(const c2* src, i64 src_stride, c2* dst, i64 dst_stride, int nlines, int n, int dir, hipStream_t s)
{
    FftGeo g;
    g.inverse = dir > 0;
    g.plane_stride = 0;
    g.tw_step = 0;
    g.tw_uniform = 0;
    g.inner_log2 = 0; g.groups_log2 = 30; g.line_stride = 0;      // line index == group index
    g.in_group_stride = src_stride; g.out_group_stride = dst_stride;
    g.in_elem_stride = g.out_elem_stride = 1;
    const i64 max_stride = std::max(src_stride, dst_stride);
    const i64 per_launch = std::max<i64>(1, std::min<i64>((((i64)1 << 31) - 1 - (i64)n * 8) / (max_stride * 8) , (i64)1 << 29));
    for (i64 l0 = 0; l0 < nlines; l0 += per_launch) {
        const int nl = (int)std::min<i64>(per_launch, nlines - l0);
        g.total_lines = nl;
        const u32 bytes = (u32)(((i64)(nl - 1) * max_stride + n) * 8);
        int rc = launch_pass<false>(n, src + l0 * src_stride, dst + l0 * dst_stride, g, bytes, s);
        if (rc) return rc;
    }
    return MI_OK;
}

struct BlueTables { const c2* chirp; const c2* bhat; int m; };

template <int M, int LPB>
static int launch_blue_lines(const c2* src, i64 src_stride, c2* dst, i64 dst_stride, int nlines, int n, int inverse,
                             const BlueTables& t, hipStream_t s)
{
    constexpr int threads = (M / 16) * LPB;
    constexpr size_t lds = ((size_t)LPB * (M + M / 16) + Plan<M>::tw_total) * sizeof(c2);
    static_assert(lds <= 160 * 1024, "a block's lines and twiddle tables must fit the CU's LDS");
    static PerDeviceAttr attr;
    const hipError_t attr_err = attr.get(nullptr, [](int*) {
        return lds > 64 * 1024 ? hipFuncSetAttribute((const void*)blue_lines_kernel<M, LPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                               : hipSuccess;
    });
    if (attr_err != hipSuccess) return hip_status(attr_err, "hipFuncSetAttribute(max dynamic LDS)");
    const i64 blocks = ((i64)nlines + LPB - 1) / LPB;
    // persistent blocks, two per CU (measured: 5-8 % faster than one block per group; holding the next
    // group's points in a second register set, as the power-of-two kernel does, was 30 % slower here)
    const int grid = (int)std::min<i64>(blocks, 256 * (i64)std::max(1, tuning().fft_chirpz_blocks_per_cu));
    blue_lines_kernel<M, LPB><<<grid, threads, lds, s>>>(src, src_stride, dst, dst_stride, nlines, n, t.chirp, t.bhat, inverse);
    return hip_status(hipGetLastError(), "chirp-z kernel launch");
}

// Any n: lines as above; `pad` holds nlines * m points when n is not a power of two.
static int any_lines(const c2* src, i64 src_stride, c2* dst, i64 dst_stride, int nlines, int n, int dir,
                     const BlueTables& t, c2* pad, hipStream_t s)
{
    if (t.m == 0) return pow2_lines(src, src_stride, dst, dst_stride, nlines, n, dir, s);
    const int m_log2 = __builtin_ctz(t.m);
    const int inverse = dir > 0;
    if (tuning().fft_fused_chirpz) {
        switch (t.m) {     // lines per block as in launch_pass's row pass
        case 16: return launch_blue_lines<16, 64>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        case 32: return launch_blue_lines<32, 64>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        case 64: return launch_blue_lines<64, 64>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        case 128: return launch_blue_lines<128, 32>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        case 256: return launch_blue_lines<256, 16>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        case 512: return launch_blue_lines<512, 8>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        case 1024: return launch_blue_lines<1024, 4>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        case 2048: return launch_blue_lines<2048, 2>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        case 4096: return launch_blue_lines<4096, 1>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        case 8192: return launch_blue_lines<8192, 1>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        case 16384: return launch_blue_lines<16384, 1>(src, src_stride, dst, dst_stride, nlines, n, inverse, t, s);
        default: break;    // m = 8 (n = 3): the separate kernels below
        }
    }
    int grid = 0;
    int rc = grid_for((i64)nlines * t.m, &grid);
    if (rc) return rc;
    blue_pre_kernel<<<grid, 256, 0, s>>>(src, src_stride, pad, t.chirp, n, m_log2, (i64)nlines * t.m, inverse);
    rc = pow2_lines(pad, t.m, pad, t.m, nlines, t.m, -1, s);
    if (rc) return rc;
    blue_mul_kernel<<<grid, 256, 0, s>>>(pad, t.bhat, m_log2, (i64)nlines * t.m);
    rc = pow2_lines(pad, t.m, pad, t.m, nlines, t.m, +1, s);
    if (rc) return rc;
    rc = grid_for((i64)nlines * n, &grid);
    if (rc) return rc;
    blue_post_kernel<<<grid, 256, 0, s>>>(pad, dst, dst_stride, t.chirp, n, m_log2, (i64)nlines * n, 1.0f / (float)t.m, inverse);
    return hip_status(hipGetLastError(), "chirp-z kernel launch");
}

// Workspace layout, in complex points: [transposed plane | padded lines | chirp_c, bhat_c, chirp_r, bhat_r]
struct BlueLayout { i64 plane_t, pad, tables, total; int mc, mr; };

static BlueLayout blue_layout(int rows, int cols)
{
    BlueLayout L;
    L.mc = blue_m(cols);
    L.mr = blue_m(rows);
    L.plane_t = (i64)rows * cols;
    L.pad = std::max((i64)rows * L.mc, (i64)cols * L.mr);
    L.tables = (i64)(L.mc ? cols + L.mc : 0) + (L.mr ? rows + L.mr : 0);
    L.total = L.plane_t + L.pad + L.tables;
    return L;
}

int fft2d_size_supported(int rows, int cols)
{
    // a plane stays below 2 GiB: the kernels address it through 32-bit buffer offsets
    auto ok = [](int n) { return n >= 2 && (n <= MI_FFT_MAX_ANY || (n <= MI_FFT_MAX_POW2 && is_pow2(n))); };
    return ok(rows) && ok(cols) && (i64)rows * cols < ((i64)1 << 28);
}

size_t fft2d_workspace_bytes(int rows, int cols)
{
    // powers of two need none; tall planes transformed IN PLACE run at full speed with one plane of it
    if (is_pow2(rows) && is_pow2(cols)) return rows >= 1024 ? (size_t)rows * cols * sizeof(c2) : 0;
    return (size_t)blue_layout(rows, cols).total * sizeof(c2);
}

int launch_fft2d_any(const void* in, void* out, int rows, int cols, int batch, int dir, void* ws, hipStream_t s,
                     bool tables_ready)
{
    if (is_pow2(rows) && is_pow2(cols)) return launch_fft2d(in, out, rows, cols, batch, dir, s, ws);
    int rc = ensure_twiddles(s);
    if (rc) return rc;
    const BlueLayout L = blue_layout(rows, cols);
    c2* plane_t = (c2*)ws;
    c2* pad = plane_t + L.plane_t;
    c2* tab = pad + L.pad;
    BlueTables tc{nullptr, nullptr, L.mc}, tr{nullptr, nullptr, L.mr};
    auto make_tables = [&](BlueTables& t, int n) -> int {
        if (!t.m) return MI_OK;
        c2* chirp = tab; c2* bhat = tab + n;
        tab += n + t.m;
        t.chirp = chirp; t.bhat = bhat;
        if (tables_ready) return MI_OK;
        blue_tables_kernel<<<(t.m + 255) / 256, 256, 0, s>>>(chirp, bhat, n, t.m);
        return pow2_lines(bhat, t.m, bhat, t.m, 1, t.m, -1, s);
    };
    rc = make_tables(tc, cols);
    if (rc == MI_OK) rc = make_tables(tr, rows);
    if (rc) return rc;
    const i64 plane = (i64)rows * cols;
    // Planes go through in groups: the transposed copies of a group live in the workspace's plane and
    // (when the fused kernels run, which need no padded lines) its padded-line area -- at least three
    // planes together.  With the separate kernels (MI_FFT_FUSED_CHIRPZ=0, or m = 8) one plane at a time.
    const bool fused = tuning().fft_fused_chirpz && (L.mc == 0 || L.mc >= 16) && (L.mr == 0 || L.mr >= 16);
    const int group = fused ? (int)std::min<i64>(std::min<i64>(batch, 65535), std::max<i64>(1, (L.plane_t + L.pad) / plane)) : 1;
    for (int b = 0; b < batch; b += group) {
        const int nb = std::min(group, batch - b);
        const c2* pin = (const c2*)in + b * plane;
        c2* pout = (c2*)out + b * plane;
        const dim3 tgrid_fwd((cols + 31) / 32, (rows + 31) / 32, nb), tgrid_back((rows + 31) / 32, (cols + 31) / 32, nb);
        rc = any_lines(pin, cols, pout, cols, rows * nb, cols, dir, tc, pad, s);            // rows (planes are back to back)
        if (rc) return rc;
        transpose_c2_kernel<<<tgrid_fwd, 256, 0, s>>>(pout, plane_t, rows, cols);
        rc = any_lines(plane_t, rows, plane_t, rows, cols * nb, rows, dir, tr, pad, s);     // columns, as rows of the transposes
        if (rc) return rc;
        transpose_c2_kernel<<<tgrid_back, 256, 0, s>>>(plane_t, pout, cols, rows);
        rc = hip_status(hipGetLastError(), "transpose kernel launch");
        if (rc) return rc;
    }
    return MI_OK;
}

}  // namespace mi
