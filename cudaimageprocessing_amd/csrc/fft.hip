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
// accesses are LPB*8-byte runs.  Twiddles come from a table in device memory computed once per
// device in double precision.  The inverse is conj(FFT(conj(x))).
#include <mutex>

#include "common.h"
#include "internal.h"

namespace mi {

struct c2 { float x, y; };

MI_DEV c2 cadd(c2 a, c2 b) { return c2{a.x + b.x, a.y + b.y}; }
MI_DEV c2 csub(c2 a, c2 b) { return c2{a.x - b.x, a.y - b.y}; }
MI_DEV c2 cmul(c2 a, c2 b) { return c2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
MI_DEV c2 mul_neg_i(c2 a) { return c2{a.y, -a.x}; }   // a * (-i)

constexpr int TW_N = 8192;
__device__ c2 g_twiddle[TW_N];   // exp(-2*pi*i*n/8192)

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
struct FftGeo {
    int lines_per_plane;   // rows (row pass) or cols (column pass)
    i64 total_lines;       // lines_per_plane * batch
    i64 plane_stride;      // rows * cols
    i64 line_stride;       // cols (row pass) or 1 (column pass)
    i64 elem_stride;       // 1 (row pass) or cols (column pass)
    int inverse;
};

MI_DEV int lds_pad(int i) { return i + (i >> 4); }   // one spare slot per 16: breaks the 128-byte stride

// One Stockham stage of radix R over this lane's 16 points (N/16 lanes per line).
// NS = product of the radices of the earlier stages.
template <int N, int R, int NS, bool LAST>
MI_DEV void fft_stage(c2 (&v)[16], c2* __restrict__ line, int t)
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
            constexpr int step = TW_N / (NS * R);
#pragma unroll
            for (int q = 1; q < R; q++) x[q] = cmul(x[q], g_twiddle[q * ktw * step]);
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

template <int N>
MI_DEV void fft_line(c2 (&v)[16], c2* __restrict__ line, int t)
{
    constexpr int L = __builtin_ctz(N);
    constexpr int A = L / 4;                  // radix-16 stages
    constexpr int REM = 1 << (L % 4);         // trailing radix (1 = none)
    if constexpr (A == 1 && REM == 1) {
        fft_stage<N, 16, 1, true>(v, line, t);
    } else if constexpr (A == 1) {
        fft_stage<N, 16, 1, false>(v, line, t);
        fft_stage<N, REM, 16, true>(v, line, t);
    } else if constexpr (A == 2 && REM == 1) {
        fft_stage<N, 16, 1, false>(v, line, t);
        fft_stage<N, 16, 16, true>(v, line, t);
    } else if constexpr (A == 2) {
        fft_stage<N, 16, 1, false>(v, line, t);
        fft_stage<N, 16, 16, false>(v, line, t);
        fft_stage<N, REM, 256, true>(v, line, t);
    } else if constexpr (A == 3 && REM == 1) {
        fft_stage<N, 16, 1, false>(v, line, t);
        fft_stage<N, 16, 16, false>(v, line, t);
        fft_stage<N, 16, 256, true>(v, line, t);
    } else {
        static_assert(A == 3 && REM == 2, "line length out of range");
        fft_stage<N, 16, 1, false>(v, line, t);
        fft_stage<N, 16, 16, false>(v, line, t);
        fft_stage<N, 16, 256, false>(v, line, t);
        fft_stage<N, 2, 4096, true>(v, line, t);
    }
}

// COL = false: lanes of a line are adjacent (lane-contiguous points of a row).
// COL = true : LPB adjacent lines (columns) are adjacent lanes (runs of LPB*8 bytes per point).
template <int N, int LPB, bool COL>
__global__ __launch_bounds__((N / 16) * LPB) void fft_lines_kernel(const c2* __restrict__ in,
                                                                   c2* __restrict__ out,
                                                                   const FftGeo g)
{
    constexpr int T = N / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    c2* lds = reinterpret_cast<c2*>(lds_raw);
    const int tid = threadIdx.x;
    const int lin = COL ? tid % LPB : tid / T;
    const int t = COL ? tid / LPB : tid % T;
    const i64 L = (i64)blockIdx.x * LPB + lin;
    const bool active = L < g.total_lines;
    const i64 plane = active ? L / g.lines_per_plane : 0;
    const i64 l = active ? L % g.lines_per_plane : 0;
    const i64 base = plane * g.plane_stride + l * g.line_stride;
    c2* line = lds + lin * (N + N / 16);

    c2 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        v[k] = active ? in[base + (i64)(t + T * k) * g.elem_stride] : c2{0.f, 0.f};
        if (g.inverse) v[k].y = -v[k].y;
    }
    fft_line<N>(v, line, t);
    if (active) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (g.inverse) v[k].y = -v[k].y;
            out[base + (i64)(t + T * k) * g.elem_stride] = v[k];
        }
    }
}

// Lines of 2, 4 or 8 points: one lane per line, all in registers.
template <int N>
__global__ __launch_bounds__(256) void fft_small_kernel(const c2* __restrict__ in,
                                                        c2* __restrict__ out, const FftGeo g)
{
    const i64 L = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= g.total_lines) return;
    const i64 base = (L / g.lines_per_plane) * g.plane_stride + (L % g.lines_per_plane) * g.line_stride;
    c2 x[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        x[k] = in[base + k * g.elem_stride];
        if (g.inverse) x[k].y = -x[k].y;
    }
    dft<N>(x);
#pragma unroll
    for (int k = 0; k < N; k++) {
        if (g.inverse) x[k].y = -x[k].y;
        out[base + k * g.elem_stride] = x[k];
    }
}

// ---------------------------------------------------------------- host side
static int env_group_mib()
{
    const char* v = getenv("MI_FFT_GROUP_MIB");
    return (v && *v) ? atoi(v) : 160;
}

static int ensure_twiddles(hipStream_t s)
{
    static std::mutex mu;
    static bool ready[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_status(e, "hipGetDevice");
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64 || !ready[dev]) {
        // stream-ordered: the transform kernels that follow on `s` see the table; the first
        // call on a device must not race with FFT calls on other streams of that device
        twiddle_init_kernel<<<TW_N / 256, 256, 0, s>>>();
        e = hipGetLastError();
        if (e != hipSuccess) return hip_status(e, "twiddle table init");
        if (dev >= 0 && dev < 64) ready[dev] = true;
    }
    return MI_OK;
}

template <int N, int LPB, bool COL>
static int launch_lines(const c2* in, c2* out, const FftGeo& g, hipStream_t s)
{
    constexpr int threads = (N / 16) * LPB;
    constexpr size_t lds = (size_t)LPB * (N + N / 16) * sizeof(c2);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    if (lds > 64 * 1024)
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute((const void*)fft_lines_kernel<N, LPB, COL>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
    if (attr_err != hipSuccess) return hip_status(attr_err, "hipFuncSetAttribute(max dynamic LDS)");
    const i64 blocks = (g.total_lines + LPB - 1) / LPB;
    if (blocks > 0x7fffffff) return fail(MI_ERR_UNSUPPORTED, "fft launch too large");
    fft_lines_kernel<N, LPB, COL><<<(int)blocks, threads, lds, s>>>(in, out, g);
    return hip_status(hipGetLastError(), "fft kernel launch");
}

template <int N>
static int launch_small(const c2* in, c2* out, const FftGeo& g, hipStream_t s)
{
    const i64 blocks = (g.total_lines + 255) / 256;
    if (blocks > 0x7fffffff) return fail(MI_ERR_UNSUPPORTED, "fft launch too large");
    fft_small_kernel<N><<<(int)blocks, 256, 0, s>>>(in, out, g);
    return hip_status(hipGetLastError(), "fft kernel launch");
}

// Row pass: as many whole lines per 256-lane block as fit.  Column pass: as many adjacent
// columns per block as 1024 lanes and 160 KiB of LDS allow, for the widest global runs.
template <bool COL>
static int launch_pass(int n, const c2* in, c2* out, const FftGeo& g, hipStream_t s)
{
    switch (n) {
    case 2: return launch_small<2>(in, out, g, s);
    case 4: return launch_small<4>(in, out, g, s);
    case 8: return launch_small<8>(in, out, g, s);
    case 16: return launch_lines<16, COL ? 64 : 64, COL>(in, out, g, s);
    case 32: return launch_lines<32, COL ? 64 : 64, COL>(in, out, g, s);
    case 64: return launch_lines<64, COL ? 64 : 64, COL>(in, out, g, s);
    case 128: return launch_lines<128, COL ? 32 : 32, COL>(in, out, g, s);
    case 256: return launch_lines<256, COL ? 32 : 16, COL>(in, out, g, s);
    case 512: return launch_lines<512, COL ? 32 : 8, COL>(in, out, g, s);
    case 1024: return launch_lines<1024, COL ? 16 : 4, COL>(in, out, g, s);
    case 2048: return launch_lines<2048, COL ? 8 : 2, COL>(in, out, g, s);
    case 4096: return launch_lines<4096, COL ? 4 : 1, COL>(in, out, g, s);
    case 8192: return launch_lines<8192, COL ? 2 : 1, COL>(in, out, g, s);
    default: return fail(MI_ERR_UNSUPPORTED, "fft line length %d", n);
    }
}

int launch_fft2d(const void* in, void* out, int rows, int cols, int batch, int dir, hipStream_t s)
{
    int rc = ensure_twiddles(s);
    if (rc) return rc;
    // Planes are transformed in groups that fit the 256 MiB Infinity Cache (row pass then column
    // pass on the same group), so the column pass finds the row pass's output on die instead of
    // re-reading it from HBM.  MI_FFT_GROUP_MIB=0 restores one row pass and one column pass over
    // the whole batch.
    const i64 plane_bytes = (i64)rows * cols * 8;
    const i64 budget = (i64)env_group_mib() << 20;
    int group = budget <= 0 ? batch : (int)std::max<i64>(1, std::min<i64>(batch, budget / plane_bytes));
    for (int b0 = 0; b0 < batch; b0 += group) {
        const int nb = std::min(group, batch - b0);
        const c2* pin = (const c2*)in + (i64)b0 * rows * cols;
        c2* pout = (c2*)out + (i64)b0 * rows * cols;
        FftGeo g;
        g.inverse = dir > 0;
        g.plane_stride = (i64)rows * cols;
        // rows: lines of `cols` contiguous points
        g.lines_per_plane = rows;
        g.total_lines = (i64)rows * nb;
        g.line_stride = cols;
        g.elem_stride = 1;
        rc = launch_pass<false>(cols, pin, pout, g, s);
        if (rc) return rc;
        // columns, in place on the row-pass output
        g.lines_per_plane = cols;
        g.total_lines = (i64)cols * nb;
        g.line_stride = 1;
        g.elem_stride = cols;
        rc = launch_pass<true>(rows, pout, pout, g, s);
        if (rc) return rc;
    }
    return MI_OK;
}

}  // namespace mi
