// fft_twopass.h -- a tall power-of-two plane transformed in TWO passes over HBM per direction
// (32 B per point: the algorithmic minimum of a 2-D transform that does not fit on chip), natural
// order in and out.  Included by fft.hip (same translation unit: the twiddle table, the small DFTs).
//
// The column transform of R = RX * R2 points is split by one Cooley-Tukey step, and the cheap half
// of it (RX = 2 or 4 points) rides along with the row transform:
//
//   cols pass  R2-point transforms down 512/(R2/32) adjacent columns (128-byte runs for R2 = 1024),
//              reading rows RX*n2 + n1 and writing rows R2*n1 + k2        (n1 < RX: "DIT" order), or
//              reading rows R2*k1 + n2 and writing rows RX*k2 + k1        ("DIF" order);
//   rows pass  the RX rows r + R2*n1 together: the RX-point butterfly across them (with the
//              step's twiddle W_R^(n1*r)) and the whole-row transform of each, in place.
//
// DIT = cols pass, then rows pass;  DIF = rows pass, then cols pass.  Both give the same natural-order
// result; which pass comes first decides which one can carry a conversion on its loads or stores
// (the rows pass touches whole contiguous rows).
//
// Why this shape.  A pass that is to run at memory speed needs its loads in flight while other
// lanes do butterflies.  A tile of 16 Ki points (128 KiB of fp32 pairs: 16 columns x 1024 rows, or
// 4 rows x 4096 columns) fills a CU's LDS, so a kernel that keeps the tile in LDS runs one block per
// CU and nothing overlaps (round 2 measured that: 2.5 TB/s).  Here the tile lives in REGISTERS --
// 512 lanes x 32 points -- and LDS is only the exchange medium between stages, half a tile at a
// time (two rows at a time in the rows pass, real and imaginary parts separately in the cols pass):
// 68 KiB + twiddles per block, two blocks per CU, one loading or storing while the other computes.
#pragma once

// Experiment switch (tools/make_variant.sh with MI_EXTRA_CXXFLAGS=-DMI_FFT2P_DBG=n; wrong results by
// construction, never in a shipped build): 1 = loads and stores only, 2 = butterflies and exchanges only.
#ifndef MI_FFT2P_DBG
#define MI_FFT2P_DBG 0
#endif

namespace mi {

MI_DEV c2 cmulf(c2 a, c2 b)      // fused: two roundings fewer and 4 instead of 6 operations
{
    return c2{__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x)};
}

// exp(-2 pi i m / 32), m < 22
MI_DEV c2 w32(int m)
{
    constexpr float C[22] = {1.0f, 0.98078528040323043058f, 0.92387953251128673848f, 0.83146961230254523567f,
                             0.70710678118654757274f, 0.55557023301960228867f, 0.38268343236508983729f, 0.19509032201612833135f,
                             0.0f, -0.19509032201612819257f, -0.38268343236508972627f, -0.55557023301960195560f,
                             -0.70710678118654746172f, -0.83146961230254534669f, -0.92387953251128673848f, -0.98078528040323043058f,
                             -1.0f, -0.98078528040323043058f, -0.92387953251128684951f, -0.83146961230254545772f,
                             -0.70710678118654768376f, -0.55557023301960217765f};
    constexpr float S[22] = {0.0f, -0.19509032201612824808f, -0.38268343236508978178f, -0.55557023301960217765f,
                             -0.70710678118654746172f, -0.83146961230254523567f, -0.92387953251128673848f, -0.98078528040323043058f,
                             -1.0f, -0.98078528040323043058f, -0.92387953251128673848f, -0.83146961230254545772f,
                             -0.70710678118654757274f, -0.55557023301960217765f, -0.38268343236508989280f, -0.19509032201612860891f,
                             0.0f, 0.19509032201612835911f, 0.38268343236508967076f, 0.55557023301960195560f,
                             0.70710678118654746172f, 0.83146961230254523567f};
    return c2{C[m], S[m]};
}

// n = 8*n1 + n2, k = k1 + 4*k2
template <>
MI_DEV void dft<32>(c2 (&x)[32])
{
#pragma unroll
    for (int n2 = 0; n2 < 8; n2++) dft4(x[n2], x[8 + n2], x[16 + n2], x[24 + n2]);   // over n1 -> k1, at x[8*k1 + n2]
#pragma unroll
    for (int k1 = 1; k1 < 4; k1++) {
#pragma unroll
        for (int n2 = 1; n2 < 8; n2++) {
            const int m = n2 * k1;
            if (m == 8) x[8 * k1 + n2] = mul_neg_i(x[8 * k1 + n2]);
            else x[8 * k1 + n2] = cmulf(x[8 * k1 + n2], w32(m));
        }
    }
    c2 o[32];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
        c2 y[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; n2++) y[n2] = x[8 * k1 + n2];
        dft<8>(y);                                                                      // over n2 -> k2
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) o[k1 + 4 * k2] = y[k2];
    }
#pragma unroll
    for (int k = 0; k < 32; k++) x[k] = o[k];
}

template <>
MI_DEV void dft<1>(c2 (&)[1]) {}

struct TwoPassGeo {
    int planes;          // planes in this launch, back to back
    i64 plane_stride;    // points
    int inverse;
    // the fused forms of imageFFTImageGenerate's pointwise steps (IO / GAIN template parameters):
    const u32* image_in; // IO_U8_IN: RGBA images, plane p = channel p % 3 of image p / 3 (uchar4ToCufftComplex, imageTools.cu:506-515)
    u32* image_out;      // IO_U8_OUT: the images written by the last rows pass (NormalizeAndZeroImaginary + ComplexRGBToUchar, :551-562, :517-527)
    float div;           // IO_U8_OUT: the normalisation's divisor, (float)(w * h)
    int gain_h;          // GAIN: rows of the image the filter sees (its width = the plane's row length: checked on the host)
    float cutoff;        // GAIN: highpassFilter's cutoff (:530-549)
    int xcd_remap;       // 1: an XCD's blocks take a contiguous run of tiles
    int nt_stores;       // 1: non-temporal stores
};

// Blocks are dealt to the 8 XCDs in turn; this gives each XCD a contiguous run of tiles instead (a
// bijection for every grid size).  As a pure copy the cols pass moves 5.2-5.4 TB/s this way, one tile
// per block, against 4.4-4.6 in launch order (tools/membench_fft.hip, profiles/r03_membench_cols2p.txt).
MI_DEV int xcd_contiguous_block()
{
    const u32 nblk = gridDim.x, b = blockIdx.x;
    const u32 q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    return (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx);
}

// w[q] = w1^q for q = 1 .. R-1 (depth <= 3 multiplications)
template <int R>
MI_DEV void twiddle_powers(c2 w1, c2 (&w)[R])
{
    w[0] = c2{1.f, 0.f};
    if constexpr (R > 1) w[1] = w1;
    if constexpr (R > 2) w[2] = cmulf(w1, w1);
    if constexpr (R > 3) w[3] = cmulf(w[2], w1);
    if constexpr (R > 4) {
        w[4] = cmulf(w[2], w[2]); w[5] = cmulf(w[4], w1); w[6] = cmulf(w[3], w[3]); w[7] = cmulf(w[4], w[3]);
    }
    static_assert(R <= 8, "power chain written for radices up to 8");
}

// ------------------------------------------------------------------ rows pass
// Row transform of NC points by NC/8 lanes, 8 points per lane: radix-8 Stockham stages (then one
// radix-2 or -4 stage when log2 NC is not a multiple of 3).  Stage s keeps one table of NS(s) base
// twiddles W_(8 NS)^k in LDS; the R-1 factors of a butterfly are its powers.
template <int N>
struct Plan8 {
    static constexpr int L = __builtin_ctz(N);
    static constexpr int A = L / 3;
    static constexpr int REM = 1 << (L % 3);
    static constexpr int n_stages = A + (REM > 1 ? 1 : 0);
    static constexpr int radix(int s) { return s < A ? 8 : REM; }
    static constexpr int ns(int s) { int p = 1; for (int i = 0; i < s; i++) p *= radix(i); return p; }
    static constexpr int tw_off(int s) { int o = 0; for (int i = 1; i < s; i++) o += ns(i); return o; }
    static constexpr int tw_total = tw_off(n_stages);
};

template <int N, int RX, int S = 0>
MI_DEV void rows_stages(c2 (&v)[RX][8], c2* __restrict__ lineA, c2* __restrict__ lineB, const c2* __restrict__ tw, int t)
{
    using P = Plan8<N>;
    if constexpr (S < P::n_stages) {
        constexpr int R = P::radix(S), NS = P::ns(S), T = N / 8, G = 8 / R;
        constexpr bool LAST = S == P::n_stages - 1;
        c2 w[G][R];
        if constexpr (NS > 1) {
#pragma unroll
            for (int m = 0; m < G; m++) twiddle_powers<R>(tw[P::tw_off(S) + ((t + T * m) % NS)], w[m]);
        }
        auto butterflies = [&](c2 (&row)[8], c2* __restrict__ line) {
#pragma unroll
            for (int m = 0; m < G; m++) {
                const int j = t + T * m;
                c2 x[R];
#pragma unroll
                for (int q = 0; q < R; q++) x[q] = row[m + q * G];
                if constexpr (NS > 1) {
#pragma unroll
                    for (int q = 1; q < R; q++) x[q] = cmulf(x[q], w[m][q]);
                }
                dft<R>(x);
                if constexpr (LAST) {
#pragma unroll
                    for (int q = 0; q < R; q++) row[m + q * G] = x[q];
                } else {
                    // lds_pad(d + q NS) = lds_pad(d) + q NS + (q NS >> 4): the low four bits of d are j % NS < NS
                    // (NS < 16) or 8 j's (NS = 1), so adding q NS never carries differently for different d
                    c2* dp = line + lds_pad((j / NS) * (NS * R) + (j % NS));
#pragma unroll
                    for (int q = 0; q < R; q++) dp[q * NS + ((q * NS) >> 4)] = x[q];
                }
            }
        };
        if constexpr (LAST) {
#pragma unroll
            for (int r = 0; r < RX; r++) butterflies(v[r], nullptr);
        } else {
#pragma unroll
            for (int r = 0; r < RX; r += 2) {
                butterflies(v[r], lineA);
                if constexpr (RX > 1) butterflies(v[r + 1], lineB);
                __syncthreads();
                static_assert(T % 16 == 0, "lds_pad(t + T k) = lds_pad(t) + T k + T k / 16");
                const c2* ra = lineA + lds_pad(t);
                const c2* rb = lineB + lds_pad(t);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    v[r][k] = ra[T * k + T * k / 16];
                    if constexpr (RX > 1) v[r + 1][k] = rb[T * k + T * k / 16];
                }
                __syncthreads();
            }
        }
        rows_stages<N, RX, S + 1>(v, lineA, lineB, tw, t);
    }
}

constexpr int IO_COMPLEX = 0, IO_U8_IN = 1, IO_U8_OUT = 2;

// ComplexRGBToUchar's cast with out-of-range values saturated (spectral.hip: sat_u8)
MI_DEV u32 sat_u8_f(float v)
{
    if (!(v > 0.0f)) return 0u;
    if (v >= 255.0f) return 255u;
    return (u32)v;
}

// fftShift -> highpassFilter -> fftShift as one factor for the element at image coordinates (x, y) (spectral.hip:
// highpass_unshifted_kernel; the same float expression, so the same bits): fftShift's swap, then the filter.
MI_DEV int shifted_coord(int v, int n) { const int half = n / 2; return v < 2 * half ? (v < half ? v + half : v - half) : v; }
MI_DEV float highpass_gain_xy(float cx, int y, int h, float cutoff)      // cx = (float)shifted x - w / 2.0f
{
    const float cy = (float)shifted_coord(y, h) - (float)h / 2.0f;
    const float distance = sqrtf(cx * cx + cy * cy);
    return 1.0f - expf(-(distance * distance) / (2.0f * cutoff * cutoff));
}

MI_DEV float flip_sign(float v, u32 mask) { return __uint_as_float(__float_as_uint(v) ^ mask); }

// One tile = the RX rows r + R2*n1 (n1 < RX) of one plane; one block of NC/8 lanes per tile.
// (All strides are compile-time: as run-time values the 32 scalar offsets of a tile's loads are
// loop invariants that the compiler keeps live -- and spills -- across the whole tile loop.)
// IO: IO_COMPLEX = planes in, planes out; IO_U8_IN = the tile's points come from RGBA images (one channel per
// plane, imaginary part 0); IO_U8_OUT = a block transforms the three colour planes of its image rows one after
// the other and writes them as RGBA pixels, real part / div, saturated (planes = 3 x images then).
template <int NC, int RX, int R2, bool DIF, bool NT, int IO = IO_COMPLEX>
__global__ __launch_bounds__(NC / 8, 4)
void fft2p_rows_kernel(const c2* __restrict__ in, c2* __restrict__ out, const TwoPassGeo g)
{
    constexpr int T = NC / 8;
    using P = Plan8<NC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    c2* lds = reinterpret_cast<c2*>(lds_raw);
    c2* lineA = lds;
    c2* lineB = lds + (NC + NC / 16);
    c2* tw = lds + 2 * (NC + NC / 16);
    const int t = threadIdx.x;
    // Base twiddles of stages 1..: W_(R NS)^k, k < NS -- entry e of the block's table, e = t and t + T.
    // Their loads are issued BEFORE the tile's (they come back first: vector-memory operations
    // complete in order) and go to LDS after the tile's loads are on their way.
    constexpr int NTW = (P::tw_total + T - 1) / T;         // table entries per lane (2 for 4096 points, 5 for 1024)
    c2 twr[NTW];
#pragma unroll
    for (int i = 0; i < NTW; i++) {
        const int e = t + T * i;
        int off = 0, ns = 1, idx = 0;
#pragma unroll
        for (int s = 1; s < P::n_stages; s++) {
            ns *= P::radix(s - 1);
            if (e >= off && e < off + ns) idx = (e - off) * (TW_N / (ns * P::radix(s)));
            off += ns;
        }
        twr[i] = g_twiddle[idx];
    }
    constexpr int r2_log2 = __builtin_ctz(R2);
    constexpr int CH = IO == IO_U8_OUT ? 3 : 1;                    // planes a block works through
    const int ntiles = (g.planes / CH) << r2_log2;
    constexpr u32 plane_bytes = (u32)((i64)RX * R2 * NC * 8);
    constexpr u32 row_group_bytes = (u32)R2 * NC * 8;              // between the RX rows of a tile
    constexpr int tw_rows = TW_N / (RX * R2);
    const u32 conj = g.inverse ? 0x80000000u : 0u;
    // one tile per block
    const int tile = g.xcd_remap ? xcd_contiguous_block() : (int)blockIdx.x;
    if (tile >= ntiles) return;
    const int unit = tile >> r2_log2, r = tile & (R2 - 1);         // unit = plane, or image (IO_U8_OUT)
    const u32 voff = ((u32)r * NC + (u32)t) * 8u;
    // IO_U8_OUT: a finished channel's bytes, the RX rows of position k packed into one register; the loop over the
    // three channels is a real loop (unrolled, the next plane's loads are hoisted above this plane's butterflies and
    // the registers run out), so the finished channels shift along: at the end pk[0] = B, pk[1] = G, pk[2] = R
    u32 pk[3][8] = {};
#pragma unroll 1
    for (int ch = 0; ch < CH; ch++) {
        const int plane = unit * CH + ch;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(in + plane * g.plane_stride), 0, (int)plane_bytes, 0x00020000);
        c2 v[RX][8];
        if constexpr (IO == IO_U8_IN) {
            // plane p = channel p % 3 of image p / 3; a pixel is 4 bytes where a point is 8
            const int image = plane / 3, shift = 8 * (plane % 3);
            const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc((void*)(g.image_in + (i64)image * (RX * R2) * NC), 0,
                                                                                  (int)(plane_bytes / 2), 0x00020000);
            u32 raw[RX][8];
#pragma unroll
            for (int n1 = 0; n1 < RX; n1++)
#pragma unroll
                for (int k = 0; k < 8; k++) raw[n1][k] = __builtin_amdgcn_raw_buffer_load_b32(rimg, voff / 2, (n1 * row_group_bytes + (T * k) * 8) / 2, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n1 = 0; n1 < RX; n1++)
#pragma unroll
                for (int k = 0; k < 8; k++) v[n1][k] = c2{(float)((raw[n1][k] >> shift) & 0xffu), 0.0f};
        } else {
#pragma unroll
            for (int n1 = 0; n1 < RX; n1++) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
#if MI_FFT2P_DBG == 2
                    const f32x2b x = f32x2b{(float)(t + tile + k), (float)(n1 - tile)};
#else
                    const f32x2b x = __builtin_bit_cast(f32x2b, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(rin, voff, n1 * row_group_bytes + (T * k) * 8, 0));
#endif
                    v[n1][k] = c2{x.x, x.y};
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // every load is issued before anything waits for one
        }
        if (ch == 0) {
#pragma unroll
            for (int i = 0; i < NTW; i++)
                if (t + T * i < P::tw_total) tw[t + T * i] = twr[i];
        }
        __syncthreads();       // (ch > 0: the previous plane's last stage reads no LDS; harmless)
#pragma unroll
        for (int n1 = 0; n1 < RX; n1++) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[n1][k].y = flip_sign(v[n1][k].y, conj);
        }
        // the RX-point step of the column transform, on each of this lane's 8 column positions
        if constexpr (RX > 1 && MI_FFT2P_DBG != 1) {
            c2 wr[RX];
            wr[0] = c2{1.f, 0.f};
#pragma unroll
            for (int n1 = 1; n1 < RX; n1++) wr[n1] = g_twiddle[__builtin_amdgcn_readfirstlane((n1 * r * tw_rows) & (TW_N - 1))];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                c2 x[RX];
#pragma unroll
                for (int n1 = 0; n1 < RX; n1++) x[n1] = (!DIF && n1) ? cmulf(v[n1][k], wr[n1]) : v[n1][k];
                dft<RX>(x);
#pragma unroll
                for (int n1 = 0; n1 < RX; n1++) v[n1][k] = (DIF && n1) ? cmulf(x[n1], wr[n1]) : x[n1];
            }
        }
        if constexpr (MI_FFT2P_DBG != 1) rows_stages<NC, RX>(v, lineA, lineB, tw, t);
        if constexpr (IO == IO_U8_OUT) {
            // NormalizeAndZeroImaginary + ComplexRGBToUchar: real part / div, saturated, one byte of the pixel
            // (the conjugation of an inverse transform does not touch the real part)
            static_assert(IO != IO_U8_OUT || RX <= 4, "one byte per row of the tile in a register");
#pragma unroll
            for (int k = 0; k < 8; k++) {
                u32 w = 0u;
#pragma unroll
                for (int n1 = 0; n1 < RX; n1++) w |= sat_u8_f(v[n1][k].x / g.div) << (8 * n1);
                pk[2][k] = pk[1][k]; pk[1][k] = pk[0][k]; pk[0][k] = w;
            }
        } else {
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(out + plane * g.plane_stride), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
            for (int n1 = 0; n1 < RX; n1++) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const c2 y = v[n1][k];
                    if (MI_FFT2P_DBG != 2 || y.x == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2b, f32x2b{y.x, flip_sign(y.y, conj)}), rout, voff,
                                                          n1 * row_group_bytes + (T * k) * 8, NT ? 2 : 0);
                }
            }
        }
    }
    if constexpr (IO == IO_U8_OUT) {
        const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc((void*)(g.image_out + (i64)unit * (RX * R2) * NC), 0,
                                                                              (int)(plane_bytes / 2), 0x00020000);
#pragma unroll
        for (int n1 = 0; n1 < RX; n1++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                // byte n1 of the three channel registers -> (R, G, B, 255)
                const u32 rg = __builtin_amdgcn_perm(pk[1][k], pk[2][k], 0x0c0c0000u | (u32)n1 | ((u32)(4 + n1) << 8));
                const u32 pixel = __builtin_amdgcn_perm(pk[0][k], rg, 0x0d000100u | ((u32)(4 + n1) << 16));
                __builtin_amdgcn_raw_buffer_store_b32(pixel, rimg, voff / 2, (n1 * row_group_bytes + (T * k) * 8) / 2, 0);
            }
    }
}

// ------------------------------------------------------------------ cols pass
// R2-point transforms down NCOL = 512 / (R2/32) adjacent columns; lane (c, t) holds points
// t + (R2/32) j, j < 32, of column c.  Radix-32 stages, then one of radix R2 / 32^a.  Between stages
// the real parts of the tile go through LDS, then the imaginary parts (4-byte words: half the space).
template <int N>
struct Plan32 {
    static constexpr int L = __builtin_ctz(N);
    static constexpr int A = L / 5;
    static constexpr int REM = 1 << (L % 5);
    static constexpr int n_stages = A + (REM > 1 ? 1 : 0);
    static constexpr int radix(int s) { return s < A ? 32 : REM; }
    static constexpr int ns(int s) { int p = 1; for (int i = 0; i < s; i++) p *= radix(i); return p; }
    static constexpr int tw_off(int s) { int o = 0; for (int i = 1; i < s; i++) o += (radix(i) - 1) * ns(i); return o; }
    static constexpr int tw_total = tw_off(n_stages);
};

template <int N, int NCOL, int S = 0>
MI_DEV void cols_stages(c2 (&v)[32], float* __restrict__ xl, const c2* __restrict__ tw, int c, int t)
{
    using P = Plan32<N>;
    if constexpr (S < P::n_stages) {
        constexpr int R = P::radix(S), NS = P::ns(S), T = N / 32, G = 32 / R;
        constexpr bool LAST = S == P::n_stages - 1;
        // Word index of point i of column c: (i + i / 32) * NCOL + c -- columns interleaved, one spare
        // point per 32.  Written as base + compile-time offset so that the offsets land in the
        // instructions' immediate fields (the bases are two registers, not sixty-four):
        //   a butterfly's outputs d + q NS: NS = 1 -> d = 32 j, no carry; NS >= 32 -> the pad of q NS adds on;
        //   a lane's points t + T k: t < T and T k is a multiple of 32, or t + (T k mod 32) < 32.
        static_assert(NS >= 32 || (NS == 1 && R == 32), "stage shapes of Plan32");
        constexpr int QS = NS >= 32 ? NS + NS / 32 : 1;
        c2 o[32];
        float* dst[G];
#pragma unroll
        for (int m = 0; m < G; m++) {
            const int j = t + T * m, ktw = j % NS;
            c2 x[R];
#pragma unroll
            for (int q = 0; q < R; q++) x[q] = v[m + q * G];
            if constexpr (NS > 1) {
#pragma unroll
                for (int q = 1; q < R; q++) x[q] = cmulf(x[q], tw[P::tw_off(S) + (q - 1) * NS + ktw]);
            }
            dft<R>(x);
#pragma unroll
            for (int q = 0; q < R; q++) {
                if constexpr (LAST) v[m + q * G] = x[q];
                else o[m * R + q] = x[q];
            }
            const int d = (j / NS) * (NS * R) + ktw;
            dst[m] = xl + (d + (d >> 5)) * NCOL + c;
        }
        if constexpr (!LAST) {
            __syncthreads();
#pragma unroll
            for (int m = 0; m < G; m++)
#pragma unroll
                for (int q = 0; q < R; q++) dst[m][q * QS * NCOL] = o[m * R + q].x;
            __syncthreads();
            const float* src = xl + (t + (t >> 5)) * NCOL + c;
#pragma unroll
            for (int k = 0; k < 32; k++) v[k].x = src[(T * k + ((T * k) >> 5)) * NCOL];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < G; m++)
#pragma unroll
                for (int q = 0; q < R; q++) dst[m][q * QS * NCOL] = o[m * R + q].y;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 32; k++) v[k].y = src[(T * k + ((T * k) >> 5)) * NCOL];
        }
        cols_stages<N, NCOL, S + 1>(v, xl, tw, c, t);
    }
}

// One tile = NCOL adjacent columns of one (plane, n1); one block of 512 lanes per tile.
// GAIN: every output is multiplied by the high-pass factor of its position (fftShift, highpassFilter, fftShift
// of imageFFTImageGenerate, imageTools.cu:704-737, as one factor) on its way out.
template <int R2, int RX, int NC, bool DIF, bool NT, bool GAIN = false>
__global__ __launch_bounds__(512, 4)
void fft2p_cols_kernel(const c2* __restrict__ in, c2* __restrict__ out, const TwoPassGeo g)
{
    constexpr int T = R2 / 32, NCOL = 512 / T;
    using P = Plan32<R2>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* xl = reinterpret_cast<float*>(lds_raw);
    c2* tw = reinterpret_cast<c2*>(lds_raw + (size_t)(R2 + R2 / 32) * NCOL * 4);
    const int tid = threadIdx.x, c = tid % NCOL, t = tid / NCOL;
    // twiddle tables of stages 1..: entry e = tid and tid + 512 of the block's table (loads first, LDS
    // writes after the tile's loads are issued: see the rows pass)
    static_assert(P::tw_total <= 2 * 512, "two table entries per lane");
    c2 twr[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + 512 * i;
        int off = 0, ns = 1, idx = 0;
#pragma unroll
        for (int s = 1; s < P::n_stages; s++) {
            ns *= P::radix(s - 1);
            const int R = P::radix(s), n = (R - 1) * ns;
            if (e >= off && e < off + n) idx = (((e - off) / ns + 1) * ((e - off) % ns) * (TW_N / (ns * R))) & (TW_N - 1);
            off += n;
        }
        twr[i] = g_twiddle[idx];
    }
    constexpr int rx_log2 = __builtin_ctz(RX);
    constexpr int cgroups = NC / NCOL, cg_log2 = __builtin_ctz(cgroups);
    const int ntiles = (g.planes << rx_log2) << cg_log2;
    constexpr u32 plane_bytes = (u32)((i64)RX * R2 * NC * 8);
    constexpr u32 row_bytes = (u32)NC * 8;
    // DIT: rows RX*n2 + n1 -> rows R2*n1 + k2;  DIF: rows R2*k1 + n2 -> rows RX*k2 + k1
    constexpr u32 in_stride = DIF ? 1u : (u32)RX, out_stride = DIF ? (u32)RX : 1u;
    constexpr u32 in_step = (u32)T * in_stride * row_bytes, out_step = (u32)T * out_stride * row_bytes;   // between a lane's points
    const u32 conj = g.inverse ? 0x80000000u : 0u;
    {   // one tile per block
        const int tile = g.xcd_remap ? xcd_contiguous_block() : (int)blockIdx.x;
        if (tile >= ntiles) return;
        const int cg = tile & (cgroups - 1), u = tile >> cg_log2, n1 = u & (RX - 1), plane = u >> rx_log2;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(in + plane * g.plane_stride), 0, (int)plane_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(out + plane * g.plane_stride), 0, (int)plane_bytes, 0x00020000);
        const u32 in_base = DIF ? (u32)R2 * n1 : (u32)n1, out_base = DIF ? (u32)n1 : (u32)R2 * n1;
        const u32 col_bytes = (u32)(cg * NCOL + c) * 8u;
        const u32 voff_in = (in_base + (u32)t * in_stride) * row_bytes + col_bytes;
        const u32 voff_out = (out_base + (u32)t * out_stride) * row_bytes + col_bytes;
        c2 v[32];
#pragma unroll
        for (int j = 0; j < 32; j++) {
#if MI_FFT2P_DBG == 2
            const f32x2b x = f32x2b{(float)(t + tile + j), (float)(c - tile)};
#else
            const f32x2b x = __builtin_bit_cast(f32x2b, (u32x2b)__builtin_amdgcn_raw_buffer_load_b64(rin, voff_in, j * in_step, 0));
#endif
            v[j] = c2{x.x, x.y};
        }
        __builtin_amdgcn_sched_barrier(0);      // every load is issued before anything waits for one
        tw[tid] = twr[0];
        if (tid + 512 < P::tw_total) tw[tid + 512] = twr[1];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; j++) v[j].y = flip_sign(v[j].y, conj);
        if constexpr (MI_FFT2P_DBG != 1) cols_stages<R2, NCOL>(v, xl, tw, c, t);
#pragma unroll
        for (int k = 0; k < 32; k++) {
            c2 y = v[k];
            y.y = flip_sign(y.y, conj);
            if constexpr (GAIN) {
                // the filter's geometry is the plane's own (the host checks: images as wide as the plane's rows are
                // long), so a lane's column is its x and only y changes from point to point
                const int x = cg * NCOL + c, row = (int)out_base + (t + T * k) * (int)out_stride;
                const float gain = highpass_gain_xy((float)shifted_coord(x, NC) - (float)NC / 2.0f, row, g.gain_h, g.cutoff);
                y.x *= gain; y.y *= gain;
                if (k % 2 == 1) __builtin_amdgcn_sched_barrier(0);      // two factors at a time: 32 of them at once spill
            }
            if (MI_FFT2P_DBG != 2 || y.x == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2b, f32x2b{y.x, y.y}), rout, voff_out, k * out_step, NT ? 2 : 0);
        }
    }
}

}  // namespace mi
