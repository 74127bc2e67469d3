// gauss_direct.hip -- the direct Gaussian 5x5 and 7x7 (imageGaussianBlur, imageTools.cu:287-311, in the
// reference's own order: ky outer, kx inner, sum = fmaf(pixel, weight, sum) from 0, truncating store), bit-exact
// against the oracle, as wave marches.  These two kernels are bound by vector-instruction issue (k*k fused
// multiply-adds per pixel and channel plus the widening of every tap), so what they cost is decided by how the
// arithmetic is laid out -- which is why they have a translation unit of their own, built with
// -fno-slp-vectorize (build.py FILE_FLAGS), and spell their packed operations themselves:
//   * left to the SLP vectoriser (round 2), half of the multiply-adds became v_pk_fma_f32 with the scalar weight
//     as an aligned register pair: the 7x7 kernel kept its 49 weights spilled in vector-register lanes and
//     fetched them back with 1473 v_readlane per 7 output rows;
//   * 5x5 keeps its window as FLOATS, in the pairs its packed multiply-adds want -- (red, green) of a pixel, and
//     the blue of pixels two columns apart, so that the blue sums of outputs k and k + 2 share an instruction:
//     150 v_pk_fma_f32 per four output pixels instead of 300 scalar ones, and a row is widened once, when it
//     enters (24 + 12 conversions per step instead of 120); 140 window registers, two waves per SIMD;
//   * 7x7 cannot hold floats (7 rows x 10 columns x 3): it keeps packed pixels and widens a row when the
//     accumulation reaches it.  Left alone, the compiler shares those conversions between the unrolled steps and
//     holds the floats across them -- 289 registers, one wave per SIMD; conversions written as volatile asm make
//     each step do its own.  Its weights sit in vector registers: an instruction that reads a scalar register
//     issues more slowly on this part (tools/opbench.hip, profiles/r03_opbench.txt), and two waves per SIMD with
//     vector weights beat three with scalar ones (0.575 against 0.626 ms per 16 x 4K).
// Measured per 16 x 4K (tools/check_gauss_direct.py): 5x5 0.323 -> 0.307 ms, 7x7 0.805 -> 0.554 ms; both land
// at 64-70 TFLOP/s of fp32 multiply-adds, whatever the instruction mix.  Round 2's kernel stays in filters.hip
// (MI_GAUSS_DIRECT_WIN=0) as a second implementation of the same sums.
// 3x3 stays with the other 3x3 kernels in filters.hip (rows in bursts through the LDS ring; it is not bound by
// its arithmetic), 9x9 and up are the LDS-tile kernels there.
#include "common.h"
#include "internal.h"

#ifndef MI_GD_VECTOR_WEIGHTS
#define MI_GD_VECTOR_WEIGHTS 1
#endif
#ifndef MI_GD_PAIRS
#define MI_GD_PAIRS 1      // 5x5: pixels (k, k+2) of a lane share packed multiply-adds
#endif

namespace mi {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct DirectW { float w[49]; };

template <int R, int POL>
__global__ __launch_bounds__(256) void gauss_direct_win_kernel(const u32* __restrict__ in, u32* __restrict__ out,
                                                               const Geo g, const DirectW gw)
{
    constexpr int K = 2 * R + 1, NC = 4 + 2 * R;
    constexpr bool FLOAT_WINDOW = R == 2;
    const WaveJob j = wave_job(g);
    if (!j.valid) return;
    const int cx = j.strip * 256 + 4 * j.lane;
    March mch;                                   // every band marches down: the order over ky is part of the result
    mch.ys = g.y_begin + j.band * g.rows_per_band;
    mch.n = min(mch.ys + g.rows_per_band, g.y_end) - mch.ys;
    mch.dir = 1;
    const RowIO<POL> io(in, out, g, j, mch, j.strip * 256);
    const u32 xm[4] = {xmask(cx, g.W), xmask(cx + 1, g.W), xmask(cx + 2, g.W), xmask(cx + 3, g.W)};

    struct RowIn { u32x4 p, h; };
    auto load_row = [&](int row) {
        RowIn r;
        r.p = io.load(row);
        r.h = io.template load_halo_px<R>(row, 1);
        return r;
    };
    // c[i] = the pixel of column i - R (i = 0 .. 3 + 2R): own 4 pixels plus R neighbours each side
    auto columns = [&](const RowIn& r, u32 (&c)[NC]) {
        u32 col[12];   // columns -4..7 at index c+4
        const u32 p[4] = {r.p.x, r.p.y, r.p.z, r.p.w};
        const u32 a[4] = {r.h.x, r.h.y, r.h.z, r.h.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            col[4 + k] = p[k];
            col[k] = from_left(p[k], k >= 4 - R ? a[(k - (4 - R)) & 3] : 0u);
            col[8 + k] = from_right(p[k], k < R ? a[k] : 0u);
        }
#pragma unroll
        for (int i = 0; i < NC; i++) c[i] = col[4 - R + i];
    };
    auto widen = [](u32 v, float (&f)[3]) {
        f[0] = (float)(v & 0xffu);
        f[1] = (float)((v >> 8) & 0xffu);
        f[2] = (float)((v >> 16) & 0xffu);
    };

    // this step's own conversion (see the head of the file): volatile, so that two steps do not share it
    auto widen_here = [](u32 v, float (&f)[3]) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("v_cvt_f32_ubyte0_e32 %0, %1" : "=v"(f[0]) : "v"(v));
        asm volatile("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(f[1]) : "v"(v));
        asm volatile("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(f[2]) : "v"(v));
#endif
    };

    // 7x7: the weights in vector registers (two waves per SIMD instead of three, and still faster: 0.575 against
    // 0.626 ms per 16 x 4K -- an instruction that reads a scalar register issues more slowly, tools/opbench.hip)
    float wv[K * K];
#pragma unroll
    for (int i = 0; i < K * K; i++) {
        wv[i] = gw.w[i];
        if constexpr (MI_GD_VECTOR_WEIGHTS && !FLOAT_WINDOW) asm volatile("" : "+v"(wv[i]));
    }

    constexpr bool PAIRS = FLOAT_WINDOW && MI_GD_PAIRS != 0;
    // PAIRS: the float window holds, per column, the pair (red, green) and, per column q < NC - 2, the pair (blue of
    // q, blue of q + 2): the red and green sums of an output share one v_pk_fma_f32, and so do the blue sums of
    // outputs k and k + 2 (same weight, pixels two columns apart) -- two multiply-adds in 4.7 cycles where a
    // v_fma_f32 takes 4 for one (tools/opbench.hip).  Each half is the scalar operation: same bits.
    f32x2 wrg[PAIRS ? K : 1][NC], wbb[PAIRS ? K : 1][NC - 2];
    float wf[FLOAT_WINDOW && !PAIRS ? K : 1][NC][3];       // the window as floats (5x5) ...
    // the weights two to a 64-bit register, picked by op_sel; acc += pix * weight[half] in both halves
    f32x2 w2[(K * K + 1) / 2];
#pragma unroll
    for (int i = 0; i < (K * K + 1) / 2; i++) {
        w2[i] = f32x2{gw.w[2 * i], 2 * i + 1 < K * K ? gw.w[2 * i + 1] : 0.f};
        if constexpr (PAIRS && MI_GD_VECTOR_WEIGHTS) asm volatile("" : "+v"(w2[i]));
    }
    auto pk_fma_bcast = [](f32x2& acc, f32x2 pix, f32x2 wpair, int half) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (MI_GD_VECTOR_WEIGHTS) {
            if (half == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(pix), "v"(wpair));
            else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(pix), "v"(wpair));
        } else {
            if (half == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(pix), "s"(wpair));
            else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(pix), "s"(wpair));
        }
#endif
    };
    u32 wp[FLOAT_WINDOW ? 1 : K][NC];            // ... or as packed pixels (7x7)
    auto enter = [&](int slot, const RowIn& r) {
        u32 c[NC];
        columns(r, c);
#pragma unroll
        for (int q = 0; q < NC; q++) {
            if constexpr (PAIRS) {
                float f[3];
                widen(c[q], f);
                wrg[slot][q] = f32x2{f[0], f[1]};
                if (q < NC - 2) wbb[slot][q].x = f[2];
                if (q >= 2) wbb[slot][q - 2].y = f[2];
            } else if constexpr (FLOAT_WINDOW) widen(c[q], wf[slot][q]);
            else wp[slot][q] = c[q];
        }
    };
#pragma unroll
    for (int i = 0; i < K - 1; i++) enter(i, load_row(mch.row(i - R)));
    RowIn nxt = load_row(mch.row(R));
    for (int base = 0; base < mch.n; base += K) {
#pragma unroll
        for (int s = 0; s < K; s++) {
            const int i = base + s;
            if (i < mch.n) {
                const int y = mch.row(i);
                enter((K - 1 + s) % K, nxt);             // row y+R; row y-R+t lives in slot (s+t)%K
                nxt = load_row(mch.row(i + R + 1));
                const u32 rowm = (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u;
                float acc[4][3] = {};
                if constexpr (PAIRS) {
                    f32x2 arg[4] = {}, abb[2] = {};
#pragma unroll
                    for (int t = 0; t < K; t++) {            // ky: top row first, as upstream
#pragma unroll
                        for (int u = 0; u < K; u++) {        // kx
                            const int wi = t * K + u;
#pragma unroll
                            for (int k = 0; k < 4; k++) pk_fma_bcast(arg[k], wrg[(s + t) % K][k + u], w2[wi / 2], wi & 1);
#pragma unroll
                            for (int k = 0; k < 2; k++) pk_fma_bcast(abb[k], wbb[(s + t) % K][k + u], w2[wi / 2], wi & 1);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) { acc[k][0] = arg[k].x; acc[k][1] = arg[k].y; }
#pragma unroll
                    for (int k = 0; k < 2; k++) { acc[k][2] = abb[k].x; acc[k + 2][2] = abb[k].y; }
                } else {
#pragma unroll
                for (int t = 0; t < K; t++) {            // ky: top row first, as upstream
                    float f[NC][3];
#pragma unroll
                    for (int q = 0; q < NC; q++) {
                        if constexpr (FLOAT_WINDOW) {
#pragma unroll
                            for (int c = 0; c < 3; c++) f[q][c] = wf[(s + t) % K][q][c];
                        } else {
                            widen_here(wp[(s + t) % K][q], f[q]);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++)
#pragma unroll
                        for (int u = 0; u < K; u++)      // kx
#pragma unroll
                            for (int c = 0; c < 3; c++)
                                acc[k][c] = __builtin_fmaf(f[k + u][c], wv[t * K + u], acc[k][c]);
                }
                }
                u32 o[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    o[k] = ((u32)acc[k][0] | ((u32)acc[k][1] << 8) | ((u32)acc[k][2] << 16) | 0xFF000000u) & xm[k] & rowm;
                io.store(y, u32x4{o[0], o[1], o[2], o[3]});
            }
        }
    }
}

// weights: the k*k table of generateGaussianKernal (row-major), k = 5 or 7
int launch_gauss_direct_win(int k, int pol, int grid, hipStream_t s, const u32* in, u32* out, const Geo& g, const float* weights)
{
    DirectW gw{};
    for (int i = 0; i < k * k; i++) gw.w[i] = weights[i];
    if (k == 5) {
        if (pol) gauss_direct_win_kernel<2, 1><<<grid, 256, 0, s>>>(in, out, g, gw);
        else gauss_direct_win_kernel<2, 0><<<grid, 256, 0, s>>>(in, out, g, gw);
    } else if (k == 7) {
        if (pol) gauss_direct_win_kernel<3, 1><<<grid, 256, 0, s>>>(in, out, g, gw);
        else gauss_direct_win_kernel<3, 0><<<grid, 256, 0, s>>>(in, out, g, gw);
    } else {
        return fail(MI_ERR_BAD_KERNEL_SIZE, "direct Gaussian march: kernel size %d", k);
    }
    return MI_OK;
}

}  // namespace mi
