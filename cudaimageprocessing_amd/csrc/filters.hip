// filters.hip -- gfx950 kernels for the u8x4 -> u8x4 filters of the reference's imageTools API
// (grayscale, Sobel, Gaussian, mean, median) and their launcher.
//
// Semantics follow kernels/imageTools.cu of the reference (citations at each kernel); the
// execution scheme ("wave march", see common.h) is this library's own.
// Build with -ffp-contract=off: every fused multiply-add is spelled out.
#include <array>
#include <mutex>
#include <vector>
#include <type_traits>

#include "common.h"
#include "internal.h"
#include "median5_network.h"

namespace mi {

// ============================================================================ grayscale
// imageGrayScale, imageTools.cu:217-231: gray = (uchar)(0.299f*R + 0.587f*G + 0.114f*B) on
// every pixel; out = (g,g,g,A_in).  The contraction is pinned:
// fma(.114,B, fma(.299,R, .587*G)).
MI_DEV u32 gray_px(u32 p)
{
    const float r = (float)(p & 0xffu), g = (float)((p >> 8) & 0xffu), b = (float)((p >> 16) & 0xffu);
    const float t = 0.587f * g;
    const float u = __builtin_fmaf(0.299f, r, t);
    const float v = __builtin_fmaf(0.114f, b, u);
    return (u32)v * 0x010101u + (p & 0xFF000000u);
}

// Flat over all pixels.  Each block owns V*256 consecutive 16-byte groups (4 pixels each) and
// each lane V of them; the grid covers the image once (no grid-stride loop): on MI355X a
// streaming pass runs fastest when the dispatcher walks the buffer front to back
// (tools/membench.hip: 6.0-6.4 TB/s against 4.2 TB/s for a 2048-block grid-stride loop).
template <int POL, int V>
__global__ __launch_bounds__(256) void gray_kernel(const u32* __restrict__ in,
                                                   u32* __restrict__ out, i64 n_px)
{
    constexpr bool ALIGNED = POL & 1, NT = POL & 2;
    const i64 groups = (n_px + 3) >> 2;  // groups of 4 pixels; the last one may be ragged
    const i64 base = (i64)blockIdx.x * (256 * V) + threadIdx.x;
    u32x4 v[V];
#pragma unroll
    for (int u = 0; u < V; u++) {
        const i64 gi = base + u * 256;
        if (gi < groups - 1) {
            const u32* p = in + 4 * gi;
            if (ALIGNED) v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p))
                                   : *reinterpret_cast<const u32x4*>(p);
            else v[u] = *reinterpret_cast<const u32x4_u*>(p);
        }
    }
#pragma unroll
    for (int u = 0; u < V; u++) {
        const i64 gi = base + u * 256;
        if (gi < groups - 1) {
            const u32x4 o = {gray_px(v[u].x), gray_px(v[u].y), gray_px(v[u].z), gray_px(v[u].w)};
            u32* q = out + 4 * gi;
            if (ALIGNED) {
                if (NT) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(q));
                else *reinterpret_cast<u32x4*>(q) = o;
            } else {
                *reinterpret_cast<u32x4_u*>(q) = o;
            }
        } else if (gi == groups - 1) {
            for (i64 k = 4 * gi; k < n_px; k++) out[k] = gray_px(in[k]);
        }
    }
}

// ============================================================================ Sobel
// imageSobelEdge, imageTools.cu:255-285.  Red channel only (:274-275).  Separated as
// V(x) = t+2m+b, D(x) = b-t (vertical), Gx = V(x+1)-V(x-1), Gy = D(x-1)+2D(x)+D(x+1); all
// exact in 16-bit integers.  magnitude = trunc(clamp(sqrt(Gx^2+Gy^2),0,255)) (:278-282) is
// evaluated as min(255, (uint)sqrt(s + 0.5f)): s is an integer, so the +0.5 keeps the argument
// >= 1/(4*255) away from every perfect square and v_sqrt_f32's 1-ulp error cannot cross it
// (swept exhaustively by tests/test_gpu_parity.py through mi_debug_sobel_mag_table_dev).
MI_DEV u32 sobel_mag(int s)
{
    const float f = __builtin_amdgcn_sqrtf((float)s + 0.5f);
    u32 m = (u32)f;
    return m > 255u ? 255u : m;
}

// (m, m, m, 255) from m <= 255 in one v_perm_b32: selector bytes 0 = m's byte 0, 0x0d = 0xFF
MI_DEV u32 gray_opaque(u32 m) { return __builtin_amdgcn_perm(0u, m, 0x0d000000u); }

MI_DEV void red_pairs(u32x4 p, u32& r01, u32& r23)
{   // (r0 | r1<<16), (r2 | r3<<16): the red bytes of 4 pixels as packed u16
    r01 = __builtin_amdgcn_perm(p.y, p.x, 0x0c040c00u);
    r23 = __builtin_amdgcn_perm(p.w, p.z, 0x0c040c00u);
}

#ifndef MI_SEP_BURST
#define MI_SEP_BURST 3    // rows per request burst of the separable Gaussian (fp32) and mean kernels; 0: one row ahead, in registers
#endif
#ifndef MI_FEED_K
#define MI_FEED_K 2       // rows per request burst of the 3x3 march kernels (RowFeed in common.h); 0: one row ahead, in registers
                          // (measured: with their 6-row bands bursts of 2 change nothing, bursts of 3 or 4 cost 3..5 %)
#endif

// Common prologue of the wave-march kernels: this wave's strip, band and I/O.
#define MI_MARCH_PROLOGUE()                                                                      \
    const WaveJob j = wave_job(g);                                                               \
    if (!j.valid) return;                                                                        \
    const int cx = j.strip * 256 + 4 * j.lane;   /* first of this lane's 4 columns */           \
    const March mch = make_march(g, j);                                                          \
    const RowIO<POL> io(in, out, g, j, mch, j.strip * 256);                                      \
    const u32 xm[4] = {xmask(cx, g.W), xmask(cx + 1, g.W), xmask(cx + 2, g.W), xmask(cx + 3, g.W)}

template <int POL>
__global__ __launch_bounds__(256) void sobel_kernel(const u32* __restrict__ in,
                                                    u32* __restrict__ out, const Geo g)
{
    MI_MARCH_PROLOGUE();
    // Red byte of the halo pixel in both halves of a packed pair: the low half is what lane 63
    // uses (first pixel right of the strip), the high half what lane 0 uses (last pixel left of it).
    auto halo_red = [](u32 h) { return (h & 0xffu) * 0x00010001u; };

    u32 t01, t23, m01, m23, b01, b23, th, mh, bh;
    // t = oldest row of the window, b = newest.  Marching up swaps their roles, which only flips
    // the sign of Gy; Gx^2 + Gy^2 does not change.
    typedef RowFeed<POL, MI_FEED_K, 1> Feed;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[Feed::BLOCK_BYTES];
    Feed feed(io, mch, ring, j.lane);
    feed.start(-1);
    typename Feed::Row r = feed.next(0, 2);
    red_pairs(r.p, t01, t23);
    th = halo_red(r.h);
    r = feed.next(1, 2);
    red_pairs(r.p, m01, m23);
    mh = halo_red(r.h);
    for (int i = 0; i < mch.n; i++) {
        const int y = mch.row(i);
        r = feed.next(i + 2, 2);         // row i + 1; the rows of the next steps are requested in bursts
        red_pairs(r.p, b01, b23);
        bh = halo_red(r.h);
        const u32 V01 = pk_add(pk_add(t01, b01), pk_add(m01, m01));
        const u32 V23 = pk_add(pk_add(t23, b23), pk_add(m23, m23));
        const u32 VH = pk_add(pk_add(th, bh), pk_add(mh, mh));
        const u32 D01 = pk_sub(b01, t01), D23 = pk_sub(b23, t23), DH = pk_sub(bh, th);
        const u32 VL = from_left(V23, VH), VR = from_right(V01, VH);
        const u32 DL = from_left(D23, DH), DR = from_right(D01, DH);
        const u32 Va = __builtin_amdgcn_alignbit(V01, VL, 16);  // (V-1, V0)
        const u32 Vb = __builtin_amdgcn_alignbit(V23, V01, 16); // (V1, V2)
        const u32 Vc = __builtin_amdgcn_alignbit(VR, V23, 16);  // (V3, V4)
        const u32 Da = __builtin_amdgcn_alignbit(D01, DL, 16);
        const u32 Db = __builtin_amdgcn_alignbit(D23, D01, 16);
        const u32 Dc = __builtin_amdgcn_alignbit(DR, D23, 16);
        const u32 Gx01 = pk_sub(Vb, Va), Gx23 = pk_sub(Vc, Vb);
        const u32 Gy01 = pk_add(pk_add(Da, Db), pk_add(D01, D01));
        const u32 Gy23 = pk_add(pk_add(Db, Dc), pk_add(D23, D23));
        // (Gx_j, Gy_j) as one s16x2, s_j = Gx_j^2 + Gy_j^2 with one dot2
        const u32 P[4] = {__builtin_amdgcn_perm(Gy01, Gx01, 0x05040100u), __builtin_amdgcn_perm(Gy01, Gx01, 0x07060302u),
                          __builtin_amdgcn_perm(Gy23, Gx23, 0x05040100u), __builtin_amdgcn_perm(Gy23, Gx23, 0x07060302u)};
        const u32 rowm = (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u;
        u32 o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const s16x2 v = __builtin_bit_cast(s16x2, P[k]);
            o[k] = gray_opaque(sobel_mag(__builtin_amdgcn_sdot2(v, v, 0, false))) & xm[k] & rowm;
        }
        io.store(y, u32x4{o[0], o[1], o[2], o[3]});
        t01 = m01; t23 = m23; th = mh; m01 = b01; m23 = b23; mh = bh;
    }
}

// The fused chain carries its median and Sobel rows as packed f16 pairs (every value there is an
// integer of magnitude <= 1020, exact in f16; non-negative f16 values order like their bit
// patterns, so the median's packed u16 min/max work on them unchanged).  With (Gx, Gy) as one
// f16 pair, s + 0.5 is a single dot2 with the 0.5 as its accumulator -- exact in f32 -- and the
// magnitude is the same min(255, (uint)sqrt(s + 0.5f)) as above.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
MI_DEV u32 sobel_mag_h(u32 gxgy)
{
    const h16x2 v = __builtin_bit_cast(h16x2, gxgy);
    const float d = __builtin_amdgcn_fdot2(v, v, 0.5f, false);
    const float f = __builtin_amdgcn_sqrtf(d);
    // floor(f), saturated at 255, in one conversion: v_cvt_pk_u8_f32 rounds to nearest and saturates, and
    // f - 0.5 is never within 0.001 of a tie (f = sqrt(s + 0.5) stays that far from every integer for
    // s < 256^2; beyond, it saturates) -- one slot where v_cvt_u32_f32 + v_min_u32 take three
    return __builtin_amdgcn_cvt_pk_u8_f32(f - 0.5f, 0u, 0u);
}

MI_DEV u32 pk_f16(float lo, float hi) { return __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
MI_DEV u32 h_add(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_bit_cast(h16x2, a) + __builtin_bit_cast(h16x2, b)); }
MI_DEV u32 h_sub(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_bit_cast(h16x2, a) - __builtin_bit_cast(h16x2, b)); }
// 2a + b
MI_DEV u32 h_fma2(u32 a, u32 b)
{
    const h16x2 two = {(_Float16)2.0f, (_Float16)2.0f};
    return __builtin_bit_cast(u32, __builtin_elementwise_fma(two, __builtin_bit_cast(h16x2, a), __builtin_bit_cast(h16x2, b)));
}

// out[(gy + gmax) * (2 gmax + 1) + gx + gmax] = the magnitude of (gx, gy)
__global__ void sobel_pairs_kernel(unsigned char* out, int gmax, int variant)
{
    const int n = 2 * gmax + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * n) return;
    const h16x2 v = {(_Float16)(float)(i % n - gmax), (_Float16)(float)(i / n - gmax)};
    out[i] = (unsigned char)sobel_mag_h(__builtin_bit_cast(u32, v));
}

__global__ void sobel_table_kernel(unsigned char* out, u32 count)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (unsigned char)sobel_mag((int)i);
}

// ============================================================================ mean 3x3
// imageMeanBlur, imageTools.cu:313-340, at the wrapper's k=3 (:806): int sums of the 9 taps
// per channel, integer division by 9, alpha 255.  Channels travel as packed u16 pairs
// lo=(R,B), hi=(G,A); s/9 == (s*7282)>>16 exactly for s < 32768 (s <= 2295 here).
MI_DEV void unpack_px(u32 p, u32& lo, u32& hi)
{
    lo = p & 0x00FF00FFu;
    hi = (p >> 8) & 0x00FF00FFu;
}

MI_DEV u32 div9_pack(u32 lo, u32 hi)
{
    const u32 qr = ((lo & 0xFFFFu) * 7282u) >> 16;
    const u32 qb = ((lo >> 16) * 7282u) & 0x00FF0000u;
    const u32 qg = (((hi & 0xFFFFu) * 7282u) >> 8) & 0x0000FF00u;
    return qr | qg | qb | 0xFF000000u;
}

template <int POL>
__global__ __launch_bounds__(256) void mean3_kernel(const u32* __restrict__ in,
                                                    u32* __restrict__ out, const Geo g)
{
    MI_MARCH_PROLOGUE();

    // horizontal 3-sums of one row, per pixel, as (lo, hi) packed pairs
    auto hsum = [&](u32x4 p, u32 h, u32 (&hl)[4], u32 (&hh)[4]) {   // h: the one-pixel halo
        u32 lo[6], hi[6];
        unpack_px(from_left(p.w, h), lo[0], hi[0]);
        unpack_px(p.x, lo[1], hi[1]);
        unpack_px(p.y, lo[2], hi[2]);
        unpack_px(p.z, lo[3], hi[3]);
        unpack_px(p.w, lo[4], hi[4]);
        unpack_px(from_right(p.x, h), lo[5], hi[5]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            hl[k] = pk_add(pk_add(lo[k], lo[k + 1]), lo[k + 2]);
            hh[k] = pk_add(pk_add(hi[k], hi[k + 1]), hi[k + 2]);
        }
    };

    u32 tl[4], th[4], ml[4], mh[4], bl[4], bh[4];
    typedef RowFeed<POL, MI_FEED_K, 1> Feed;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[Feed::BLOCK_BYTES];
    Feed feed(io, mch, ring, j.lane);
    feed.start(-1);
    typename Feed::Row r = feed.next(0, 2);
    hsum(r.p, r.h, tl, th);
    r = feed.next(1, 2);
    hsum(r.p, r.h, ml, mh);
    for (int i = 0; i < mch.n; i++) {
        const int y = mch.row(i);
        r = feed.next(i + 2, 2);
        hsum(r.p, r.h, bl, bh);
        const u32 rowm = (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u;
        u32 o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 sl = pk_add(pk_add(tl[k], ml[k]), bl[k]);
            const u32 sh = pk_add(pk_add(th[k], mh[k]), bh[k]);
            o[k] = div9_pack(sl, sh) & xm[k] & rowm;
            tl[k] = ml[k]; th[k] = mh[k]; ml[k] = bl[k]; mh[k] = bh[k];
        }
        io.store(y, u32x4{o[0], o[1], o[2], o[3]});
    }
}

// ============================================================================ median 3x3
// Median blur (advertised in the reference's README.md:3, no reference code): per channel
// the 5th smallest of the 9 flat taps, alpha 255, same write set as the other stencils.
// Wave-register sorting network on packed u16 pairs (v_pk_min_u16 / v_pk_max_u16): each
// 3-pixel column is sorted once per row and shared by the three outputs that use it
// (the neighbours' sorted edge columns arrive by DPP); then
// median9 = med3( max3(lows), med3(mids), min3(highs) ).
MI_DEV void sort2(u32& a, u32& b)
{
    const u32 lo = pk_min(a, b), hi = pk_max(a, b);
    a = lo; b = hi;
}
// Median of three per 16-bit half.  There is no packed median, but v_med3_u16 on each half
// (op_sel picks the high halves and writes the high half of the destination, keeping the low
// one) is two operations where the packed min/max network takes four.
MI_DEV u32 med3(u32 a, u32 b, u32 c)
{
    u32 d;
    asm("v_med3_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    asm("v_med3_u16 %0, %1, %2, %3 op_sel:[1,1,1,1]" : "+v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Register layout: R and B of each pixel share a register (R | B<<16), the G bytes of two
// adjacent pixels share one (G0 | G1<<16): 6 registers per 4 pixels, nothing spent on alpha.
// Row r of the 3-row window lives in ring slot r % 3 (the row loop is unrolled by 3, so every
// index is static and rows are never copied).
template <int POL>
__global__ __launch_bounds__(256) void median3_kernel(const u32* __restrict__ in,
                                                      u32* __restrict__ out, const Geo g)
{
    MI_MARCH_PROLOGUE();
    constexpr u32 SEL_G = 0x0c050c01u;   // (byte 1 of src1) | (byte 1 of src0) << 16

    // slot layout: 0..3 = RB of own pixels, 4 = RB of the halo pixel (left of the strip in lane
    // 0, right of it in lane 63), 5 = G of pixels 0,1, 6 = G of pixels 2,3, 7 = G of the halo
    // pixels (low half: right halo, high half: left halo)
    auto unpack = [&](u32x4 p, u32 h, u32 (&w)[8]) {   // h: the one-pixel halo
        w[0] = p.x & 0x00FF00FFu; w[1] = p.y & 0x00FF00FFu; w[2] = p.z & 0x00FF00FFu; w[3] = p.w & 0x00FF00FFu;
        w[4] = h & 0x00FF00FFu;
        w[5] = __builtin_amdgcn_perm(p.y, p.x, SEL_G);
        w[6] = __builtin_amdgcn_perm(p.w, p.z, SEL_G);
        w[7] = __builtin_amdgcn_perm(h, h, SEL_G);
    };

    u32 win[3][8];
    typedef RowFeed<POL, MI_FEED_K, 1> Feed;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[Feed::BLOCK_BYTES];
    Feed feed(io, mch, ring, j.lane);
    feed.start(-1);
    typename Feed::Row r = feed.next(0, 2);
    unpack(r.p, r.h, win[0]);
    r = feed.next(1, 2);
    unpack(r.p, r.h, win[1]);
    for (int base = 0; base < mch.n; base += 3) {
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const int i = base + s;
            if (i < mch.n) {
            const int y = mch.row(i);
            r = feed.next(i + 2, 2);
            unpack(r.p, r.h, win[(s + 2) % 3]);          // input step i+1
            // sort every 3-pixel column: lo <= mid <= hi
            u32 lo[8], mid[8], hi[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                u32 a = win[s % 3][k], b = win[(s + 1) % 3][k], c = win[(s + 2) % 3][k];
                sort2(a, b); sort2(b, c); sort2(a, b);
                lo[k] = a; mid[k] = b; hi[k] = c;
            }
            // ---- R,B: one register per pixel; columns -1..4 = {left neighbour's 3, own 0..3, right's 0}
            u32 rb[4];
            {
                u32 cl[6], cm[6], ch[6];
#pragma unroll
                for (int k = 0; k < 4; k++) { cl[k + 1] = lo[k]; cm[k + 1] = mid[k]; ch[k + 1] = hi[k]; }
                cl[0] = from_left(lo[3], lo[4]); cm[0] = from_left(mid[3], mid[4]); ch[0] = from_left(hi[3], hi[4]);
                cl[5] = from_right(lo[0], lo[4]); cm[5] = from_right(mid[0], mid[4]); ch[5] = from_right(hi[0], hi[4]);
                // the max of the lows / min of the highs / median of the mids of 3 adjacent columns;
                // outputs 0,1 share the pair of columns (1,2), outputs 2,3 the pair (3,4)
                const u32 lp = pk_max(cl[1], cl[2]), lq = pk_max(cl[3], cl[4]);
                const u32 hp = pk_min(ch[1], ch[2]), hq = pk_min(ch[3], ch[4]);
                const u32 mpn = pk_min(cm[1], cm[2]), mpx = pk_max(cm[1], cm[2]);
                const u32 mqn = pk_min(cm[3], cm[4]), mqx = pk_max(cm[3], cm[4]);
                const u32 a[4] = {pk_max(cl[0], lp), pk_max(lp, cl[3]), pk_max(cl[2], lq), pk_max(lq, cl[5])};
                const u32 c[4] = {pk_min(ch[0], hp), pk_min(hp, ch[3]), pk_min(ch[2], hq), pk_min(hq, ch[5])};
                const u32 b[4] = {pk_max(pk_min(cm[0], mpx), mpn), pk_max(pk_min(cm[3], mpx), mpn),
                                  pk_max(pk_min(cm[2], mqx), mqn), pk_max(pk_min(cm[5], mqx), mqn)};
#pragma unroll
                for (int k = 0; k < 4; k++) rb[k] = med3(a[k], b[k], c[k]);
            }
            // ---- G: two pixels per register; shifted pairs (-1,0), (1,2), (3,4) by alignbit
            u32 g01, g23;
            {
                const u32 lL = from_left(lo[6], lo[7]), mL = from_left(mid[6], mid[7]), hL = from_left(hi[6], hi[7]);
                const u32 lR = from_right(lo[5], lo[7]), mR = from_right(mid[5], mid[7]), hR = from_right(hi[5], hi[7]);
                const u32 la = __builtin_amdgcn_alignbit(lo[5], lL, 16), lb = __builtin_amdgcn_alignbit(lo[6], lo[5], 16), lc = __builtin_amdgcn_alignbit(lR, lo[6], 16);
                const u32 ma = __builtin_amdgcn_alignbit(mid[5], mL, 16), mb = __builtin_amdgcn_alignbit(mid[6], mid[5], 16), mc = __builtin_amdgcn_alignbit(mR, mid[6], 16);
                const u32 ha = __builtin_amdgcn_alignbit(hi[5], hL, 16), hb = __builtin_amdgcn_alignbit(hi[6], hi[5], 16), hc = __builtin_amdgcn_alignbit(hR, hi[6], 16);
                g01 = med3(pk_max(pk_max(la, lo[5]), lb), med3(ma, mid[5], mb), pk_min(pk_min(ha, hi[5]), hb));
                g23 = med3(pk_max(pk_max(lb, lo[6]), lc), med3(mb, mid[6], mc), pk_min(pk_min(hb, hi[6]), hc));
            }
            const u32 rowm = (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u;
            // (R, G, B, 255): R and B from rb (bytes 0, 2), G from the pair register, 0xFF constant
            u32x4 o;
            o.x = __builtin_amdgcn_perm(g01, rb[0], 0x0d020400u) & xm[0] & rowm;
            o.y = __builtin_amdgcn_perm(g01, rb[1], 0x0d020600u) & xm[1] & rowm;
            o.z = __builtin_amdgcn_perm(g23, rb[2], 0x0d020400u) & xm[2] & rowm;
            o.w = __builtin_amdgcn_perm(g23, rb[3], 0x0d020600u) & xm[3] & rowm;
            io.store(y, o);
            }
        }
    }
}

// ============================================================================ Gaussian, separable
// imageGaussianBlur, imageTools.cu:287-311, as a row pass followed by a column pass with the
// normalised 1-D weights w (w x w^T equals generateGaussianKernal's 2-D kernel up to rounding).
// The row pass runs on every input row as it is loaded; the K most recent row-pass results
// stay in registers and feed the column pass.  fp32 throughout, truncating store (:306-308),
// alpha 255, flat-index taps.  Differences against the reference's 2-D summation order are
// rounding only: max |delta| = 1 LSB after truncation.
struct GaussW { float w[16]; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
MI_DEV f32x2 splat2(float v) { return f32x2{v, v}; }

// A single wave per SIMD issues VALU work at about half rate on this part (tools/valubench.hip), so
// the wide kernels, whose window would take the whole register file, are held to 256 registers
// (two waves per SIMD); for R = 6, 7 that spills a few values to scratch and is still faster.
template <int R, int POL>
__global__ __launch_bounds__(256, (R >= 5 ? 2 : 1)) void gauss_sep_kernel(const u32* __restrict__ in,
                                                        u32* __restrict__ out, const Geo g,
                                                        const GaussW gw)
{
    constexpr int K = 2 * R + 1;
    constexpr int HOPS = (R + 3) / 4;   // 4-pixel groups needed beyond each edge of the strip
    constexpr int N1 = R < 4 ? R : 4, N2 = R > 4 ? R - 4 : 1;   // pixels used of halo groups 1 and 2
    MI_MARCH_PROLOGUE();

    // input rows in bursts through a per-wave LDS ring (RowFeed, common.h) with the 4-pixel groups beside the strip
    // (3x3: 6-row bands, where bursts gain nothing and the ring would cap the occupancy: one row ahead, in registers)
    typedef RowFeed<POL, (R == 1 ? MI_FEED_K : MI_SEP_BURST), 4 * HOPS> Feed;
    typedef typename Feed::Row RowIn;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[Feed::BLOCK_BYTES];
    Feed feed(io, mch, ring, j.lane);
    const bool first_lane = j.lane == 0;

    // Row pass: h[3*p+c] = sum_t w[t] * channel c of columns (2p - R + t, 2p + 1 - R + t), p = 0,1:
    // two adjacent output pixels per packed fp32 operation (v_pk_mul_f32 / v_pk_fma_f32, each
    // half rounded exactly like the scalar operation).
    auto hrow = [&](const RowIn& r, f32x2 (&h)[6]) {
        u32 col[20];  // columns -8..11 at index c+8; unused entries are dead code
        const u32 p[4] = {r.p.x, r.p.y, r.p.z, r.p.w};
        // halo group pixels: element i is the group's pixel 4-N+i in lane 0 (left of the strip) and
        // its pixel i in lane 63 (right of it); the shifts only read those two lanes' edge values
        const u32 n4[4] = {r.near.x, r.near.y, r.near.z, r.near.w}, f4[4] = {r.far.x, r.far.y, r.far.z, r.far.w};
        u32 a1[4] = {0u, 0u, 0u, 0u}, a2[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < N1; i++) a1[i] = first_lane ? n4[4 - N1 + i] : n4[i];
        if constexpr (HOPS > 1) {
#pragma unroll
            for (int i = 0; i < N2; i++) a2[i] = first_lane ? f4[4 - N2 + i] : f4[i];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            col[8 + k] = p[k];
            col[4 + k] = from_left(p[k], k >= 4 - N1 ? a1[(k - (4 - N1)) & 3] : 0u);
            col[12 + k] = from_right(p[k], k < N1 ? a1[k] : 0u);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            col[k] = from_left(col[4 + k], (HOPS > 1 && k >= 4 - N2) ? a2[(k - (4 - N2)) & 3] : 0u);
            col[16 + k] = from_right(col[12 + k], (HOPS > 1 && k < N2) ? a2[k] : 0u);
        }
        float f[4 + 2 * R][3];
#pragma unroll
        for (int i = 0; i < 4 + 2 * R; i++) {
            const u32 v = col[8 - R + i];
            f[i][0] = (float)(v & 0xffu);
            f[i][1] = (float)((v >> 8) & 0xffu);
            f[i][2] = (float)((v >> 16) & 0xffu);
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                f32x2 s = splat2(gw.w[0]) * f32x2{f[2 * q][c], f[2 * q + 1][c]};
#pragma unroll
                for (int t = 1; t < K; t++)
                    s = __builtin_elementwise_fma(splat2(gw.w[t]), f32x2{f[2 * q + t][c], f[2 * q + 1 + t][c]}, s);
                h[3 * q + c] = s;
            }
    };

    f32x2 win[K][6];
    // input rows row(-R) .. row(R-1) fill slots 0 .. K-2 (the weights are symmetric, so the
    // marching direction does not matter to the column pass); the first store follows row(R)
    feed.start(-R);
#pragma unroll
    for (int i = 0; i < K - 1; i++)
        hrow(feed.next(i, K - 1), win[i]);
    constexpr int UNR = K;
    for (int base = 0; base < mch.n; base += UNR) {
#pragma unroll
        for (int s = 0; s < UNR; s++) {
            const int i = base + s;
            if (i < mch.n) {
            const int y = mch.row(i);
            // newest input row y+R goes to slot (K-1+s)%K; tap t of the column pass is slot (s+t)%K
            hrow(feed.next(i + K - 1, K - 1), win[(K - 1 + s) % K]);
            const u32 rowm = (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u;
            u32 o[4];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                f32x2 acc[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    // centre tap, then the symmetric pairs (w[t] == w[K-1-t]): the sum does not
                    // depend on whether the band marches down or up, so any decomposition into
                    // bands or strips gives bit-identical output
                    f32x2 a = splat2(gw.w[R]) * win[(s + R) % K][3 * q + c];
#pragma unroll
                    for (int t = 0; t < R; t++)
                        a = __builtin_elementwise_fma(splat2(gw.w[t]), win[(s + t) % K][3 * q + c] + win[(s + K - 1 - t) % K][3 * q + c], a);
                    acc[c] = a;
                }
                o[2 * q] = ((u32)acc[0].x | ((u32)acc[1].x << 8) | ((u32)acc[2].x << 16) | 0xFF000000u) & xm[2 * q] & rowm;
                o[2 * q + 1] = ((u32)acc[0].y | ((u32)acc[1].y << 8) | ((u32)acc[2].y << 16) | 0xFF000000u) & xm[2 * q + 1] & rowm;
            }
            io.store(y, u32x4{o[0], o[1], o[2], o[3]});
            }
        }
    }
}

// ============================================================================ Gaussian, separable, 7x7 .. 15x15
// The kernel above keeps its window of K row-pass results as fp32: K x 12 registers per lane, 180 for
// 15x15 -- the whole register file at two waves per SIMD, 79 of them spilled to scratch, and the vector
// ALUs idle half the time (profiles/r02_valu_roofline.md).  Here (kernel sizes 7 to 15; measured against the
// fp32 kernel on 16 4K frames: 7x7 0.245 -> 0.217 ms, 9x9 0.272 -> 0.230, 13x13 0.420 -> 0.303, 15x15
// 0.502 -> 0.358) the window is 16-bit fixed point, two ROWS per register:
//   * the row pass sums 16-bit pairs of adjacent pixels with v_dot2_u32_u16 and 0.16 fixed-point weights
//     whose sum is exactly 2^16 (the centre weight takes the rounding residue); its result, rounded to
//     8.8 fixed point (|error| <= 1/512), is what the window keeps;
//   * arrivals 2j and 2j+1 (rows in marching order) share a register; a step's window is R + 1
//     consecutive pairs -- (R + 1) x 12 registers, 96 for 15x15 -- whatever the parity of the step:
//     an even step weights them (w0,w1) (w2,w3) .. (w2R,0), an odd one (0,w0) (w1,w2) .. (w2R-1,w2R);
//   * the column pass is R + 1 v_dot2_u32_u16 per value with the same weights: the accumulator is the
//     result in 8.24 fixed point, its top byte the truncated output (imageTools.cu:306-308 truncates too).
//     Integer sums do not depend on the order: any decomposition into bands or strips gives the same bytes.
// Error against the exact real-valued sum: each pass 255 * sum |w_t - wc_t / 2^16| <= 255 * 2K * 2^-17
// = 0.03, plus 1/512 from the 8.8 rows: < 0.07 in all, far inside the +-1 LSB the separable form is
// allowed; a constant image comes out exactly constant.  (fp32 multiply-adds cost one issue slot per
// tap, a 16-bit dot product 0.8, and the byte -> float conversions go away: profiles/r02_valubench.txt.)
struct GaussWide { u32 wc_even[8]; u32 wc_odd[8]; };

#ifndef MI_GW_WAVES
#define MI_GW_WAVES 2
#endif
#ifndef MI_GW_BURST
#define MI_GW_BURST 3     // rows per request burst (RowFeed; 2, 3 and 4 measure the same, 6 and 8 cost 1..8 %); 0: one row ahead, in registers
#endif
template <int R, int POL>
__global__ __launch_bounds__(256, MI_GW_WAVES) void gauss_wide_kernel(const u32* __restrict__ in, u32* __restrict__ out,
                                                            const Geo g, const GaussWide gw)
{
    constexpr int NP = R + 1;
    constexpr int HOPS = (R + 3) / 4;
    constexpr int N1 = R < 4 ? R : 4, N2 = R > 4 ? R - 4 : 1;
    MI_MARCH_PROLOGUE();

    // input rows in bursts through a per-wave LDS ring (RowFeed, common.h), with the one or two 4-pixel groups
    // beside the strip; a1 / a2 below: element i = the nearer / farther group's pixel 4-N+i in lane 0, i in lane 63
    typedef RowFeed<POL, MI_GW_BURST, 4 * HOPS> Feed;
    typedef typename Feed::Row RowIn;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[Feed::BLOCK_BYTES];
    Feed feed(io, mch, ring, j.lane);
    const bool first_lane = j.lane == 0;
    // Row pass, in integers too: the strip's pixels and their neighbours (columns -8 .. 11) as 16-bit pairs of
    // horizontally adjacent pixels of one channel (one v_perm_b32 per pair), R + 1 dot products per output
    // with the same 0.16 weights -- shifted by one for the outputs whose first tap is the second half of a
    // pair.  h[v], v = 2 * (3 * (k / 2) + c) + k % 2 for pixel k and channel c, is the row sum in 8.16 fixed
    // point plus 128 (so that cutting it to 8.8 below rounds to nearest).
    auto irow = [&](const RowIn& r, u32 (&h)[12]) __attribute__((always_inline)) {
        u32 col[20];
        const u32 p[4] = {r.p.x, r.p.y, r.p.z, r.p.w};
        const u32 n4[4] = {r.near.x, r.near.y, r.near.z, r.near.w}, f4[4] = {r.far.x, r.far.y, r.far.z, r.far.w};
        u32 a1[4] = {0u, 0u, 0u, 0u}, a2[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < N1; i++) a1[i] = first_lane ? n4[4 - N1 + i] : n4[i];
        if constexpr (HOPS > 1) {
#pragma unroll
            for (int i = 0; i < N2; i++) a2[i] = first_lane ? f4[4 - N2 + i] : f4[i];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            col[8 + k] = p[k];
            col[4 + k] = from_left(p[k], k >= 4 - N1 ? a1[(k - (4 - N1)) & 3] : 0u);
            col[12 + k] = from_right(p[k], k < N1 ? a1[k] : 0u);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            col[k] = from_left(col[4 + k], (HOPS > 1 && k >= 4 - N2) ? a2[(k - (4 - N2)) & 3] : 0u);
            col[16 + k] = from_right(col[12 + k], (HOPS > 1 && k < N2) ? a2[k] : 0u);
        }
        constexpr int P0 = (8 - R) / 2, P1 = (11 + R) / 2;      // pairs of columns 2j-8, 2j-7 that some output reads
        u32 pair[10][3];
#pragma unroll
        for (int jp = P0; jp <= P1; jp++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                pair[jp][c] = __builtin_amdgcn_perm(col[2 * jp + 1], col[2 * jp], 0x0c040c00u + (u32)c * 0x00010001u);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int ci0 = k - R + 8;                          // the first tap's column, counted from -8
            const u32* wcr = (ci0 % 2 == 0) ? gw.wc_even : gw.wc_odd;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                u32 a = 128u;
#pragma unroll
                for (int t = 0; t < NP; t++)
                    a = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pair[ci0 / 2 + t][c]), __builtin_bit_cast(u16x2, wcr[t]), a, false);
                h[2 * (3 * (k / 2) + c) + (k % 2)] = a;
            }
        }
    };

    // win[slot][v]: pair j (arrivals 2j, 2j+1) lives in slot j % NP
    u32 win[NP][12];
    auto arrive = [&](const RowIn& r, int a) __attribute__((always_inline)) {   // a = arrival index (static after unrolling)
        u32 h[12];
        irow(r, h);
        const int slot = (a / 2) % NP;
        if (a % 2 == 0) {
#pragma unroll
            for (int e = 0; e < 12; e++) win[slot][e] = __builtin_amdgcn_perm(0u, h[e], 0x0c0c0201u);         // (8.8 of this row, 0)
        } else {   // the partner of the even row already in the slot's low half
#pragma unroll
            for (int e = 0; e < 12; e++) win[slot][e] = __builtin_amdgcn_perm(h[e], win[slot][e], 0x06050100u);   // (even row, odd row)
        }
    };
    // arrivals 0 .. 2R-1 = input rows row(-R) .. row(R-1); the first store follows arrival 2R
    feed.start(-R);
#pragma unroll
    for (int a = 0; a < 2 * R; a++) arrive(feed.next(a, 2 * R), a);

    constexpr int UNR = 2 * NP;      // pair slots and parities repeat with this period
    for (int base = 0; base < mch.n; base += UNR) {
#pragma unroll
        for (int s = 0; s < UNR; s++) {
            const int i = base + s;
            if (i < mch.n) {
            const int y = mch.row(i);
            // arrival i + 2R (2R is even: its parity and, up to a constant, its slot are those of s)
            arrive(feed.next(i + 2 * R, 2 * R), s + 2 * R);
            const u32 rowm = (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u;
            // the window: pairs first .. first + R, first = floor(s / 2) (+ R * k, which the ring hides)
            const u32* wc = (s % 2 == 0) ? gw.wc_even : gw.wc_odd;
            u32 acc[12];
#pragma unroll
            for (int v = 0; v < 12; v++) {
                u32 a = 0u;
#pragma unroll
                for (int t = 0; t < NP; t++)
                    a = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, win[(s / 2 + t) % NP][v]), __builtin_bit_cast(u16x2, wc[t]), a, false);
                acc[v] = a;
            }
            u32 o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // acc index of pixel k, channel c: 2 * (3 * (k / 2) + c) + (k % 2); output byte = top byte
                const u32 r8 = acc[2 * (3 * (k / 2) + 0) + (k % 2)], g8 = acc[2 * (3 * (k / 2) + 1) + (k % 2)],
                          b8 = acc[2 * (3 * (k / 2) + 2) + (k % 2)];
                const u32 rg = __builtin_amdgcn_perm(g8, r8, 0x0c0c0703u);          // (R, G, 0, 0)
                o[k] = __builtin_amdgcn_perm(b8, rg, 0x0d070100u) & xm[k] & rowm;   // (R, G, B, 255)
            }
            io.store(y, u32x4{o[0], o[1], o[2], o[3]});
            }
        }
    }
}

// ============================================================================ mean, 5x5 .. 15x15
// imageMeanBlur, imageTools.cu:313-340, for kernel sizes 5..15 (the reference's wrapper
// hard-codes 3, :806; its kernel takes any size): integer sums of the k*k flat taps per channel,
// integer division by k*k, alpha 255.  Separable and exact: a row pass of sliding K-sums on
// packed u16 pairs ((R,B) and (G,A) of a pixel; a row sum is at most 15*255), a running column
// sum over the K most recent row sums (at most 225*255 = 57375, still 16 bits), and the division
// as (s * ceil(2^24 / k^2)) >> 24, which is exact for every reachable s (checked at launch).
// high 32 bits of the product of two values below 2^24 (v_mul_hi_u32_u24: the operands are masked
// so that the compiler can prove the range)
MI_DEV u32 mulhi_u24(u32 a, u32 b)
{
    return (u32)(((unsigned long long)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)) >> 32);
}

#ifndef MI_MEAN_WAVES
#define MI_MEAN_WAVES 2
#endif
template <int R, int POL>
__global__ __launch_bounds__(256, (R >= 5 ? MI_MEAN_WAVES : 1)) void mean_sep_kernel(const u32* __restrict__ in,
                                                       u32* __restrict__ out, const Geo g,
                                                       const u32 magic)
{
    constexpr int K = 2 * R + 1;
    constexpr int HOPS = (R + 3) / 4;
    constexpr int N1 = R < 4 ? R : 4, N2 = R > 4 ? R - 4 : 1;   // pixels used of halo groups 1 and 2
    constexpr int NH = 6;
    MI_MARCH_PROLOGUE();

    // input rows in bursts through a per-wave LDS ring (RowFeed, common.h) with the 4-pixel groups beside the strip
    typedef RowFeed<POL, MI_SEP_BURST, 4 * HOPS> Feed;
    typedef typename Feed::Row RowIn;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[Feed::BLOCK_BYTES];
    Feed feed(io, mch, ring, j.lane);
    const bool first_lane = j.lane == 0;
    // h[k] = (R,B) sums over columns k-R .. k+R for the lane's pixel k; h[4], h[5] = the G sums of pixels (0,1), (2,3)
    // (six registers a row: the alpha sums are never used, and the window is what limits the waves per SIMD)
    auto hrow = [&](const RowIn& r, u32 (&h)[NH]) {
        u32 col[20];  // columns -8..11 at index c+8; unused entries are dead code
        const u32 p[4] = {r.p.x, r.p.y, r.p.z, r.p.w};
        // halo group pixels: element i is the group's pixel 4-N+i in lane 0 (left of the strip) and
        // its pixel i in lane 63 (right of it); the shifts only read those two lanes' edge values
        const u32 n4[4] = {r.near.x, r.near.y, r.near.z, r.near.w}, f4[4] = {r.far.x, r.far.y, r.far.z, r.far.w};
        u32 a1[4] = {0u, 0u, 0u, 0u}, a2[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < N1; i++) a1[i] = first_lane ? n4[4 - N1 + i] : n4[i];
        if constexpr (HOPS > 1) {
#pragma unroll
            for (int i = 0; i < N2; i++) a2[i] = first_lane ? f4[4 - N2 + i] : f4[i];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            col[8 + k] = p[k];
            col[4 + k] = from_left(p[k], k >= 4 - N1 ? a1[(k - (4 - N1)) & 3] : 0u);
            col[12 + k] = from_right(p[k], k < N1 ? a1[k] : 0u);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            col[k] = from_left(col[4 + k], (HOPS > 1 && k >= 4 - N2) ? a2[(k - (4 - N2)) & 3] : 0u);
            col[16 + k] = from_right(col[12 + k], (HOPS > 1 && k < N2) ? a2[k] : 0u);
        }
        u32 lo[4 + 2 * R], hi[4 + 2 * R];
#pragma unroll
        for (int i = 0; i < 4 + 2 * R; i++) unpack_px(col[8 - R + i], lo[i], hi[i]);
        u32 sl = lo[0], sh = hi[0], gs[4];
#pragma unroll
        for (int i = 1; i < K; i++) { sl = pk_add(sl, lo[i]); sh = pk_add(sh, hi[i]); }
        h[0] = sl; gs[0] = sh;
#pragma unroll
        for (int k = 1; k < 4; k++) {
            sl = pk_add(pk_sub(sl, lo[k - 1]), lo[k + K - 1]);
            sh = pk_add(pk_sub(sh, hi[k - 1]), hi[k + K - 1]);
            h[k] = sl; gs[k] = sh;
        }
        h[4] = __builtin_amdgcn_perm(gs[1], gs[0], 0x05040100u);
        h[5] = __builtin_amdgcn_perm(gs[3], gs[2], 0x05040100u);
    };

    u32 win[K][NH];
    u32 cs[NH] = {0u, 0u, 0u, 0u, 0u, 0u};
    feed.start(-R);
#pragma unroll
    for (int i = 0; i < K - 1; i++) {
        hrow(feed.next(i, K - 1), win[i]);
#pragma unroll
        for (int e = 0; e < NH; e++) cs[e] = pk_add(cs[e], win[i][e]);
    }
    constexpr int UNR = K;
    for (int base = 0; base < mch.n; base += UNR) {
#pragma unroll
        for (int s = 0; s < UNR; s++) {
            const int i = base + s;
            if (i < mch.n) {
            const int y = mch.row(i);
            // rows y-R+t live in slot (s+t)%K; the newest one, y+R, replaces the row dropped last step
            hrow(feed.next(i + K - 1, K - 1), win[(K - 1 + s) % K]);
#pragma unroll
            for (int e = 0; e < NH; e++) cs[e] = pk_add(cs[e], win[(K - 1 + s) % K][e]);
            const u32 rowm = (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u;
            u32 o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // (s * magic) >> 24 as the high word of the 24-bit product (s << 8) * magic: s < 2^16 and
                // magic < 2^24, and v_mul_hi_u32_u24 runs at full rate where a 32-bit multiply takes four slots
                const u32 qr = mulhi_u24((cs[k] & 0xFFFFu) << 8, magic);
                const u32 qb = mulhi_u24((cs[k] >> 16) << 8, magic);
                const u32 qg = mulhi_u24((k % 2 ? cs[4 + k / 2] >> 16 : cs[4 + k / 2] & 0xFFFFu) << 8, magic);
                o[k] = (qr | (qg << 8) | (qb << 16) | 0xFF000000u) & xm[k] & rowm;
            }
            io.store(y, u32x4{o[0], o[1], o[2], o[3]});
#pragma unroll
            for (int e = 0; e < NH; e++) cs[e] = pk_sub(cs[e], win[s % K][e]);   // drop row y-R
            }
        }
    }
}

// ============================================================================ mean 17x17 .. 31x31
// A window of K = 2R+1 row sums no longer fits the registers (31 rows x 8), so this kernel keeps none:
// the running column sums gain the row that enters the window and lose the row that leaves it, and the
// leaving row is simply READ AGAIN (it was loaded K steps earlier by the same wave: an L2 / Infinity
// Cache hit) and its horizontal sums recomputed.  The horizontal box sum is a difference of prefix
// sums: packed 16-bit (R|B<<16, G|A<<16) inclusive prefix sums across the whole wave -- three adds in
// the lane, a six-step DPP scan of the lane totals -- which may wrap around, because only differences
// P[x+R] - P[x-R-1] <= 31*255 are ever used; those two prefix values come from the lanes about R/4
// away through ds_bpermute.  Four margin lanes on each side feed their neighbours only (56 productive
// lanes, 224-pixel strips), so the same code serves every R <= 15.  Column sums are 32-bit (31*31*255
// needs 18 bits); the division is the exact multiply-high by ceil(2^32/k^2) that the host verified
// for every reachable sum.  Integer arithmetic throughout: bit-exact, any decomposition.
MI_DEV u32 wave_scan_pk16(u32 v)
{   // inclusive prefix sum over the 64 lanes, per 16-bit half (the gfx9 DPP scan: row shifts, then row broadcasts)
    v = pk_add(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = pk_add(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = pk_add(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = pk_add(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = pk_add(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v = pk_add(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return v;
}

template <int R, int POL>
__global__ __launch_bounds__(256) void mean_prefix_kernel(const u32* __restrict__ in, u32* __restrict__ out,
                                                          const Geo g, const u32 magic32)
{
    static_assert(R >= 1 && R <= 15, "four margin lanes cover a reach of 16 pixels");
    constexpr int MRG = 4, OUTW = 4 * (64 - 2 * MRG);
    const WaveJob j = wave_job(g);
    if (!j.valid) return;
    const int cx = j.strip * OUTW - 4 * MRG + 4 * j.lane;
    const March mch = make_march(g, j);
    const RowIO<POL> io(in, out, g, j, mch, j.strip * OUTW - 4 * MRG, MRG);
    const u32 xm[4] = {xmask(cx, g.W), xmask(cx + 1, g.W), xmask(cx + 2, g.W), xmask(cx + 3, g.W)};
    // source lanes of the two prefix values of pixel k: column k+R is pixel (k+R)%4 of lane +(k+R)/4,
    // column k-R-1 is pixel (k-R-1) mod 4 of lane -((R+1-k)+3)/4
    int hi_addr[4], lo_addr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        hi_addr[k] = ((j.lane + (k + R) / 4) & 63) * 4;
        lo_addr[k] = ((j.lane - (R + 1 - k + 3) / 4) & 63) * 4;
    }
    // h[2k] = (R,B), h[2k+1] = (G,A) box sums over columns k-R .. k+R, for the lane's pixel k, of the packed
    // 16-bit values pl / ph (one row's pixels, or the DIFFERENCE of the entering and the leaving row: prefix
    // sums are linear and everything is modulo 2^16, so the halves of the result are then signed)
    auto box = [&](u32 (&pl)[4], u32 (&ph)[4], u32 (&h)[8]) {
#pragma unroll
        for (int k = 1; k < 4; k++) { pl[k] = pk_add(pl[k], pl[k - 1]); ph[k] = pk_add(ph[k], ph[k - 1]); }
        const u32 el = pk_sub(wave_scan_pk16(pl[3]), pl[3]), eh = pk_sub(wave_scan_pk16(ph[3]), ph[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) { pl[k] = pk_add(pl[k], el); ph[k] = pk_add(ph[k], eh); }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int a = (k + R) % 4, b = ((k - R - 1) % 4 + 4) % 4;
            const u32 al = (u32)__builtin_amdgcn_ds_bpermute(hi_addr[k], (int)pl[a]);
            const u32 ah = (u32)__builtin_amdgcn_ds_bpermute(hi_addr[k], (int)ph[a]);
            const u32 bl = (u32)__builtin_amdgcn_ds_bpermute(lo_addr[k], (int)pl[b]);
            const u32 bh = (u32)__builtin_amdgcn_ds_bpermute(lo_addr[k], (int)ph[b]);
            h[2 * k] = pk_sub(al, bl);
            h[2 * k + 1] = pk_sub(ah, bh);
        }
    };
    auto hrow = [&](u32x4 p, u32 (&h)[8]) {
        const u32 px[4] = {p.x, p.y, p.z, p.w};
        u32 pl[4], ph[4];
#pragma unroll
        for (int k = 0; k < 4; k++) unpack_px(px[k], pl[k], ph[k]);
        box(pl, ph, h);
    };
    auto hrow_diff = [&](u32x4 pin, u32x4 pout, u32 (&h)[8]) {
        const u32 pi[4] = {pin.x, pin.y, pin.z, pin.w}, po[4] = {pout.x, pout.y, pout.z, pout.w};
        u32 pl[4], ph[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 il, ih, ol, oh;
            unpack_px(pi[k], il, ih);
            unpack_px(po[k], ol, oh);
            pl[k] = pk_sub(il, ol); ph[k] = pk_sub(ih, oh);
        }
        box(pl, ph, h);
    };
    // column sums: S[3k + c], c = R, G, B of pixel k
    u32 S[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    auto add_row = [&](const u32 (&h)[8]) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            S[3 * k] += h[2 * k] & 0xFFFFu; S[3 * k + 2] += h[2 * k] >> 16; S[3 * k + 1] += h[2 * k + 1] & 0xFFFFu;
        }
    };
    // rows row(-R) .. row(R-1) prime the sums
    {
        u32x4 nxt = io.load(mch.row(-R));
        for (int t = -R; t < R; t++) {
            const u32x4 cur = nxt;
            nxt = io.load(mch.row(t + 1));      // the last one is row(R): step 0's entering row
            u32 h[8];
            hrow(cur, h);
            add_row(h);
        }
        // two steps' rows in flight: register sets A (even steps) and B (odd steps) are refilled right after
        // use for the step after next, so no set is ever copied (the loop is unrolled by two)
        auto step = [&](int i, u32x4& rin, u32x4& rout) {
            const int y = mch.row(i);
            const u32x4 cin = rin, cout = rout;
            rin = io.load(mch.row(i + 2 + R));          // enters at step i + 2
            rout = io.load(mch.row(i + 1 - R));         // leaves at step i + 2
            u32 h[8];
            if (i == 0) {
                hrow(cin, h);
                add_row(h);
            } else {
                hrow_diff(cin, cout, h);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    S[3 * k] += (u32)__builtin_amdgcn_sbfe((int)h[2 * k], 0, 16);
                    S[3 * k + 2] += (u32)((int)h[2 * k] >> 16);
                    S[3 * k + 1] += (u32)__builtin_amdgcn_sbfe((int)h[2 * k + 1], 0, 16);
                }
            }
            const u32 rowm = (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u;
            u32 o[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                o[k] = (__umulhi(S[3 * k], magic32) | (__umulhi(S[3 * k + 1], magic32) << 8) |
                        (__umulhi(S[3 * k + 2], magic32) << 16) | 0xFF000000u) & xm[k] & rowm;
            io.store(y, u32x4{o[0], o[1], o[2], o[3]});
        };
        // step 0 adds its entering row like the priming rows; step i > 0 adds (row(i+R) - row(i-R-1)) in ONE
        // row pass: the box sums of the difference, sign-extended into the 32-bit column sums
        u32x4 in_a = nxt, out_a = nxt;                  // step 0: row(R); no leaving row
        u32x4 in_b = io.load(mch.row(1 + R)), out_b = io.load(mch.row(-R));   // step 1
        for (int i = 0; i < mch.n; i += 2) {
            step(i, in_a, out_a);
            if (i + 1 < mch.n) step(i + 1, in_b, out_b);
        }
    }
}

// ============================================================================ Gaussian, direct (3x3; 5x5 and 7x7: gauss_direct.hip)
// imageGaussianBlur, imageTools.cu:287-311, in the reference's own order -- ky outer, kx inner,
// sum = fmaf(pixel, weight, sum) from 0, truncating store -- so the result is bit-exact against the
// oracle, as a wave march: the K most recent rows stay in registers (own 4 pixels plus R neighbours
// each side), every output row is K*K fused multiply-adds per pixel and channel.
// All bands march down: the accumulation order over ky is part of the result.
struct GaussW2 { float w[49]; };

template <int R, int POL>
__global__ __launch_bounds__(256) void gauss_direct_march_kernel(const u32* __restrict__ in,
                                                                 u32* __restrict__ out, const Geo g,
                                                                 const GaussW2 gw)
{
    constexpr int K = 2 * R + 1, NC = 4 + 2 * R;
    const WaveJob j = wave_job(g);
    if (!j.valid) return;
    const int cx = j.strip * 256 + 4 * j.lane;
    March mch;
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

    // the window keeps packed pixels (K x (4+2R) registers); a row is widened to floats when the
    // accumulation reaches it, so only one row of floats is live at a time
    // 3x3 (the size the reference's demo asks for): rows in bursts through the LDS ring like the other 3x3
    // kernels; the wider ones are bound by their k*k multiply-adds and keep the one-row register prefetch
    typedef RowFeed<POL, (R == 1 ? MI_FEED_K : 0), 1> Feed;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[Feed::BLOCK_BYTES];
    Feed feed(io, mch, ring, j.lane);
    auto fed_row = [&](int c) {
        const typename Feed::Row fr = feed.next(c, K - 1);
        RowIn r;
        r.p = fr.p;
        r.h = u32x4{fr.h, 0u, 0u, 0u};
        return r;
    };
    u32 win[K][NC];
    RowIn nxt{};
    if constexpr (R == 1) {
        feed.start(-R);
#pragma unroll
        for (int i = 0; i < K - 1; i++) columns(fed_row(i), win[i]);
    } else {
#pragma unroll
        for (int i = 0; i < K - 1; i++) columns(load_row(mch.row(i - R)), win[i]);
        nxt = load_row(mch.row(R));
    }
    for (int base = 0; base < mch.n; base += K) {
#pragma unroll
        for (int s = 0; s < K; s++) {
            const int i = base + s;
            if (i < mch.n) {
            const int y = mch.row(i);
            if constexpr (R == 1) {
                columns(fed_row(i + K - 1), win[(K - 1 + s) % K]);
            } else {
                columns(nxt, win[(K - 1 + s) % K]);      // row y+R; row y-R+t lives in slot (s+t)%K
                nxt = load_row(mch.row(i + R + 1));
            }
            const u32 rowm = (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u;
            float acc[4][3] = {};
#pragma unroll
            for (int t = 0; t < K; t++) {            // ky: top row first, as upstream
                float f[NC][3];
#pragma unroll
                for (int q = 0; q < NC; q++) {
                    const u32 v = win[(s + t) % K][q];
                    f[q][0] = (float)(v & 0xffu);
                    f[q][1] = (float)((v >> 8) & 0xffu);
                    f[q][2] = (float)((v >> 16) & 0xffu);
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int u = 0; u < K; u++)      // kx
#pragma unroll
                        for (int c = 0; c < 3; c++)
                            acc[k][c] = __builtin_fmaf(f[k + u][c], gw.w[t * K + u], acc[k][c]);
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

// ============================================================================ fused chain
// gray -> Gaussian 5x5 -> median 3x3 -> Sobel (BASELINE.json config 2) in one pass: 8 bytes of
// HBM traffic per pixel instead of 32.  After the grayscale stage the three colour channels are
// equal, and every later stage treats them alike, so the pipeline carries ONE value per pixel;
// the arithmetic of each stage is the single-filter kernel's own, operation for operation, and
// each stage's output is zeroed outside the reference's write set before the next stage reads
// it, so the result is bit-identical to running the four kernels one after the other.
// The stages have a combined reach of 4 pixels: strips overlap by MRG >= 1 lanes on each side
// (64 - 2 MRG productive lanes) and a band reads 4 extra rows above and below.
// MI_CHAIN_DBG (experiment builds only, tools/make_variant.sh with MI_EXTRA_CXXFLAGS=-DMI_CHAIN_DBG=n): what a
// stage costs, measured by leaving it out -- 1: loads and stores only, 2: no square root, 3: no median,
// 4: no grayscale arithmetic, 5: no Gaussian, 6: arithmetic only (no loads in the loop, no stores).  The results are wrong by construction; 0 is the product.
#ifndef MI_CHAIN_DBG
#define MI_CHAIN_DBG 0
#endif
#ifndef MI_CHAIN_BURST
#define MI_CHAIN_BURST 3  // rows per request burst (ring of 2 * BURST rows per wave in LDS); 0: one row per step, in registers
#endif
#ifndef MI_CHAIN_WAVES
#define MI_CHAIN_WAVES 1  // waves per SIMD the register allocation must allow (1: whatever the code needs)
#endif
#ifndef MI_CHAIN_BPERM
#define MI_CHAIN_BPERM 0
#endif
template <int POL, int MRG>
__global__ __launch_bounds__(256, MI_CHAIN_WAVES) void chain_kernel(const u32* __restrict__ in,
                                                    u32* __restrict__ out, const Geo g,
                                                    const GaussW gw)
{
    typedef RowFeed<POL, MI_CHAIN_BURST, 0> Feed;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[Feed::BLOCK_BYTES];
    const WaveJob j = wave_job(g);
    if (!j.valid) return;
    constexpr int OUTW = 4 * (64 - 2 * MRG);
    const int cx = j.strip * OUTW - 4 * MRG + 4 * j.lane;
    const March mch = make_march(g, j);
    const RowIO<POL> io(in, out, g, j, mch, j.strip * OUTW - 4 * MRG, MRG);
    const u32 xm[4] = {xmask(cx, g.W), xmask(cx + 1, g.W), xmask(cx + 2, g.W), xmask(cx + 3, g.W)};
    const u32 pm01 = (xm[0] & 0xFFFFu) | (xm[1] & 0xFFFF0000u), pm23 = (xm[2] & 0xFFFFu) | (xm[3] & 0xFFFF0000u);
    auto rowmask = [&](int y) { return (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u; };
    // neighbour exchange: DPP wave shifts, or (stage groups named by the bits of MI_CHAIN_BPERM: 1 gray, 2 median,
    // 4 Sobel) the LDS crossbar -- ds_bpermute takes no vector-ALU issue slot; what wraps around into the
    // edge lanes only reaches margin lanes
    const int src_l = ((j.lane + 63) & 63) * 4, src_r = ((j.lane + 1) & 63) * 4;
    auto xl = [&](u32 v, auto tag) {
        if constexpr ((MI_CHAIN_BPERM & decltype(tag)::value) != 0) return (u32)__builtin_amdgcn_ds_bpermute(src_l, (int)v);
        else return from_left(v);
    };
    auto xr = [&](u32 v, auto tag) {
        if constexpr ((MI_CHAIN_BPERM & decltype(tag)::value) != 0) return (u32)__builtin_amdgcn_ds_bpermute(src_r, (int)v);
        else return from_right(v);
    };

    // stage 1+2a: grayscale of 4 pixels (as floats holding integers), then the Gaussian row pass
    // (two pixels per packed fp32 operation; each lane of a packed operation is the scalar one)
    auto gray_hrow = [&](u32x4 p, f32x2 (&h)[2]) {
        float f[8];
        const u32 px[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const u32 a = px[2 * q], b = px[2 * q + 1];
            const f32x2 r = {(float)(a & 0xffu), (float)(b & 0xffu)};
            const f32x2 gg = {(float)((a >> 8) & 0xffu), (float)((b >> 8) & 0xffu)};
            const f32x2 bl = {(float)((a >> 16) & 0xffu), (float)((b >> 16) & 0xffu)};
            const f32x2 t = 0.587f * gg;
            const f32x2 u = __builtin_elementwise_fma((f32x2)(0.299f), r, t);
            f32x2 v = __builtin_elementwise_fma((f32x2)(0.114f), bl, u);
            if constexpr (MI_CHAIN_DBG == 4) v = r;
            f[2 + 2 * q] = __builtin_floorf(v.x);        // == (float)(uchar)v, v >= 0
            f[3 + 2 * q] = __builtin_floorf(v.y);
        }
        f[0] = __builtin_bit_cast(float, xl(__builtin_bit_cast(u32, f[4]), std::integral_constant<int, 1>{}));
        f[1] = __builtin_bit_cast(float, xl(__builtin_bit_cast(u32, f[5]), std::integral_constant<int, 1>{}));
        f[6] = __builtin_bit_cast(float, xr(__builtin_bit_cast(u32, f[2]), std::integral_constant<int, 1>{}));
        f[7] = __builtin_bit_cast(float, xr(__builtin_bit_cast(u32, f[3]), std::integral_constant<int, 1>{}));
#pragma unroll
        for (int q = 0; q < 2; q++) {
            f32x2 sum = gw.w[0] * f32x2{f[2 * q], f[2 * q + 1]};
#pragma unroll
            for (int t = 1; t < 5; t++)
                sum = __builtin_elementwise_fma((f32x2)(gw.w[t]), f32x2{f[2 * q + t], f[2 * q + 1 + t]}, sum);
            if constexpr (MI_CHAIN_DBG == 5) sum = f32x2{f[2 * q + 2], f[2 * q + 3]};
            h[q] = sum;
        }
    };

    // A wave whose strip and band (with the 4-pixel reach of the chain) lie inside the reference's
    // write set needs none of the per-stage zeroing masks: they are all ones there.  Most waves
    // are such interior waves; the branch is wave-uniform and the two bodies are the same code
    // otherwise, so the result does not depend on which one runs.
    const int x_first = j.strip * OUTW - 4 * MRG, y_lo = (mch.dir > 0 ? mch.ys : mch.ys - (mch.n - 1));
    const bool interior = x_first >= 1 && x_first + 255 <= g.W - 2 && y_lo - 4 >= 1 && y_lo + mch.n - 1 + 4 <= g.H - 2;
    auto run = [&](auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    f32x2 win[5][2];
    // Gaussian output rows (the median's window) and median output rows (Sobel's window) as
    // packed f16 pairs (see sobel_mag_h), in rings indexed by step mod 5 like `win`: with the loop unrolled by 5
    // every index is a compile-time constant and no row is ever copied
    u32 G01[5] = {0, 0, 0, 0, 0}, G23[5] = {0, 0, 0, 0, 0}, M01[5] = {0, 0, 0, 0, 0}, M23[5] = {0, 0, 0, 0, 0};
    // How the input rows arrive.  With its arithmetic removed this kernel took the same time as with it:
    // it was bound by the shape of its memory accesses -- one row per step requested by each of thousands of
    // bands.  As a pure copy that shape runs at 4.9 TB/s; requesting K consecutive rows of the strip at one
    // time runs at 5.4 (tools/marchbench.hip, profiles/r02_marchbench.txt).  RowFeed (common.h) does that:
    // bursts of K = MI_CHAIN_BURST rows straight into a per-wave ring in LDS, one row read back per step.
    Feed feed(io, mch, ring, j.lane);
#pragma unroll
    for (int i = 0; i < 4; i++) gray_hrow(io.load(mch.row(i - 4)), win[i]);
    feed.start(0);

    const int steps = mch.n + 4;   // input steps 0 .. n+3; step i completes output row i-4
    for (int base = 0; base < steps; base += 5) {
#pragma unroll
        for (int s = 0; s < 5; s++) {
            const int i = base + s;
            if (i < steps) {
            const u32x4 cur = feed.next(i, 4).p;      // (the first store is step 4's)
            if constexpr (MI_CHAIN_DBG == 1) {
                if (i >= 4) io.store(mch.row(i - 4), cur);
                continue;
            }
            gray_hrow(cur, win[(4 + s) % 5]);
            // ---- Gaussian column pass: row(i-2), zero outside the write set
            const u32 grow = rowmask(mch.row(i - 2));
            float fb[4];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                f32x2 a = gw.w[2] * win[(s + 2) % 5][q];
                a = __builtin_elementwise_fma((f32x2)(gw.w[0]), win[s % 5][q] + win[(s + 4) % 5][q], a);
                a = __builtin_elementwise_fma((f32x2)(gw.w[1]), win[(s + 1) % 5][q] + win[(s + 3) % 5][q], a);
                if constexpr (MI_CHAIN_DBG == 5) a = win[(s + 2) % 5][q];
                fb[2 * q] = __builtin_truncf(a.x);
                fb[2 * q + 1] = __builtin_truncf(a.y);
            }
            u32 gb01 = pk_f16(fb[0], fb[1]), gb23 = pk_f16(fb[2], fb[3]);
            if constexpr (EDGE) { gb01 &= pm01 & grow; gb23 &= pm23 & grow; }
            G01[s] = gb01; G23[s] = gb23;
            // ---- median 3x3: row(i-3) from the Gaussian rows of steps i-2, i-1, i
            u32 l01 = G01[(s + 3) % 5], m01 = G01[(s + 4) % 5], h01 = gb01;
            u32 l23 = G23[(s + 3) % 5], m23 = G23[(s + 4) % 5], h23 = gb23;
            sort2(l01, m01); sort2(m01, h01); sort2(l01, m01);
            sort2(l23, m23); sort2(m23, h23); sort2(l23, m23);
            // (a band's first steps only prime the rings: the median's first needed row is step 2's, Sobel's
            // first row step 4's -- the stages are skipped until then, 5 % of a 30-row band's arithmetic)
            u32 med01 = 0u, med23 = 0u;
            if constexpr (MI_CHAIN_DBG == 3) { med01 = G01[(s + 4) % 5]; med23 = G23[(s + 4) % 5]; } else
            if (i >= 2) {
                constexpr std::integral_constant<int, 2> T2{};
                const u32 lL = xl(l23, T2), mL = xl(m23, T2), hL = xl(h23, T2);
                const u32 lR = xr(l01, T2), mR = xr(m01, T2), hR = xr(h01, T2);
                const u32 la = __builtin_amdgcn_alignbit(l01, lL, 16), lb = __builtin_amdgcn_alignbit(l23, l01, 16), lc = __builtin_amdgcn_alignbit(lR, l23, 16);
                const u32 ma = __builtin_amdgcn_alignbit(m01, mL, 16), mb = __builtin_amdgcn_alignbit(m23, m01, 16), mc = __builtin_amdgcn_alignbit(mR, m23, 16);
                const u32 ha = __builtin_amdgcn_alignbit(h01, hL, 16), hb = __builtin_amdgcn_alignbit(h23, h01, 16), hc = __builtin_amdgcn_alignbit(hR, h23, 16);
                const u32 mrow = rowmask(mch.row(i - 3));
                med01 = med3(pk_max(pk_max(la, l01), lb), med3(ma, m01, mb), pk_min(pk_min(ha, h01), hb));
                med23 = med3(pk_max(pk_max(lb, l23), lc), med3(mb, m23, mc), pk_min(pk_min(hb, h23), hc));
                if constexpr (EDGE) { med01 &= pm01 & mrow; med23 &= pm23 & mrow; }
            }
            M01[s] = med01; M23[s] = med23;
            // ---- Sobel: row(i-4) from the median rows of steps i-2, i-1, i
            if (i >= 4) {
                const u32 mt01 = M01[(s + 3) % 5], mt23 = M23[(s + 3) % 5], mm01 = M01[(s + 4) % 5], mm23 = M23[(s + 4) % 5];
                const u32 V01 = h_fma2(mm01, h_add(mt01, med01)), V23 = h_fma2(mm23, h_add(mt23, med23));
                const u32 D01 = h_sub(med01, mt01), D23 = h_sub(med23, mt23);
                constexpr std::integral_constant<int, 4> T4{};
                const u32 VL = xl(V23, T4), VR = xr(V01, T4);
                const u32 DL = xl(D23, T4), DR = xr(D01, T4);
                const u32 Va = __builtin_amdgcn_alignbit(V01, VL, 16);
                const u32 Vb = __builtin_amdgcn_alignbit(V23, V01, 16);
                const u32 Vc = __builtin_amdgcn_alignbit(VR, V23, 16);
                const u32 Da = __builtin_amdgcn_alignbit(D01, DL, 16);
                const u32 Db = __builtin_amdgcn_alignbit(D23, D01, 16);
                const u32 Dc = __builtin_amdgcn_alignbit(DR, D23, 16);
                const u32 Gx01 = h_sub(Vb, Va), Gx23 = h_sub(Vc, Vb);
                const u32 Gy01 = h_fma2(D01, h_add(Da, Db)), Gy23 = h_fma2(D23, h_add(Db, Dc));
                const u32 P[4] = {__builtin_amdgcn_perm(Gy01, Gx01, 0x05040100u), __builtin_amdgcn_perm(Gy01, Gx01, 0x07060302u),
                                  __builtin_amdgcn_perm(Gy23, Gx23, 0x05040100u), __builtin_amdgcn_perm(Gy23, Gx23, 0x07060302u)};
                const int y = mch.row(i - 4);
                const u32 srow = rowmask(y);
                u32 o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if constexpr (MI_CHAIN_DBG == 2) o[k] = gray_opaque(P[k] & 0xffu); else
                    o[k] = gray_opaque(sobel_mag_h(P[k]));
                    if constexpr (EDGE) o[k] &= xm[k] & srow;
                }
                if (MI_CHAIN_DBG != 6 || g.batch < 0) io.store_in_band(y, u32x4{o[0], o[1], o[2], o[3]});
            }
            }
        }
    }
    };
    if (interior) run(std::false_type{}); else run(std::true_type{});
}

// ============================================================================ generic kernels
// One thread per pixel, taps read one by one through the cache: the fallback for kernel sizes
// without a specialised kernel, and the bit-exact "direct" Gaussian.
struct Weights2D { float w[MI_MAX_KERNEL_SIZE * MI_MAX_KERNEL_SIZE]; };   // 3844 B: fits the 4 KiB of kernel arguments

MI_DEV u32 tap_px(const u32* __restrict__ in, const Geo& g, i64 f)
{
    return (f >= g.in_lo && f < g.in_hi) ? in[f - g.in_origin] : 0u;
}

// imageGaussianBlur, imageTools.cu:287-311, in the reference's own order: ky outer, kx inner,
// sum = fmaf(pixel, weight, sum) from 0, truncating store.  Bit-exact against the oracle.
__global__ __launch_bounds__(256) void gauss_direct_kernel(const u32* __restrict__ in,
                                                           u32* __restrict__ out, const Geo g,
                                                           int k, const Weights2D wt)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = g.y_begin + blockIdx.y;
    const int frame = blockIdx.z;
    if (x >= g.W) return;
    in += (i64)frame * g.in_frame_stride;
    out += (i64)frame * g.out_frame_stride;
    u32 o = 0;
    if (x >= 1 && x <= g.W - 2 && y >= 1 && y <= g.H - 2) {
        const int r = k / 2;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int ky = -r; ky <= r; ky++)
            for (int kx = -r; kx <= r; kx++) {
                const u32 p = tap_px(in, g, (i64)(y + ky) * g.W + (x + kx));
                const float kv = wt.w[(ky + r) * k + (kx + r)];
                sx = __builtin_fmaf((float)(p & 0xffu), kv, sx);
                sy = __builtin_fmaf((float)((p >> 8) & 0xffu), kv, sy);
                sz = __builtin_fmaf((float)((p >> 16) & 0xffu), kv, sz);
            }
        o = (u32)sx | ((u32)sy << 8) | ((u32)sz << 16) | 0xFF000000u;
    }
    out[(i64)y * g.W + x - g.out_origin] = o;
}

// The same for kernel sizes above 31, whose k*k weights no longer fit the kernel arguments: weights in device memory.
__global__ __launch_bounds__(256) void gauss_direct_ptr_kernel(const u32* __restrict__ in, u32* __restrict__ out, const Geo g,
                                                               int k, const float* __restrict__ wt)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = g.y_begin + blockIdx.y;
    const int frame = blockIdx.z;
    if (x >= g.W) return;
    in += (i64)frame * g.in_frame_stride;
    out += (i64)frame * g.out_frame_stride;
    u32 o = 0;
    if (x >= 1 && x <= g.W - 2 && y >= 1 && y <= g.H - 2) {
        const int r = k / 2;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int ky = -r; ky <= r; ky++)
            for (int kx = -r; kx <= r; kx++) {
                const u32 p = tap_px(in, g, (i64)(y + ky) * g.W + (x + kx));
                const float kv = wt[(ky + r) * k + (kx + r)];
                sx = __builtin_fmaf((float)(p & 0xffu), kv, sx);
                sy = __builtin_fmaf((float)((p >> 8) & 0xffu), kv, sy);
                sz = __builtin_fmaf((float)((p >> 16) & 0xffu), kv, sz);
            }
        o = (u32)sx | ((u32)sy << 8) | ((u32)sz << 16) | 0xFF000000u;
    }
    out[(i64)y * g.W + x - g.out_origin] = o;
}

// imageMeanBlur, imageTools.cu:313-340, for any odd k.
__global__ __launch_bounds__(256) void mean_generic_kernel(const u32* __restrict__ in,
                                                           u32* __restrict__ out, const Geo g,
                                                           int k)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = g.y_begin + blockIdx.y;
    const int frame = blockIdx.z;
    if (x >= g.W) return;
    in += (i64)frame * g.in_frame_stride;
    out += (i64)frame * g.out_frame_stride;
    u32 o = 0;
    if (x >= 1 && x <= g.W - 2 && y >= 1 && y <= g.H - 2) {
        const int r = k / 2;
        int mx = 0, my = 0, mz = 0;
        for (int ky = -r; ky <= r; ky++)
            for (int kx = -r; kx <= r; kx++) {
                const u32 p = tap_px(in, g, (i64)(y + ky) * g.W + (x + kx));
                mx += (int)(p & 0xffu); my += (int)((p >> 8) & 0xffu); mz += (int)((p >> 16) & 0xffu);
            }
        const int d = k * k;
        o = (u32)(mx / d) | ((u32)(my / d) << 8) | ((u32)(mz / d) << 16) | 0xFF000000u;
    }
    out[(i64)y * g.W + x - g.out_origin] = o;
}

// imageMeanBlur for kernel sizes 17..31, where the wave march's window no longer fits in registers:
// one block per 64 x 16 output tile.  The tile and its border go to LDS (by flat index, so the
// taps wrap from row to row like the reference's), then a row pass leaves k-sums per channel in
// LDS -- (R,B) share a word, a row sum is at most 31*255 -- and a column pass adds k of those in
// 32 bits; the division is (s * magic) >> 32 with magic = ceil(2^32 / k^2), exact for every
// reachable s (checked on the host).
constexpr int MT_W = 64, MT_H = 16;
__global__ __launch_bounds__(256) void mean_tile_kernel(const u32* __restrict__ in, u32* __restrict__ out,
                                                        const Geo g, int k, u32 magic)
{
    extern __shared__ u32 lds[];
    const int r = k / 2;
    const int PW = MT_W + 2 * r, PH = MT_H + 2 * r;
    u32* pix = lds;                    // [PH][PW]
    u32* sum_rb = pix + PH * PW;       // [PH][MT_W]
    u32* sum_g = sum_rb + PH * MT_W;   // [PH][MT_W]
    const int x0 = blockIdx.x * MT_W, y0 = g.y_begin + blockIdx.y * MT_H, frame = blockIdx.z;
    in += (i64)frame * g.in_frame_stride;
    out += (i64)frame * g.out_frame_stride;
    const int tid = threadIdx.x;

    for (int idx = tid; idx < PH * PW; idx += 256) {
        const int ty = idx / PW, tx = idx - ty * PW;
        pix[idx] = tap_px(in, g, (i64)(y0 - r + ty) * g.W + (x0 - r + tx));
    }
    __syncthreads();
    // row pass: a task is 8 consecutive outputs of one tile row, the first a full k-sum, the others
    // by sliding (one tap in, one out)
    for (int task = tid; task < PH * (MT_W / 8); task += 256) {
        const int ty = task / (MT_W / 8), ox = (task % (MT_W / 8)) * 8;
        const u32* row = pix + ty * PW + ox;
        u32 rb = 0, gg = 0;
        for (int t = 0; t < k; t++) {
            const u32 p = row[t];
            rb += p & 0x00ff00ffu;
            gg += (p >> 8) & 0xffu;
        }
        sum_rb[ty * MT_W + ox] = rb;
        sum_g[ty * MT_W + ox] = gg;
#pragma unroll
        for (int e = 1; e < 8; e++) {
            const u32 a = row[k + e - 1], d = row[e - 1];
            rb += (a & 0x00ff00ffu) - (d & 0x00ff00ffu);   // each half stays within [0, 31*255]: no borrow crosses
            gg += ((a >> 8) & 0xffu) - ((d >> 8) & 0xffu);
            sum_rb[ty * MT_W + ox + e] = rb;
            sum_g[ty * MT_W + ox + e] = gg;
        }
    }
    __syncthreads();
    // column pass: a lane owns 4 vertically adjacent outputs of one column, again one full sum and three slides
    {
        const int ox = tid % MT_W, oy0 = (tid / MT_W) * (MT_H / 4);
        const int x = x0 + ox;
        u32 ar = 0, ab = 0, ag = 0;
        for (int t = 0; t < k - 1; t++) {
            const u32 rb = sum_rb[(oy0 + t) * MT_W + ox];
            ar += rb & 0xffffu;
            ab += rb >> 16;
            ag += sum_g[(oy0 + t) * MT_W + ox];
        }
#pragma unroll
        for (int e = 0; e < MT_H / 4; e++) {
            const int oy = oy0 + e, y = y0 + oy;
            const u32 rb = sum_rb[(oy + k - 1) * MT_W + ox];
            ar += rb & 0xffffu;
            ab += rb >> 16;
            ag += sum_g[(oy + k - 1) * MT_W + ox];
            if (x < g.W && y < g.y_end) {
                u32 o = 0;
                if (x >= 1 && x <= g.W - 2 && y >= 1 && y <= g.H - 2)
                    o = __umulhi(ar, magic) | (__umulhi(ag, magic) << 8) | (__umulhi(ab, magic) << 16) | 0xFF000000u;
                out[(i64)y * g.W + x - g.out_origin] = o;
            }
            const u32 old = sum_rb[oy * MT_W + ox];
            ar -= old & 0xffffu;
            ab -= old >> 16;
            ag -= sum_g[oy * MT_W + ox];
        }
    }
}

// The direct (reference-order, bit-exact) Gaussian for kernel sizes 9..31: too many taps for the
// wave march's registers, so one block per 64 x 16 output tile, with the tile and its border in
// LDS like mean_tile_kernel.  A lane owns 4 horizontally adjacent outputs: every pixel of a window
// row is read and widened once and feeds up to four accumulations, each of which still runs
// ky outer, kx inner, sum = fmaf(pixel, weight, sum) from 0.
template <int K>
__global__ __launch_bounds__(256) void gauss_direct_tile_kernel(const u32* __restrict__ in, u32* __restrict__ out,
                                                                const Geo g, const Weights2D wt)
{
    constexpr int R = K / 2, PW = MT_W + 2 * R, PH = MT_H + 2 * R;
    __shared__ u32 pix[PH * PW];
    __shared__ float wl[K * K];
    const int x0 = blockIdx.x * MT_W, y0 = g.y_begin + blockIdx.y * MT_H, frame = blockIdx.z;
    in += (i64)frame * g.in_frame_stride;
    out += (i64)frame * g.out_frame_stride;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < PH * PW; idx += 256) {
        const int ty = idx / PW, tx = idx - ty * PW;
        pix[idx] = tap_px(in, g, (i64)(y0 - R + ty) * g.W + (x0 - R + tx));
    }
    for (int idx = tid; idx < K * K; idx += 256) wl[idx] = wt.w[idx];
    __syncthreads();
    const int gx = tid % (MT_W / 4), oy = tid / (MT_W / 4);
    float acc[4][3] = {};
#pragma unroll 1   // unrolled, all K*K weights become live at once and occupancy collapses
    for (int ky = 0; ky < K; ky++) {
        float w[K];
#pragma unroll
        for (int kx = 0; kx < K; kx++) w[kx] = wl[ky * K + kx];
        const u32* row = pix + (oy + ky) * PW + 4 * gx;
#pragma unroll
        for (int t = 0; t < K + 3; t++) {
            const u32 p = row[t];
            const float f[3] = {(float)(p & 0xffu), (float)((p >> 8) & 0xffu), (float)((p >> 16) & 0xffu)};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int kx = t - j;        // pixel t of the row is tap kx of output j
                if (kx >= 0 && kx < K) {
#pragma unroll
                    for (int c = 0; c < 3; c++) acc[j][c] = __builtin_fmaf(f[c], w[kx], acc[j][c]);
                }
            }
        }
    }
    const int y = y0 + oy;
    if (y >= g.y_end) return;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + 4 * gx + j;
        if (x >= g.W) continue;
        u32 o = 0;
        if (x >= 1 && x <= g.W - 2 && y >= 1 && y <= g.H - 2)
            o = (u32)acc[j][0] | ((u32)acc[j][1] << 8) | ((u32)acc[j][2] << 16) | 0xFF000000u;
        out[(i64)y * g.W + x - g.out_origin] = o;
    }
}

// ============================================================================ median 5x5
// 13th smallest of the 25 flat taps per channel, alpha 255, same write set as the other stencils.
// Wave march with one margin lane on each side (62 productive lanes, like the fused chain).
// Work shared between neighbouring outputs (networks in median5_network.h):
//   * every 5-pixel column is sorted once per row, and the sort itself is shared between vertically
//     adjacent rows (sorted pairs -> sorted quads -> one insertion per window, see the kernel body);
//   * adjacent sorted columns are merged pairwise (merge55) -- a lane merges its columns (0,1)
//     and (2,3), the pair to its right arrives from the neighbouring lane;
//   * two outputs that share four columns share ranks 7..12 of those 20 values (ranks7to12):
//     outputs 1,2 use the lane's own two pairs, output 3 and the right neighbour's output 0 use
//     pair (2,3) and the neighbour's pair (0,1);
//   * each output then picks the 6th smallest of those 6 values and its fifth column (select6of11).
// R and B of a pixel share a register (one column per register).  The G bytes of pixels (0, 2) and
// of pixels (1, 3) share one each, so the two G output registers -- outputs (0, 2) and (1, 3) -- need
// the packed columns X-2 .. X2 and X-1 .. X3 (X_j = columns j and j + 2): they share four of them
// like two adjacent R,B outputs, and the composite columns come from the neighbours by alignbit.
// Neighbour exchange: the R,B values cross lanes through the LDS crossbar (ds_bpermute: no LDS
// memory, no vector-ALU issue slot -- this kernel is bound by vector issue), the G ones by DPP
// (measured: all DPP 0.413 ms, all ds_bpermute 0.405, this mix 0.399 per 16 x 4K).
#ifndef MI_MED5_BPERM
#define MI_MED5_BPERM 1
#endif
#ifndef MI_MED5_MIX       // 0: every exchange as MI_MED5_BPERM says; 1: G by DPP; 2: R,B by DPP
#define MI_MED5_MIX 1
#endif
#ifndef MI_MED5_WAVES
#define MI_MED5_WAVES 1
#endif
#if MI_MED5_MIX == 1
#define GL dl
#define GR dr
#define RL fl
#define RR fr
#elif MI_MED5_MIX == 2
#define GL fl
#define GR fr
#define RL dl
#define RR dr
#else
#define GL fl
#define GR fr
#define RL fl
#define RR fr
#endif
template <int POL>
__global__ __launch_bounds__(256, MI_MED5_WAVES) void median5_kernel(const u32* __restrict__ in,
                                                      u32* __restrict__ out, const Geo g)
{
    const WaveJob j = wave_job(g);
    if (!j.valid) return;
    constexpr int OUTW = 4 * 62;
    const int cx = j.strip * OUTW - 4 + 4 * j.lane;
    const March mch = make_march(g, j);
    const RowIO<POL> io(in, out, g, j, mch, j.strip * OUTW - 4, 1);
    const u32 xm[4] = {xmask(cx, g.W), xmask(cx + 1, g.W), xmask(cx + 2, g.W), xmask(cx + 3, g.W)};
    constexpr u32 SEL_G = 0x0c050c01u;   // (byte 1 of src1) | (byte 1 of src0) << 16
#if MI_MED5_BPERM
    // neighbour exchange through the LDS crossbar (ds_bpermute: no LDS memory, no VALU issue slot)
    // instead of DPP moves; the wrap-around lanes are margin lanes whose outputs are not stored
    const int src_l = ((j.lane + 63) & 63) * 4, src_r = ((j.lane + 1) & 63) * 4;
    auto fl = [&](u32 v) { return (u32)__builtin_amdgcn_ds_bpermute(src_l, (int)v); };
    auto fr = [&](u32 v) { return (u32)__builtin_amdgcn_ds_bpermute(src_r, (int)v); };
#else
    auto fl = [](u32 v) { return from_left(v); };
    auto fr = [](u32 v) { return from_right(v); };
#endif
    auto dl = [](u32 v) { return from_left(v); };
    auto dr = [](u32 v) { return from_right(v); };
    auto mn = [](u32 a, u32 b) { return pk_min(a, b); };
    auto mx = [](u32 a, u32 b) { return pk_max(a, b); };

    // slots 0..3 = (R | B << 16) of the lane's pixels, 4 = G of pixels 0,2, 5 = G of pixels 1,3
    auto unpack = [&](u32x4 p, u32 (&w)[6]) {
        w[0] = p.x & 0x00FF00FFu; w[1] = p.y & 0x00FF00FFu; w[2] = p.z & 0x00FF00FFu; w[3] = p.w & 0x00FF00FFu;
        w[4] = __builtin_amdgcn_perm(p.z, p.x, SEL_G);
        w[5] = __builtin_amdgcn_perm(p.w, p.y, SEL_G);
    };

    // Column sorts shared between vertically adjacent outputs.  Rows arrive one per step (arrival a =
    // step + 4); consecutive arrivals (2m, 2m+1) are sorted once into a pair P_m, consecutive pairs are
    // merged once into a quad Q_(m-1) (arrivals 2m-2 .. 2m+1), and the two windows that contain that quad
    // -- arrivals 2m-3 .. 2m+1 and 2m-2 .. 2m+2 -- each insert their fifth row into it: 12 min/max per
    // column and step where sorting five rows from scratch takes 18.  The loop is unrolled by four steps so
    // that the pair ring, the ring of raw odd arrivals and every other index are compile-time constants.
    auto sort2v = [&](u32 a, u32 b, u32 (&o)[2]) { o[0] = mn(a, b); o[1] = mx(a, b); };
    auto merge22 = [&](const u32 (&p)[2], const u32 (&q)[2], u32 (&o)[4]) {
        const u32 t1 = mx(p[0], q[0]), t2 = mn(p[1], q[1]);
        o[0] = mn(p[0], q[0]); o[1] = mn(t1, t2); o[2] = mx(t1, t2); o[3] = mx(p[1], q[1]);
    };
    auto insert5 = [&](u32 x, const u32 (&a)[4], u32 (&o)[5]) {
        o[0] = mn(a[0], x);
        o[1] = mx(a[0], mn(a[1], x));
        o[2] = mx(a[1], mn(a[2], x));
        o[3] = mx(a[2], mn(a[3], x));
        o[4] = mx(a[3], x);
    };
    u32 odd_raw[2][6], even_raw[6], P[2][6][2], Q[6][4];
    {
        u32 r0[6], r1[6], r2[6], r3[6];
        unpack(io.load(mch.row(-2)), r0); unpack(io.load(mch.row(-1)), r1);
        unpack(io.load(mch.row(0)), r2);  unpack(io.load(mch.row(1)), r3);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            sort2v(r0[k], r1[k], P[0][k]); sort2v(r2[k], r3[k], P[1][k]);
            merge22(P[0][k], P[1][k], Q[k]);
            odd_raw[0][k] = r1[k]; odd_raw[1][k] = r3[k]; even_raw[k] = 0u;
        }
    }
    u32x4 nxt = io.load(mch.row(2));
    for (int base = 0; base < mch.n; base += 4) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int i = base + s;
            if (i < mch.n) {
            const int y = mch.row(i);
            u32 cur[6];
            unpack(nxt, cur);
            nxt = io.load(mch.row(i + 3));
            u32 c[6][5];   // sorted columns
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (s % 2 == 0) {          // even arrival 2m+2: the quad of the step before plus this row
                    insert5(cur[k], Q[k], c[k]);
                    even_raw[k] = cur[k];
                } else {                   // odd arrival 2m+1, m = 2 + i/2 (its parity is s/2 for base % 4 == 0)
                    const int mp = s / 2;
                    sort2v(even_raw[k], cur[k], P[mp][k]);
                    merge22(P[mp ^ 1][k], P[mp][k], Q[k]);
                    insert5(odd_raw[mp][k], Q[k], c[k]);
                    odd_raw[mp][k] = cur[k];
                }
            }
            u32 rb[4], g02, g13;
            {   // ---- R,B
                u32 m01[10], m23[10], rm01[10], sa[6], sb[6], lsb[6], lc3[5], rc0[5];
                merge55(c[0], c[1], m01, mn, mx);
                merge55(c[2], c[3], m23, mn, mx);
#pragma unroll
                for (int k = 0; k < 10; k++) rm01[k] = RR(m01[k]);
                ranks7to12(m01, m23, sa, mn, mx);      // columns 0..3
                ranks7to12(m23, rm01, sb, mn, mx);     // columns 2..5
#pragma unroll
                for (int k = 0; k < 6; k++) lsb[k] = RL(sb[k]);   // columns -2..1
#pragma unroll
                for (int k = 0; k < 5; k++) { lc3[k] = RL(c[3][k]); rc0[k] = RR(c[0][k]); }
                rb[0] = select6of11(lsb, c[2], mn, mx);   // window -2..2
                rb[1] = select6of11(sa, lc3, mn, mx);     // window -1..3
                rb[2] = select6of11(sa, rc0, mn, mx);     // window  0..4
                rb[3] = select6of11(sb, c[1], mn, mx);    // window  1..5
            }
            {   // ---- G: packed columns X_j = (column j, column j + 2); own X0 = c[4], X1 = c[5].  Outputs (0, 2)
                // take X-2 .. X2 and outputs (1, 3) take X-1 .. X3: the two registers share X-1 .. X2 exactly as
                // two adjacent R,B outputs share four columns -- two merges, ONE ranks7to12, two selections
                u32 xm2[5], xm1[5], x2[5], x3[5], ma[10], mb[10], sg[6];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    xm2[k] = __builtin_amdgcn_alignbit(c[4][k], GL(c[4][k]), 16);   // (left lane's G2, G0)
                    xm1[k] = __builtin_amdgcn_alignbit(c[5][k], GL(c[5][k]), 16);   // (left lane's G3, G1)
                    x2[k] = __builtin_amdgcn_alignbit(GR(c[4][k]), c[4][k], 16);    // (G2, right lane's G0)
                    x3[k] = __builtin_amdgcn_alignbit(GR(c[5][k]), c[5][k], 16);    // (G3, right lane's G1)
                }
                merge55(xm1, c[4], ma, mn, mx);
                merge55(c[5], x2, mb, mn, mx);
                ranks7to12(ma, mb, sg, mn, mx);
                g02 = select6of11(sg, xm2, mn, mx);       // outputs 0, 2
                g13 = select6of11(sg, x3, mn, mx);        // outputs 1, 3
            }
            const u32 rowm = (y >= 1 && y <= g.H - 2) ? 0xFFFFFFFFu : 0u;
            u32x4 o;
            o.x = __builtin_amdgcn_perm(g02, rb[0], 0x0d020400u) & xm[0] & rowm;
            o.y = __builtin_amdgcn_perm(g13, rb[1], 0x0d020400u) & xm[1] & rowm;
            o.z = __builtin_amdgcn_perm(g02, rb[2], 0x0d020600u) & xm[2] & rowm;
            o.w = __builtin_amdgcn_perm(g13, rb[3], 0x0d020600u) & xm[3] & rowm;
            io.store(y, o);
            }
        }
    }
}

// ============================================================================ utility kernels
// One 16-byte non-temporal load + store per lane, the grid covers the buffer once (the streaming
// shape that measured fastest, see gray_kernel).
__global__ __launch_bounds__(256) void copy16_kernel(const u32x4* __restrict__ src,
                                                     u32x4* __restrict__ dst, i64 n16)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

__global__ __launch_bounds__(256) void copy1_kernel(const unsigned char* __restrict__ src,
                                                    unsigned char* __restrict__ dst, i64 n)
{
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void synth_kernel(u32* __restrict__ out, uint64_t seed,
                                                    uint64_t first, uint64_t count)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint64_t z = seed * 0x9E3779B97F4A7C15ULL + (first + i) + 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        out[i] = (u32)z | 0xFF000000u;
    }
}

// ============================================================================ launcher
// Extra dynamic LDS per block: it holds nothing, it caps the blocks a CU runs at one time.  With their
// input rows arriving in bursts the march kernels run fastest with about three waves per SIMD
// (tools/marchbench.hip: fewer concurrent fronts, each denser); the request is within the 64 KiB a
// kernel may ask for without an attribute.
constexpr i64 LDS_PAD_MIN_WAVES = 12288;   // a smaller launch wants every wave slot it can get (measured on 1 and 4 4K frames)
static size_t lds_pad(int r, i64 total_waves)
{
    if (total_waves < LDS_PAD_MIN_WAVES) return 0;
    // measured per 16 x 4K: 3x3 kernels 0.192-0.196 -> 0.187-0.190 ms with 24 KiB (three blocks per CU) and
    // 12-row bands; Gaussian and mean 5x5 0.199 -> 0.197 / 0.194 with 16, mean 7x7 0.200 -> 0.194 with 24;
    // the wider kernels are capped by their registers already
    int kib = r == 1 ? tuning().lds_pad_kib : tuning().sep_lds_pad_kib;
    if (r > 1 && kib < 0) kib = r == 2 ? 16 : r <= 4 ? 24 : 0;
    // (static ring + padding stay within 64 KiB: 20 KiB for the 3x3 kernels, 30 KiB for the separable ones)
    return (size_t)std::max(0, std::min(kib, r == 1 ? 40 : 32)) << 10;
}
static int grid_cap() { return 256 * 8; }  // 256 CUs x 8 blocks of 256 threads

int launch_copy(const void* src, void* dst, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return MI_OK;
    if ((((uintptr_t)src | (uintptr_t)dst | bytes) & 15) == 0) {
        // at most 2^31-1 blocks of 4 KiB per launch
        const i64 per_launch = (i64)0x7fffffff * 256;
        for (i64 done = 0, n16 = (i64)(bytes >> 4); done < n16; done += per_launch) {
            const i64 n = std::min(per_launch, n16 - done);
            copy16_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>((const u32x4*)src + done, (u32x4*)dst + done, n);
        }
    } else {
        const int grid = (int)std::min<i64>(((i64)bytes + 255) / 256, grid_cap());
        copy1_kernel<<<grid, 256, 0, s>>>((const unsigned char*)src, (unsigned char*)dst, (i64)bytes);
    }
    return hip_status(hipGetLastError(), "copy kernel launch");
}

int launch_synth(void* out, uint64_t seed, uint64_t first, uint64_t count, hipStream_t s)
{
    if (count == 0) return MI_OK;
    const int grid = (int)std::min<uint64_t>((count + 255) / 256, (uint64_t)grid_cap() * 4);
    synth_kernel<<<grid, 256, 0, s>>>((u32*)out, seed, first, count);
    return hip_status(hipGetLastError(), "synth kernel launch");
}

int launch_sobel_table(void* out, uint32_t count, hipStream_t s)
{
    if (count == 0) return MI_OK;
    sobel_table_kernel<<<(count + 255) / 256, 256, 0, s>>>((unsigned char*)out, count);
    return hip_status(hipGetLastError(), "sobel table kernel launch");
}

int launch_sobel_pairs(void* out, int gmax, int variant, hipStream_t s)
{
    const int n = 2 * gmax + 1;
    sobel_pairs_kernel<<<(n * n + 255) / 256, 256, 0, s>>>((unsigned char*)out, gmax, variant);
    return hip_status(hipGetLastError(), "sobel pairs kernel launch");
}

int filter_halo_rows(int filter, int ksize, int w)
{
    switch (filter) {
    case MI_FILTER_GRAYSCALE: return 0;
    case MI_FILTER_SOBEL: return 1;
    case MI_FILTER_GAUSSIAN:
    case MI_FILTER_GAUSSIAN_DIRECT:
    case MI_FILTER_MEAN:
    case MI_FILTER_MEDIAN: {
        const int r = ksize / 2;
        if (r <= 1) return 1;
        // the taps are flat indices: r pixels to the side reach ceil(r / w) rows further (one row unless the
        // image is narrower than the radius)
        return r + (w >= r ? 1 : (r + w - 1) / w);
    }
    default: return -1;
    }
}

// Launch KERNEL<..., POL>: POL 1 for 16-byte-aligned images whose width is a multiple of 4
// (aligned dwordx4 accesses), POL 0 otherwise.
#define MI_POLICY_CASES(pol, LAUNCH) \
    if (pol) { LAUNCH(1); } else { LAUNCH(0); }

template <int R>
static void launch_gauss_sep(int pol, int grid, hipStream_t s, const u32* in, u32* out,
                             const Geo& g, const GaussW& gw)
{
#define MI_L(P) gauss_sep_kernel<R, P><<<grid, 256, lds_pad(R, g.total_waves), s>>>(in, out, g, gw)
    MI_POLICY_CASES(pol, MI_L)
#undef MI_L
}

template <int R>
static void launch_gauss_wide(int pol, int grid, hipStream_t s, const u32* in, u32* out, const Geo& g, const GaussWide& gw)
{
#define MI_L(P) gauss_wide_kernel<R, P><<<grid, 256, 0, s>>>(in, out, g, gw)
    MI_POLICY_CASES(pol, MI_L)
#undef MI_L
}

// Weights of gauss_wide_kernel: 0.16 fixed point whose sum is exactly 2^16, paired for an even first tap
// (w0,w1) .. (w2R,0) and an odd one (0,w0) (w1,w2) ..; false when a weight
// does not fit 16 bits (a sigma so small that the centre weight rounds to 1: the fp32 kernel takes those).
static bool gauss_wide_weights(int k, float sigma, GaussWide* gw)
{
    float w[MI_MAX_KERNEL_SIZE];
    gaussian_kernel_1d(k, sigma, w);
    const int R = k / 2;
    long wc[MI_MAX_KERNEL_SIZE], sum = 0;
    for (int t = 0; t < k; t++) { wc[t] = lrint((double)w[t] * 65536.0); sum += wc[t]; }
    wc[R] += 65536 - sum;
    for (int t = 0; t < k; t++) if (wc[t] < 0 || wc[t] > 65535) return false;
    for (int j = 0; j < 8; j++) {
        auto at = [&](int t) -> u32 { return (t >= 0 && t < k) ? (u32)wc[t] : 0u; };
        gw->wc_even[j] = at(2 * j) | (at(2 * j + 1) << 16);
        gw->wc_odd[j] = at(2 * j - 1) | (at(2 * j) << 16);
    }
    return true;
}

// gauss_mfma.hip: separable Gaussians 7x7..15x15 on the matrix cores (aligned images); 1 = not applicable
int launch_gauss_mfma(int k, float sigma, hipStream_t s, const u32* in, u32* out, Geo g);
int launch_mean_mfma(int k, hipStream_t s, const u32* in, u32* out, Geo g);

// gauss_direct.hip: the 5x5 and 7x7 direct Gaussians (weights: the k*k table, row-major)
int launch_gauss_direct_win(int k, int pol, int grid, hipStream_t s, const u32* in, u32* out, const Geo& g, const float* weights);

template <int R>
static void launch_gauss_direct(int pol, int grid, hipStream_t s, const u32* in, u32* out, const Geo& g, const GaussW2& gw)
{
#define MI_L(P) gauss_direct_march_kernel<R, P><<<grid, 256, (R == 1 ? lds_pad(1, g.total_waves) : 0), s>>>(in, out, g, gw)
    MI_POLICY_CASES(pol, MI_L)
#undef MI_L
}

// magic with (s * magic) >> 24 == s / k^2 for every s <= 255 k^2 and s * magic < 2^32, verified
// exhaustively once per process (0 if there is none).
static u32 mean_magic(int k)
{
    static const std::array<u32, 16> table = [] {
        std::array<u32, 16> t{};
        for (int ks = 5; ks <= 15; ks += 2) {
            const u32 d = (u32)(ks * ks), m = ((1u << 24) + d - 1) / d;
            bool ok = true;
            for (u32 sum = 0; sum <= 255u * d && ok; sum++)
                ok = sum < (1u << 16) && m < (1u << 24) && (u32)(((uint64_t)sum * m) >> 24) == sum / d;
            t[ks] = ok ? m : 0u;
        }
        return t;
    }();
    return (k >= 0 && k < 16) ? table[k] : 0u;
}

// ceil(2^32 / k^2) for k = 17..31 when (s * m) >> 32 == s / k^2 for every s <= 255 k^2, else 0
static u32 mean_magic32(int k)
{
    auto magic = [](int ks) {
        const uint64_t d = (uint64_t)ks * ks, m = ((1ull << 32) + d - 1) / d;
        bool ok = m < (1ull << 32);
        for (uint64_t sum = 0; sum <= 255u * d && ok; sum++) ok = ((sum * m) >> 32) == sum / d;
        return ok ? (u32)m : 0u;
    };
    static const std::array<u32, 32> table = [&] {
        std::array<u32, 32> t{};
        for (int ks = 17; ks <= 31; ks += 2) t[ks] = magic(ks);
        return t;
    }();
    if (k >= 0 && k < 32) return table[k];
    // larger sizes (the LDS-tile kernel): verified on first use, then remembered
    static std::mutex mu;
    static std::array<u32, MI_MAX_KERNEL_SIZE_ANY + 1> big{};
    static std::array<bool, MI_MAX_KERNEL_SIZE_ANY + 1> known{};
    if (k > MI_MAX_KERNEL_SIZE_ANY) return 0u;
    std::lock_guard<std::mutex> lock(mu);
    if (!known[k]) { big[k] = magic(k); known[k] = true; }
    return big[k];
}

template <int R>
static void launch_mean_prefix(int pol, int grid, hipStream_t s, const u32* in, u32* out, const Geo& g, u32 magic32)
{
#define MI_L(P) mean_prefix_kernel<R, P><<<grid, 256, 0, s>>>(in, out, g, magic32)
    MI_POLICY_CASES(pol, MI_L)
#undef MI_L
}

template <int R>
static void launch_mean_sep(int pol, int grid, hipStream_t s, const u32* in, u32* out, const Geo& g, u32 magic)
{
#define MI_L(P) mean_sep_kernel<R, P><<<grid, 256, lds_pad(R, g.total_waves), s>>>(in, out, g, magic)
    MI_POLICY_CASES(pol, MI_L)
#undef MI_L
}

// RowIO<1> addresses a band through 32-bit byte offsets from a descriptor that starts up to 16 rows
// above the band; offsets at or above 2^31 double as the "do not load / do not store" sentinel.  A band
// of rpb rows with every row the kernels touch around it (lead rows, halo and wrap rows, prefetched
// rows: fewer than 48) must stay below that, or the launch takes the 64-bit-address policy (POL 0).
static bool band_fits_buffer_offsets(int rpb, int W)
{
    return ((i64)rpb + 16 + 48) * W * 4 < 0x7FFFFF00LL;
}

int launch_chain(const void* d_in, void* d_out, int W, int H, int batch, float sigma, hipStream_t stream)
{
    const u32* in = (const u32*)d_in;
    u32* out = (u32*)d_out;
    bool aligned = (W % 4 == 0) && ((((uintptr_t)in | (uintptr_t)out) & 15) == 0);
    Geo g;
    g.W = W; g.H = H; g.batch = batch;
    g.y_begin = 0; g.y_end = H;
    g.N = (i64)W * H;
    g.in_origin = 0; g.in_lo = 0; g.in_hi = g.N; g.out_origin = 0;
    g.in_frame_stride = g.N; g.out_frame_stride = g.N;
    // margin lanes per side: one is enough for the chain's 4-pixel reach; two make the strips 240
    // pixels wide, whose rows start on 64-byte boundaries (+3..6 % measured), and are taken when that
    // costs at most one strip in 31 (3840 = 16 x 240, 1920 = 8 x 240; 7680 = 32 x 240 against 31 strips
    // of 248: 0.213 ms against 0.2275 for 4 frames)
    int mrg = tuning().chain_margin;
    if (mrg != 1 && mrg != 2) mrg = (i64)((W + 239) / 240) * 31 <= (i64)((W + 247) / 248) * 32 ? 2 : 1;
    const int outw = 4 * (64 - 2 * mrg);
    g.strips = (W + outw - 1) / outw;
    // tall bands amortise the 8 extra rows a band runs through the early stages, short ones keep the
    // rows that are requested at one time close together (measured: best at 32..34 rows, 2 % slower at
    // 30 and 36, 4 % at 42); shrink them only to fill the machine
    int rpb = tuning().chain_rows_per_band;
    if (rpb <= 0) {
        rpb = W > 5760 ? 32 : 34;          // 7680-wide rows: 0.206 ms at 32 (and 30), 0.215 at 34, for 4 frames
        // (small batches: measured on 1, 2 and 4 4K frames and on single 1080p and 675x900 frames)
        while (rpb > 8 && (i64)batch * g.strips * ((H + rpb - 1) / rpb) < 4096) rpb >>= 1;
    }
    rpb = std::max(1, std::min(rpb, (int)((1LL << 28) / W)));
    g.rows_per_band = rpb;
    g.bands = (H + rpb - 1) / rpb;
    const i64 total = (i64)batch * g.bands * g.strips;
    if (total > 0x7fffffff / 64) return fail(MI_ERR_UNSUPPORTED, "launch too large (%lld waves)", (long long)total);
    g.total_waves = (int)total;
    g.xcd_remap = 1;
    GaussW gw;
    gaussian_kernel_1d(5, sigma, gw.w);
    const int grid = (int)((total + 3) / 4);
    aligned = aligned && band_fits_buffer_offsets(rpb, W);
    const size_t pad = total < LDS_PAD_MIN_WAVES ? 0 : (size_t)std::max(0, std::min(tuning().chain_lds_pad_kib, 32)) << 10;
    if (!aligned) chain_kernel<0, 1><<<grid, 256, 0, stream>>>(in, out, g, gw);
    else if (mrg == 2) chain_kernel<1, 2><<<grid, 256, pad, stream>>>(in, out, g, gw);
    else chain_kernel<1, 1><<<grid, 256, pad, stream>>>(in, out, g, gw);
    return hip_status(hipGetLastError(), "chain kernel launch");
}

int launch_filter(const FilterArgs& a)
{
    const int W = a.w, H = a.h_total;
    const u32* in = (const u32*)a.in;
    u32* out = (u32*)a.out;
    const i64 in_frame = (i64)a.in_rows * W, out_frame = (i64)(a.y_end - a.y_begin) * W;
    const bool aligned = (W % 4 == 0) && ((((uintptr_t)in | (uintptr_t)out) & 15) == 0);

    if (a.filter == MI_FILTER_GRAYSCALE) {
        // Point op: a flat pass over the output rows.  When the input holds exactly the output
        // rows, all frames are one contiguous run of pixels and take a single launch.
        const bool contiguous = (a.in_row0 == a.y_begin && in_frame == out_frame);
        const i64 skip = (i64)(a.y_begin - a.in_row0) * W;
        const int launches = contiguous ? 1 : a.batch;
        for (int f = 0; f < launches; f++) {
            const i64 n = contiguous ? out_frame * a.batch : out_frame;
            const u32* src = in + (i64)f * in_frame + skip;
            u32* dst = out + (i64)f * out_frame;
            const bool al = ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
            const i64 groups = (n + 3) / 4;
            const i64 blocks = (groups + 255) / 256;   // one 16-byte group per lane
            if (blocks > 0x7fffffff) return fail(MI_ERR_UNSUPPORTED, "launch too large");
            if (al) gray_kernel<3, 1><<<(int)blocks, 256, 0, a.stream>>>(src, dst, n);
            else gray_kernel<0, 1><<<(int)blocks, 256, 0, a.stream>>>(src, dst, n);
        }
        return hip_status(hipGetLastError(), "grayscale kernel launch");
    }

    Geo g;
    g.W = W; g.H = H; g.batch = a.batch;
    g.y_begin = a.y_begin; g.y_end = a.y_end;
    g.N = (i64)W * H;
    g.in_origin = (i64)a.in_row0 * W;
    g.in_lo = std::max<i64>(0, g.in_origin);
    g.in_hi = std::min<i64>(g.N, (i64)(a.in_row0 + a.in_rows) * W);
    g.out_origin = (i64)a.y_begin * W;
    g.in_frame_stride = in_frame;
    g.out_frame_stride = out_frame;
    const int rows = a.y_end - a.y_begin;

    // separable Gaussians 17x17 .. 31x31: on the matrix cores (gauss_mfma_wide_kernel, within one level like the smaller
    // ones) where that applies, else the direct kernels (bit-exact, slower)
    if (a.filter == MI_FILTER_GAUSSIAN && a.ksize > 15) {
        if (aligned && a.ksize <= std::min(31, tuning().gauss_mfma_max) && tuning().gauss_mfma) {
            const int rc = launch_gauss_mfma(a.ksize, a.sigma, a.stream, in, out, g);
            if (rc != 1) return rc;
        }
        FilterArgs d = a;
        d.filter = MI_FILTER_GAUSSIAN_DIRECT;
        return launch_filter(d);
    }

    // mean blur 11x11 .. 31x31 on the matrix cores (exact; gauss_mfma.hip; MI_MEAN_MFMA moves the threshold,
    // MI_MEAN_MFMA_MAX the largest size)
    if (a.filter == MI_FILTER_MEAN && aligned && tuning().mean_mfma > 0 && a.ksize >= tuning().mean_mfma && a.ksize >= 5 &&
        a.ksize <= std::min(31, tuning().mean_mfma_max)) {
        const int rc = launch_mean_mfma(a.ksize, a.stream, in, out, g);
        if (rc != 1) return rc;
    }

    // generic one-thread-per-pixel kernels
    const bool direct_march = a.filter == MI_FILTER_GAUSSIAN_DIRECT && a.ksize >= 3 && a.ksize <= 7;
    const bool direct = a.filter == MI_FILTER_GAUSSIAN_DIRECT && !direct_march;
    const bool mean_prefix = a.filter == MI_FILTER_MEAN && a.ksize >= 17 && a.ksize <= 31 && mean_magic32(a.ksize) &&
                             tuning().mean_prefix;
    const bool mean_generic = a.filter == MI_FILTER_MEAN && (a.ksize < 3 || a.ksize > 15) && !mean_prefix;
    // sizes above 31 (the reference takes any odd size, imageTools.cu:313-340): the LDS-tile kernel while its tile and
    // border fit 64 KiB (k <= 49), one lane per pixel beyond
    const size_t mean_tile_lds = ((size_t)(MT_H + 2 * (a.ksize / 2)) * (MT_W + 2 * (a.ksize / 2)) +
                                  2 * (size_t)(MT_H + 2 * (a.ksize / 2)) * MT_W) * sizeof(u32);
    if (mean_generic && a.ksize >= 17 && mean_tile_lds <= 64 * 1024 && mean_magic32(a.ksize)) {
        g.rows_per_band = 1; g.bands = rows; g.strips = 1; g.total_waves = 0; g.xcd_remap = 0;
        const size_t lds = mean_tile_lds;
        const int tiles_y = (rows + MT_H - 1) / MT_H;
        for (int f0 = 0; f0 < a.batch; f0 += 65535) {
            const int nb = std::min(65535, a.batch - f0);
            for (int t0 = 0; t0 < tiles_y; t0 += 65535) {
                Geo gg = g;
                gg.y_begin = a.y_begin + t0 * MT_H;
                const dim3 grid((W + MT_W - 1) / MT_W, std::min(65535, tiles_y - t0), nb);
                mean_tile_kernel<<<grid, 256, lds, a.stream>>>(in + (i64)f0 * in_frame, out + (i64)f0 * out_frame, gg,
                                                              a.ksize, mean_magic32(a.ksize));
            }
        }
        return hip_status(hipGetLastError(), "tiled mean kernel launch");
    }
    if (direct && a.ksize >= 9 && a.ksize <= MI_MAX_KERNEL_SIZE) {
        g.rows_per_band = 1; g.bands = rows; g.strips = 1; g.total_waves = 0; g.xcd_remap = 0;
        Weights2D wt;
        gaussian_kernel_2d(a.ksize, a.sigma, wt.w);
        const int tiles_y = (rows + MT_H - 1) / MT_H;
        for (int f0 = 0; f0 < a.batch; f0 += 65535) {
            const int nb = std::min(65535, a.batch - f0);
            for (int t0 = 0; t0 < tiles_y; t0 += 65535) {
                Geo gg = g;
                gg.y_begin = a.y_begin + t0 * MT_H;
                const dim3 grid((W + MT_W - 1) / MT_W, std::min(65535, tiles_y - t0), nb);
                const u32* pin = in + (i64)f0 * in_frame;
                u32* pout = out + (i64)f0 * out_frame;
                switch (a.ksize) {
#define MI_TILE_CASE(K) case K: gauss_direct_tile_kernel<K><<<grid, 256, 0, a.stream>>>(pin, pout, gg, wt); break;
                MI_TILE_CASE(9) MI_TILE_CASE(11) MI_TILE_CASE(13) MI_TILE_CASE(15) MI_TILE_CASE(17) MI_TILE_CASE(19)
                MI_TILE_CASE(21) MI_TILE_CASE(23) MI_TILE_CASE(25) MI_TILE_CASE(27) MI_TILE_CASE(29) MI_TILE_CASE(31)
#undef MI_TILE_CASE
                default: return fail(MI_ERR_BAD_KERNEL_SIZE, "direct Gaussian: kernel size %d", a.ksize);
                }
            }
        }
        return hip_status(hipGetLastError(), "tiled direct Gaussian kernel launch");
    }
    if (direct || mean_generic) {
        g.rows_per_band = 1; g.bands = rows; g.strips = 1; g.total_waves = 0; g.xcd_remap = 0;
        // Gaussian sizes above 31: the k*k weights (host-generated, like generateGaussianKernal's) exceed the
        // kernel arguments and go through a device buffer made for this call -- such a call allocates,
        // copies synchronously and waits for its stream before it frees the buffer (not capturable)
        float* big_weights = nullptr;
        if (direct && a.ksize > MI_MAX_KERNEL_SIZE) {
            std::vector<float> w((size_t)a.ksize * a.ksize);
            gaussian_kernel_2d(a.ksize, a.sigma, w.data());
            hipError_t e = hipMalloc((void**)&big_weights, w.size() * sizeof(float));
            if (e == hipSuccess) e = hipMemcpy(big_weights, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) { if (big_weights) (void)hipFree(big_weights); return hip_status(e, "Gaussian weight table"); }
        }
        struct FreeWeights {
            float* p; hipStream_t s;
            ~FreeWeights() { if (p) { (void)hipStreamSynchronize(s); (void)hipFree(p); } }
        } free_weights{big_weights, a.stream};
        for (int f0 = 0; f0 < a.batch; f0 += 65535) {
            const int nb = std::min(65535, a.batch - f0);
            for (int r0 = 0; r0 < rows; r0 += 65535) {
                const int nr = std::min(65535, rows - r0);
                Geo gg = g;
                gg.y_begin = a.y_begin + r0;
                const dim3 grid((W + 255) / 256, nr, nb);
                const u32* pin = in + (i64)f0 * in_frame;
                u32* pout = out + (i64)f0 * out_frame;
                if (direct && a.ksize > MI_MAX_KERNEL_SIZE) {
                    gauss_direct_ptr_kernel<<<grid, 256, 0, a.stream>>>(pin, pout, gg, a.ksize, big_weights);
                } else if (direct) {
                    Weights2D wt;
                    gaussian_kernel_2d(a.ksize, a.sigma, wt.w);
                    gauss_direct_kernel<<<grid, 256, 0, a.stream>>>(pin, pout, gg, a.ksize, wt);
                } else {
                    mean_generic_kernel<<<grid, 256, 0, a.stream>>>(pin, pout, gg, a.ksize);
                }
            }
        }
        return hip_status(hipGetLastError(), "generic filter kernel launch");
    }

    // wave-march kernels
    const bool median5 = a.filter == MI_FILTER_MEDIAN && a.ksize == 5;
    // a wave covers 64 lanes x 4 pixels; the 5x5 median keeps one margin lane on each side
    g.strips = median5 ? (W + 247) / 248 : mean_prefix ? (W + 223) / 224 : (W + 255) / 256;
    // Rows per band.  Short bands keep the waves that run together on nearby rows, so the
    // access stream stays close to a front-to-back sweep (measured on MI355X: 3x3 filters peak
    // at 6 or 12 rows per band with rows requested in bursts of two, the 5x5 Gaussian at 16-24); the halo rows a short band re-reads
    // are shared with the neighbouring band through L2 (see March), not re-fetched from HBM.
    // A taller band amortises the row pass of a wide Gaussian over more output rows.
    int rpb = tuning().rows_per_band;
    if (rpb <= 0) {
        const int r = (a.filter == MI_FILTER_GAUSSIAN || a.filter == MI_FILTER_MEAN || direct_march) ? a.ksize / 2 : 1;
        rpb = r == 1 ? 12 : std::min(10 * r, 64);
        if (median5) rpb = 32;   // VALU-bound: amortise the 4 extra rows of a band
        // wide Gaussians: 2R extra row passes per band, two waves per SIMD -> taller bands, fewer waves
        const i64 min_waves = r >= 5 ? 2048 : 4096;
        if (r >= 5) rpb = 128;
        while (rpb > 4 && (i64)a.batch * g.strips * ((rows + rpb - 1) / rpb) < min_waves) rpb >>= 1;
    }
    // POL 1 addresses a band through 32-bit buffer offsets: keep a band under 1 GiB
    rpb = std::max(1, std::min(rpb, (int)((1LL << 28) / W)));
    g.rows_per_band = rpb;
    g.bands = (rows + rpb - 1) / rpb;
    const i64 total = (i64)a.batch * g.bands * g.strips;
    if (total > 0x7fffffff / 64) return fail(MI_ERR_UNSUPPORTED, "launch too large (%lld waves)", (long long)total);
    g.total_waves = (int)total;
    g.xcd_remap = 1;
    const int pol = (aligned && band_fits_buffer_offsets(rpb, W)) ? 1 : 0;
    const int grid = (int)((total + 3) / 4);

    switch (a.filter) {
    case MI_FILTER_SOBEL:
    {
#define MI_L(P) sobel_kernel<P><<<grid, 256, lds_pad(1, g.total_waves), a.stream>>>(in, out, g)
        MI_POLICY_CASES(pol, MI_L)
#undef MI_L
    }
        break;
    case MI_FILTER_MEAN:
        if (a.ksize == 3) {
#define MI_L(P) mean3_kernel<P><<<grid, 256, lds_pad(1, g.total_waves), a.stream>>>(in, out, g)
            MI_POLICY_CASES(pol, MI_L)
#undef MI_L
        } else if (mean_prefix) {
            const u32 m32 = mean_magic32(a.ksize);
            switch (a.ksize / 2) {
            case 8: launch_mean_prefix<8>(pol, grid, a.stream, in, out, g, m32); break;
            case 9: launch_mean_prefix<9>(pol, grid, a.stream, in, out, g, m32); break;
            case 10: launch_mean_prefix<10>(pol, grid, a.stream, in, out, g, m32); break;
            case 11: launch_mean_prefix<11>(pol, grid, a.stream, in, out, g, m32); break;
            case 12: launch_mean_prefix<12>(pol, grid, a.stream, in, out, g, m32); break;
            case 13: launch_mean_prefix<13>(pol, grid, a.stream, in, out, g, m32); break;
            case 14: launch_mean_prefix<14>(pol, grid, a.stream, in, out, g, m32); break;
            default: launch_mean_prefix<15>(pol, grid, a.stream, in, out, g, m32); break;
            }
        } else {
            const u32 magic = mean_magic(a.ksize);
            if (!magic) return fail(MI_ERR_UNSUPPORTED, "no exact reciprocal for mean kernel size %d", a.ksize);
            switch (a.ksize / 2) {
            case 2: launch_mean_sep<2>(pol, grid, a.stream, in, out, g, magic); break;
            case 3: launch_mean_sep<3>(pol, grid, a.stream, in, out, g, magic); break;
            case 4: launch_mean_sep<4>(pol, grid, a.stream, in, out, g, magic); break;
            case 5: launch_mean_sep<5>(pol, grid, a.stream, in, out, g, magic); break;
            case 6: launch_mean_sep<6>(pol, grid, a.stream, in, out, g, magic); break;
            case 7: launch_mean_sep<7>(pol, grid, a.stream, in, out, g, magic); break;
            default: return fail(MI_ERR_BAD_KERNEL_SIZE, "separable mean supports kernel sizes 5..15");
            }
        }
        break;
    case MI_FILTER_MEDIAN:
    {
#define MI_L(P) if (median5) median5_kernel<P><<<grid, 256, 0, a.stream>>>(in, out, g); \
                else median3_kernel<P><<<grid, 256, lds_pad(1, g.total_waves), a.stream>>>(in, out, g)
        MI_POLICY_CASES(pol, MI_L)
#undef MI_L
    }
        break;
    case MI_FILTER_GAUSSIAN_DIRECT: {
        GaussW2 gw;
        gaussian_kernel_2d(a.ksize, a.sigma, gw.w);
        if (a.ksize == 3) {
            launch_gauss_direct<1>(pol, grid, a.stream, in, out, g, gw);
        } else if (!tuning().gauss_direct_win) {
            if (a.ksize == 5) launch_gauss_direct<2>(pol, grid, a.stream, in, out, g, gw);
            else launch_gauss_direct<3>(pol, grid, a.stream, in, out, g, gw);
        } else {                                      // 5x5, 7x7: bound by their arithmetic (gauss_direct.hip)
            const int rc = launch_gauss_direct_win(a.ksize, pol, grid, a.stream, in, out, g, gw.w);
            if (rc) return rc;
        }
        break;
    }
    case MI_FILTER_GAUSSIAN: {
        if (aligned && a.ksize >= 7 && a.ksize <= 15 && tuning().gauss_mfma && tuning().gauss_wide) {
            const int rc = launch_gauss_mfma(a.ksize, a.sigma, a.stream, in, out, g);
            if (rc != 1) return rc;
        }
        GaussW gw;
        gaussian_kernel_1d(a.ksize, a.sigma, gw.w);
        switch (a.ksize / 2) {
        case 1: launch_gauss_sep<1>(pol, grid, a.stream, in, out, g, gw); break;
        case 2: launch_gauss_sep<2>(pol, grid, a.stream, in, out, g, gw); break;
        case 3: case 4: {
            GaussWide ww;
            if (tuning().gauss_wide && gauss_wide_weights(a.ksize, a.sigma, &ww)) {
                if (a.ksize == 7) launch_gauss_wide<3>(pol, grid, a.stream, in, out, g, ww);
                else launch_gauss_wide<4>(pol, grid, a.stream, in, out, g, ww);
            } else if (a.ksize == 7) launch_gauss_sep<3>(pol, grid, a.stream, in, out, g, gw);
            else launch_gauss_sep<4>(pol, grid, a.stream, in, out, g, gw);
            break;
        }
        case 5: case 6: case 7: {
            GaussWide ww;
            if (tuning().gauss_wide && gauss_wide_weights(a.ksize, a.sigma, &ww)) {
                if (a.ksize == 11) launch_gauss_wide<5>(pol, grid, a.stream, in, out, g, ww);
                else if (a.ksize == 13) launch_gauss_wide<6>(pol, grid, a.stream, in, out, g, ww);
                else launch_gauss_wide<7>(pol, grid, a.stream, in, out, g, ww);
            } else if (a.ksize == 11) launch_gauss_sep<5>(pol, grid, a.stream, in, out, g, gw);
            else if (a.ksize == 13) launch_gauss_sep<6>(pol, grid, a.stream, in, out, g, gw);
            else launch_gauss_sep<7>(pol, grid, a.stream, in, out, g, gw);
            break;
        }
        default: return fail(MI_ERR_BAD_KERNEL_SIZE, "separable Gaussian supports kernel sizes 3..15");
        }
        break;
    }
    default:
        return fail(MI_ERR_INVALID_ARG, "unknown filter %d", a.filter);
    }
    return hip_status(hipGetLastError(), "filter kernel launch");
}

}  // namespace mi
