// gauss_mfma.hip -- separable Gaussian blur 7x7 .. 31x31 and exact mean blur 5x5 .. 31x31 with both passes on the matrix
// cores (gfx950).  gauss_mfma_kernel: radius <= 8; gauss_mfma_wide_kernel (further down): radius 9 .. 15.
//
// Reference: imageGaussianBlur, /root/reference/kernels/imageTools.cu:287-311 (direct k x k fp32 sum, truncating
// store).  The separable form is allowed +-1 level; gauss_wide_kernel (filters.hip) meets it in
// 16-bit fixed point on the vector ALUs and is bound by them (75 vector instructions per pixel at 15x15, two waves
// per SIMD: 0.44 of the HBM roofline).  A 1-D filter over 16 outputs is a banded (Toeplitz) 32 x 16 matrix applied
// to a window of 32 inputs, so each pass of a 16 x 16-pixel tile is v_mfma_f32_16x16x32_f16 instructions:
//
//   row pass     D[row][x]  = sum_k  A[row][k] * T[k][x]      A = 16 image rows x (16 pixels x 2 channels) as f16
//   column pass  D2[x][y]   = sum_k  A2[x][k]  * T2[k][y]     A2 = the row pass's result for 32 rows, as f16
//
// What makes this cheap is that the contraction index k may be permuted freely (the same way in both operands):
//   * a lane's four RGBA pixels (one 16-byte chunk) become row-pass operand registers with ONE v_perm_b32 per channel
//     and pixel pair: the bytes (b0, 0, b1, 0) are the f16 pair (b0 * 2^-24, b1 * 2^-24) -- f16 SUBNORMALS, which the
//     MFMA takes at face value -- so the weight operand carries 2^15 and the column pass's the remaining 2^9; a tile's
//     operand is the chunks of its two halves (pixels 16 t - 8 .. 16 t + 7 and the next 16), k slot = (half, pixel);
//   * the row pass's result (lane: pixel x = lane % 16, rows 4 * (lane / 16) .. + 3) IS the column pass's operand
//     layout once k is read as (row block, lane / 16, row % 4): two v_cvt_pk_f16_f32 and no cross-lane traffic;
//   * with the column pass transposed (pixels as M, output rows as N) a lane ends up with four adjacent pixels of one
//     output row: a 16-byte chunk.
// 12 MFMAs and about 45 vector instructions per 256 pixels where the fixed-point kernel spends 300 vector instructions.
// (Round 3's first form interleaved two channels in one operand -- (1024 + R, 1024 + G) as 0x6400 | byte -- with zero
// weights in the other channel's slots: 18 MFMAs per tile, 242 registers, 1.5 % slower; the mean 6.5 % slower.)
//
// Execution: one wave per workgroup owns a strip of 128 pixels (8 tiles) and walks down 16-row blocks aligned to
// ABSOLUTE image rows (so the arithmetic of an output pixel does not depend on how a launch is cut into bands or
// strips: any decomposition gives the same bytes).  A block's rows arrive by LDS-DMA (buffer_load ... lds, 16 bytes
// per lane) in two buffers used in turn; the 16-byte chunk c of row r lands in slot (c + r) % 32 of the row (the DMA's
// lanes fetch their chunks in that order), which makes both the operand reads (16 rows, one chunk) and the result
// writes conflict-free.  Results go back into the buffer they came from and leave as dense 512-byte row segments.
//
// Arithmetic (Gaussian): each weight as TWO f16 values (leading part and remainder; the centre's remainder rounded UP
// so that sum(w) >= 1: a constant image stays constant), row-pass results kept as f16 of (value - 128) / 512 (spacing
// <= 1/16 of a level), fp32 accumulation, 1/256 added before the truncating store.  Error before truncation < 0.05,
// inside the +-1 level the separable form is allowed.
#include <algorithm>
#include <cmath>
#include <mutex>

#include "common.h"
#include "internal.h"

namespace mi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

struct GaussMfma {
    float w[17];     // weights w[0 .. 2R], zero beyond, as two f16 values each: w = w[t] + wl[t] to 2^-22 of its size
    float wl[17];
    float c1;        // row-pass accumulator start: -128 / 512 (mean: 2^14)
    float c2;        // column-pass accumulator start: 128 * sum(w) + bias - 0.5 (v_cvt_pk_u8_f32 rounds to nearest)
    float rcp, d;    // mean (MODE 1): output = nearest integer to sum * rcp + d == sum / k^2, checked for every sum on the host
    int R;
    int ob_first;    // first output block of the launch: output block b holds rows 16 b + 8 .. 16 b + 23
    int nob;         // output blocks of the launch
    int bpb;         // output blocks per band
};

#ifndef MI_GM_WAVES
#define MI_GM_WAVES 2
#endif
#ifndef MI_GM_DBG
#define MI_GM_DBG 0       // experiment builds only (wrong output): 1 = no stores, 2 = no loads, 3 = neither, 4 = loads and stores without the arithmetic, 5 = the same without the halo chunks, 6 = as 5 and without a band's first (priming) block: with MI_ROWS_PER_BAND=16 a plain copy in 16-row x 512-byte tiles
#endif
#ifndef MI_GM_BIAS
#define MI_GM_BIAS (1.0 / 256.0)   // added before the truncating store: covers fp32 accumulation error, so that a constant image stays constant
#endif

[[maybe_unused]] constexpr int GM_MAIN = 16 * 512;          // 16 rows x 128 pixels
#define GM_BUF (GM_MAIN + 1024)            // + 16 rows x (8 pixels left, 8 pixels right)

template <int N>
static MI_DEV void gm_wait_vm()
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (MI_ROWFEED_FULL_WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
static MI_DEV void gm_wait_lds()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

// MODE 0: Gaussian.  MODE 1: mean blur (imageMeanBlur, imageTools.cu:313-340) of size 2R + 1 <= 17, EXACT: weights 1.0
// (times 2^15), and the row-pass accumulator starts at 2^14, whose last place is the pass's unit 2^-9 -- the integer row
// sum (<= 17 * 255) is then simply the low 16 bits of the fp32 result.  The window keeps the sums two to a register;
// they enter the column pass as two exact parts, their high bytes against weights 256 and their low bytes against weights 1
// (f16 subnormals again; one v_perm_b32 and one v_and_b32 per register pair); every partial sum is an integer multiple of the unit below 2^24 units, hence exact whatever the
// order inside the MFMA; the integer division by k^2 is one fused multiply-add and the rounding of v_cvt_pk_u8_f32,
// verified for every reachable sum on the host.  Table part 0 = 1.0 in the window, part 1 = 256.0; the column weights carry
// 2^7 instead of 2^9, so the column pass leaves sum * 2^-17.
// NT: tiles per strip.  8 = 128-pixel strips (512-byte row segments, 18 KiB of LDS and 214 registers a wave: two waves per
// SIMD); 4 = 64-pixel strips (256-byte segments, 10 KiB, three waves per SIMD; MI_GM_STRIP=64: 4-5 % slower); 16 = 256-pixel
// strips (1-KiB segments, 34 KiB, one wave per SIMD; MI_GM_STRIP=256: 20 % slower).  Same bytes.
template <int MODE, int NT>
__global__ __launch_bounds__(64, NT == 8 ? MI_GM_WAVES : NT == 4 ? 3 : 1) void gauss_mfma_kernel(const u32* __restrict__ in_, u32* __restrict__ out_,
                                                                      const Geo g, const GaussMfma gm)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const WaveJob j = wave_job(g);
    if (!j.valid) return;
    const int lane = j.lane, r16 = lane & 15, grp = lane >> 4;
    const int R = gm.R, W = g.W, pitch = W * 4;
    constexpr int SL = 4 * NT;            // 16-byte chunks (LDS slots) per row of the strip
    constexpr int ROWB = 16 * SL;         // bytes per row of the strip
    constexpr int RPI = 64 / SL;          // rows one DMA or store instruction covers
    constexpr int NI = 16 / RPI;          // such instructions per 16-row block
    constexpr int MAIN = 16 * ROWB, BUF = MAIN + 1024;
    const int cx0 = j.strip * (16 * NT);
    const int ntile = min(NT, (W - cx0 + 15) >> 4);
    // this band's output blocks [ob0, ob0 + n) and output rows [ya, yb)
    const int ob0 = gm.ob_first + j.band * gm.bpb;
    const int n = min(gm.bpb, gm.ob_first + gm.nob - ob0);
    const int ya = max(g.y_begin, 16 * ob0 + 8), yb = min(g.y_end, 16 * (ob0 + n) + 8);
    // (odd bands walking upwards, as in the wide kernel below and the vector-ALU marches, measured no gain here: 0.2035
    // against 0.2032 ms per 16 x 4K, 1.597 against 1.589 ms on 32768^2 -- with 64-row bands L2 already serves half of the overlap)

    __shared__ __attribute__((aligned(1024))) unsigned char ring[2 * BUF];

    // ---- Toeplitz operands, built once per wave from a zero-padded weight table in LDS: tbl[24 + t] = w[t]
    {
        float wv = 0.f, wlv = 0.f;
#pragma unroll
        for (int t = 0; t < 17; t++) { wv = (lane == 24 + t) ? gm.w[t] : wv; wlv = (lane == 24 + t) ? gm.wl[t] : wlv; }
        reinterpret_cast<float*>(ring)[lane] = wv;
        reinterpret_cast<float*>(ring)[64 + lane] = wlv;
    }
    const float* tbl = reinterpret_cast<const float*>(ring);
    // part 0: the weights' leading f16, part 1: their remainder (mean: 1.0 and 256.0).  ONE row-pass operand per part, for
    // every channel: k slot i < 4 = pixel 4 grp + i of the tile's first half (pixels 16 t - 8 ..), i >= 4 = pixel
    // 4 grp + i - 4 of its second half; output pixel r16; tap index = input pixel - output pixel + R.  The pixel bytes
    // enter as f16 subnormals (byte * 2^-24), so these weights carry 2^15 and the column pass's 2^9.
    // Column pass: k slot i = row 4 grp + i of the upper block (i < 4), row 4 grp + i - 4 of the lower one; output row 8 + r16.
    h8 brow[2], b2s[2];
#pragma unroll
    for (int part = 0; part < 2; part++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int relx = (i < 4 ? 4 * grp + i : 16 + 4 * grp + (i - 4)) - 8;
            brow[part][i] = (_Float16)(tbl[64 * part + 24 + relx - r16 + R] * 32768.f);
            const int rely = i < 4 ? 4 * grp + i : 16 + 4 * grp + (i - 4);
            b2s[part][i] = (_Float16)(tbl[64 * part + 24 + rely - 8 - r16 + R] * (MODE == 0 ? 512.f : 128.f));     // (mean: unit 2^-17, so that 256 * 128 fits f16)
        }
    gm_wait_lds();       // the table has been read: the ring is free for the first rows

    // ---- addressing
    const u32* in = in_ + (i64)j.frame * g.in_frame_stride;
    u32* out = out_ + (i64)j.frame * g.out_frame_stride;
    const i64 fmin = (i64)(16 * ob0) * W + cx0 - 8;              // lowest flat pixel index this wave can ask for
    const i64 base = fmin > g.in_lo ? fmin : g.in_lo;
    i64 bytes = (g.in_hi - base) * 4;
    bytes = bytes < 0 ? 0 : (bytes > 0x7FFFFF00LL ? 0x7FFFFF00LL : bytes);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (base - g.in_origin)), 0, (int)bytes, 0x00020000);
    const int delta0 = (int)((fmin + 8 - base) * 4);             // byte offset of (row 16 ob0, pixel cx0)
    // main rows: DMA instruction i brings rows RPI i .. of the block; lane -> (row RPI i + hi, slot), chunk (slot - row) % SL
    const int hi = lane / SL, slot = lane % SL;
    const int c0 = (slot - hi) & (SL - 1);
    const int main_lane = delta0 + hi * pitch;
    // halo: lane -> (group q = lane / 16, row r16); groups 0, 1 = chunks -2, -1; groups 2, 3 = chunks SL, SL + 1
    const int halo_lane = delta0 + r16 * pitch + (grp < 2 ? grp - 2 : SL - 2 + grp) * 16;
    auto request = [&](int s, int b) __attribute__((always_inline)) {     // block s of the band -> buffer b
        if (MI_GM_DBG == 2 || MI_GM_DBG == 3 || (MI_GM_DBG == 6 && s == 0)) return;
        unsigned char* dst = ring + b * BUF;
        const int rowoff = 16 * s * pitch;
#pragma unroll
        for (int i = 0; i < NI; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + 1024 * i), 16,
                                                     main_lane + rowoff + RPI * i * pitch + (((c0 - RPI * i) & (SL - 1)) << 4), 0, 0, 0);
        if (MI_GM_DBG != 5 && MI_GM_DBG != 6)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + MAIN), 16,
                                                 halo_lane + rowoff, 0, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(out + ((i64)ya * W - g.out_origin)), 0,
                                                                        max(yb - ya, 0) * pitch, 0x00020000);
    const int st_px = cx0 + 4 * slot;                              // first pixel of the chunk this lane stores
    const int st_lane = st_px < W ? hi * pitch + st_px * 4 : (int)0x80000000;
    const bool st_first = st_px == 0, st_last = st_px + 4 == W;

    // LDS byte offsets (within a buffer) of this lane's operand chunks: half j = pixels 16 j - 8 .. 16 j + 7, chunk 4 j - 2 + grp
    const int kap = (grp - 2 + r16) & (SL - 1);
    const int halo_at = MAIN + ((16 * grp + r16) << 4);
    const int rd_row = r16 * ROWB;
    const int wr_kap = (grp + r16) & (SL - 1);

    const f32x4 c1v = {gm.c1, gm.c1, gm.c1, gm.c1}, c2v = {gm.c2, gm.c2, gm.c2, gm.c2};
    u32 win[NT][3][2];          // the row pass of the upper block, per tile and channel: rows 4 grp .. + 3 (see `cur` below)

    request(0, 0);
    request(1, 1);
    for (int s = 0; s <= n; s++) {
        const int b = s & 1;
        if (MI_GM_DBG == 2 || MI_GM_DBG == 3) {}
        else if (MI_GM_DBG == 6) { if (s >= 1) gm_wait_vm<0>(); }
        else if (MI_GM_DBG == 5) { if (s + 1 <= n) { if (s <= 1) gm_wait_vm<NI>(); else gm_wait_vm<2 * NI>(); } else gm_wait_vm<0>(); }
        else if (s + 1 <= n) { if (s <= 1 || MI_GM_DBG == 1) gm_wait_vm<NI + 1>(); else gm_wait_vm<2 * NI + 1>(); }
        else gm_wait_vm<0>();
        unsigned char* buf = ring + b * BUF;
        const bool prime = s == 0;

        u32x4 raw[NT + 1];          // this lane's chunk of half h = pixels 16 h - 8 .. 16 h + 7 of the block's 16 rows
#pragma unroll
        for (int h = 0; h <= NT; h++) {
            int at = rd_row + (((kap + 4 * h) & (SL - 1)) << 4);
            if (h == 0) at = grp < 2 ? halo_at : at;
            if (h == NT) at = grp < 2 ? at : halo_at;
            raw[h] = *reinterpret_cast<const u32x4*>(buf + at);
        }
        // a half's pixels as f16 pairs per channel, (pixel 0, pixel 1) and (pixel 2, pixel 3): bytes (b0, 0, b1, 0)
        u32 aP[3][2];
        auto pairs = [&](const u32x4& q, u32 (&pp)[3][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                pp[c][0] = __builtin_amdgcn_perm(q.y, q.x, 0x0c040c00u + (u32)c * 0x00010001u);
                pp[c][1] = __builtin_amdgcn_perm(q.w, q.z, 0x0c040c00u + (u32)c * 0x00010001u);
            }
        };
        pairs(raw[0], aP);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (t < ntile && MI_GM_DBG < 4) {
                u32 nP[3][2];
                pairs(raw[t + 1], nP);
                f32x4 mid[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const h8 a = __builtin_bit_cast(h8, u32x4{aP[c][0], aP[c][1], nP[c][0], nP[c][1]});
                    if (MODE == 0) {     // (the small terms first: they are added to the start value at full precision)
                        mid[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, brow[1], c1v, 0, 0, 0);
                        mid[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, brow[0], mid[c], 0, 0, 0);
                    } else {
                        mid[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, brow[0], c1v, 0, 0, 0);
                    }
                    aP[c][0] = nP[c][0]; aP[c][1] = nP[c][1];
                }
                // what the window keeps of this block's rows 4 grp .. + 3, two registers per channel: MODE 0 f16 pairs; MODE 1
                // the integer row sums (below 2^13), two to a register
                u32 cur[3][2];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if constexpr (MODE == 0) {
                        cur[c][0] = __builtin_bit_cast(u32, __builtin_convertvector(f32x2v{mid[c].x, mid[c].y}, h2));
                        cur[c][1] = __builtin_bit_cast(u32, __builtin_convertvector(f32x2v{mid[c].z, mid[c].w}, h2));
                    } else {     // the accumulator started at 2^14, whose last place is the row pass's unit 2^-9: the sum is its low 16 bits
                        // (copies first: __builtin_bit_cast of a vector ELEMENT read element 0 whatever the index, hipcc 7.2)
                        const float m0 = mid[c][0], m1 = mid[c][1], m2 = mid[c][2], m3 = mid[c][3];
                        cur[c][0] = __builtin_amdgcn_perm(__float_as_uint(m1), __float_as_uint(m0), 0x05040100u);
                        cur[c][1] = __builtin_amdgcn_perm(__float_as_uint(m3), __float_as_uint(m2), 0x05040100u);
                    }
                }
                if (!prime) {
                    f32x4 o3[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const u32x4 a2 = {win[t][c][0], win[t][c][1], cur[c][0], cur[c][1]};
                        if constexpr (MODE == 0) {
                            o3[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a2), b2s[1], c2v, 0, 0, 0);
                            o3[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a2), b2s[0], o3[c], 0, 0, 0);
                        } else {     // the two exact parts of the sums, sum >> 8 (weights 256) and sum & 255 (weights 1), as f16 subnormals:
                            u32x4 ah, al;     // one v_perm_b32 and one v_and_b32 per register pair
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                ah[i] = __builtin_amdgcn_perm(a2[i], a2[i], 0x0c030c01u);
                                al[i] = a2[i] & 0x00ff00ffu;
                            }
                            o3[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ah), b2s[1], c2v, 0, 0, 0);
                            o3[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, al), b2s[0], o3[c], 0, 0, 0);
                        }
                    }
                    u32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        u32 v = 0xFF000000u;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const float f = MODE == 0 ? o3[c][i] : __builtin_fmaf(o3[c][i], gm.rcp, gm.d);
                            v = __builtin_amdgcn_cvt_pk_u8_f32(f, c, v);
                        }
                        o[i] = v;
                    }
                    // output row r16 of the block, pixels 16 t + 4 grp .. + 3: chunk 4 t + grp, slot (chunk + row) % 32
                    *reinterpret_cast<u32x4*>(buf + rd_row + (((wr_kap + 4 * t) & (SL - 1)) << 4)) = o;
                }
#pragma unroll
                for (int c = 0; c < 3; c++) { win[t][c][0] = cur[c][0]; win[t][c][1] = cur[c][1]; }
            }
        }
        if (!prime) {
            // the block's 16 output rows, 16 ob + 8 + r for output block ob = ob0 + s - 1, as dense row segments
            const int y0 = 16 * (ob0 + s - 1) + 8;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const int r = RPI * i + hi;
                u32x4 v = *reinterpret_cast<const u32x4*>(buf + r * ROWB + (((slot + r) & (SL - 1)) << 4));
                const int y = y0 + r;
                const bool rowok = y >= 1 && y <= g.H - 2;
                v.x = (rowok && !st_first) ? v.x : 0u;
                v.y = rowok ? v.y : 0u;
                v.z = rowok ? v.z : 0u;
                v.w = (rowok && !st_last) ? v.w : 0u;
                if (MI_GM_DBG == 1 || MI_GM_DBG == 3) { if (v.x == 0x12345678u && v.y == 0x9abcdef0u) __builtin_amdgcn_raw_buffer_store_b128(v, rout, st_lane, 0, 2); }
                else __builtin_amdgcn_raw_buffer_store_b128(v, rout, st_lane + (y0 + RPI * i - ya) * pitch, 0, 2);
            }
        }
        gm_wait_lds();      // every read of this buffer has returned before the DMA may overwrite it
        if (s + 2 <= n) request(s + 2, b);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Sizes 19x19 .. 31x31 (radius 9 .. 15): the same scheme on ALIGNED 16-pixel halves and 16-row blocks.  An output tile
// needs the halves t - 1, t, t + 1 and the blocks above, at and below it, so each pass of a tile is three K = 16 matrix
// instructions per part (mean) or two K = 32 ones (Gaussian): a lane's 16-byte chunk is the K = 16 operand of a channel
// after the two v_perm_b32, two chunks side by side a K = 32 one; the row pass's result for a block is the column pass's
// operand for that block as it stands, the window keeps the row pass of TWO earlier
// blocks, the halo is 16 pixels on each side (two DMA instructions a block), and a band of n output blocks loads n + 2.
struct GaussMfmaWide {
    float w[31];     // as GaussMfma
    float wl[31];
    float c1, c2, rcp, d;
    int R;
    int ob_first;    // first output block of the launch: output block b holds rows 16 b .. 16 b + 15
    int nob;
    int bpb;
    int serp;        // 1: odd bands walk upwards, so that both bands at a boundary read its rows at the same time (the second from L2)
};

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2v __attribute__((ext_vector_type(2)));
#define GW_BUF (GM_MAIN + 2048)            // + 16 rows x (16 pixels left, 16 pixels right)

template <int MODE>
__global__ __launch_bounds__(64, 2) void gauss_mfma_wide_kernel(const u32* __restrict__ in_, u32* __restrict__ out_,
                                                                const Geo g, const GaussMfmaWide gm)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const WaveJob j = wave_job(g);
    if (!j.valid) return;
    const int lane = j.lane, r16 = lane & 15, grp = lane >> 4;
    const int R = gm.R, W = g.W, pitch = W * 4;
    const int cx0 = j.strip * 128;
    const int ntile = min(8, (W - cx0 + 15) >> 4);
    const int ob0 = gm.ob_first + j.band * gm.bpb;
    const int n = min(gm.bpb, gm.ob_first + gm.nob - ob0);
    const int ya = max(g.y_begin, 16 * ob0), yb = min(g.y_end, 16 * (ob0 + n));
    const bool rev = gm.serp && (j.band & 1);      // wave-uniform

    __shared__ __attribute__((aligned(1024))) unsigned char ring[2 * GW_BUF];

    // ---- Toeplitz operands from a zero-padded weight table in LDS: tbl[128 part + 32 + t] = w[t]
    {
        float wv = 0.f, wlv = 0.f;
#pragma unroll
        for (int t = 0; t < 31; t++) { wv = (lane == 32 + t) ? gm.w[t] : wv; wlv = (lane == 32 + t) ? gm.wl[t] : wlv; }
        float* T = reinterpret_cast<float*>(ring);
        T[lane] = wv; T[64 + lane] = 0.f;
        T[128 + lane] = wlv; T[192 + lane] = 0.f;
    }
    const float* tbl = reinterpret_cast<const float*>(ring);
    // operand d (0, 1, 2 = the half / block before, at, after the output's): k slot i = element 4 grp + i of it, output
    // element r16; tap index = 16 (d - 1) + 4 grp + i - r16 + R.  The same index serves rows and columns.
    // Gaussian: d = 0 and 1 share one K = 32 operand (k slot i < 4: d = 0, else d = 1); d = 2 is the second half of another
    // one whose first half (d = 1 again) meets zero weights: four matrix instructions per pass instead of six.
    // (v_mfma_f32_16x16x16_f16 takes the same 16 cycles as the K = 32 form on this part, tools/mfmabench.hip; a chain that
    // MIXES the two forms came out wrong -- two of a lane's four results zero -- with hipcc 7.2, so it is one or the other.)
    // Mean: three K = 16 operands.  Its vector ALUs are the busier unit, and the K = 32 form costs vector moves (the
    // middle half / block sits in two register tuples): measured 0.2375 ms against 0.2330 per 16 x 4K, while the
    // Gaussian gains 8 % (0.2524 -> 0.2327).
    h8 brow01[2], bcol01[2], brow2[2], bcol2[2];
    h4 brow[3], bcol[2][3];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int part = 0; part < 2; part++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float wt = tbl[128 * part + 32 + 16 * ((i >> 2) - 1) + 4 * grp + (i & 3) - r16 + R];
                brow01[part][i] = (_Float16)(wt * 32768.f);
                bcol01[part][i] = (_Float16)(wt * 512.f);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float wt = tbl[128 * part + 32 + 16 + 4 * grp + i - r16 + R];
                brow2[part][i] = (_Float16)0.f;
                bcol2[part][i] = (_Float16)0.f;
                brow2[part][4 + i] = (_Float16)(wt * 32768.f);
                bcol2[part][4 + i] = (_Float16)(wt * 512.f);
            }
        }
    } else {
#pragma unroll
        for (int d = 0; d < 3; d++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int at = 32 + 16 * (d - 1) + 4 * grp + i - r16 + R;
                brow[d][i] = (_Float16)(tbl[at] * 32768.f);
                bcol[0][d][i] = (_Float16)(tbl[at] * 128.f);            // against the sums' low bytes: unit 2^-17
                bcol[1][d][i] = (_Float16)(tbl[128 + at] * 128.f);      // against their high bytes: 256 * 128
            }
        // a band that walks upwards meets its blocks in the order lower, middle, upper: the mean's sums are exact in any
        // order, so it swaps the two outer weight operands once instead of selecting registers in every tile
        if (rev) {
#pragma unroll
            for (int part = 0; part < 2; part++) { const h4 x = bcol[part][0]; bcol[part][0] = bcol[part][2]; bcol[part][2] = x; }
        }
    }
    gm_wait_lds();

    // ---- addressing (as gauss_mfma_kernel, with input block s of the band = rows 16 (ob0 - 1 + s) ..)
    const u32* in = in_ + (i64)j.frame * g.in_frame_stride;
    u32* out = out_ + (i64)j.frame * g.out_frame_stride;
    const i64 fmin = (i64)(16 * (ob0 - 1)) * W + cx0 - 16;
    const i64 base = fmin > g.in_lo ? fmin : g.in_lo;
    i64 bytes = (g.in_hi - base) * 4;
    bytes = bytes < 0 ? 0 : (bytes > 0x7FFFFF00LL ? 0x7FFFFF00LL : bytes);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (base - g.in_origin)), 0, (int)bytes, 0x00020000);
    const int delta0 = (int)((fmin + 16 - base) * 4);            // byte offset of (row 16 (ob0 - 1), pixel cx0)
    const int hi = lane >> 5, slot = lane & 31;
    const int c0 = (slot - hi) & 31;
    const int main_lane = delta0 + hi * pitch;
    const int halo_lane = delta0 + r16 * pitch + (grp - 4) * 16;  // left: chunks -4 .. -1 of row r16; right: + 36 chunks
    auto request = [&](int s, int b) __attribute__((always_inline)) {
        unsigned char* dst = ring + b * GW_BUF;
        const int rowoff = 16 * (rev ? n + 1 - s : s) * pitch;       // step s brings input block s of the band, or n + 1 - s
#pragma unroll
        for (int i = 0; i < 8; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + 1024 * i), 16,
                                                     main_lane + rowoff + 2 * i * pitch + (((c0 - 2 * i) & 31) << 4), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + GM_MAIN), 16,
                                                 halo_lane + rowoff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + GM_MAIN + 1024), 16,
                                                 halo_lane + rowoff + 36 * 16, 0, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(out + ((i64)ya * W - g.out_origin)), 0,
                                                                        max(yb - ya, 0) * pitch, 0x00020000);
    const int st_px = cx0 + 4 * slot;
    const int st_lane = st_px < W ? hi * pitch + st_px * 4 : (int)0x80000000;
    const bool st_first = st_px == 0, st_last = st_px + 4 == W;

    const int kap = (grp + r16) & 31;      // chunk 4 h + grp of row r16 sits in slot (kap + 4 h) % 32
    const int rd_row = r16 * 512;
    const f32x4 c1v = {gm.c1, gm.c1, gm.c1, gm.c1}, c2v = {gm.c2, gm.c2, gm.c2, gm.c2};
    u32 w1[8][3][2], w2[8][3][2];          // the row pass of the previous block and of the one before it

    request(0, 0);
    request(1, 1);
    for (int s = 0; s <= n + 1; s++) {
        const int b = s & 1;
        if (s <= n) { if (s <= 2) gm_wait_vm<10>(); else gm_wait_vm<18>(); }
        else gm_wait_vm<0>();
        unsigned char* buf = ring + b * GW_BUF;

        // half h (-1 .. 8) of the block's row r16, pixels 16 h + 4 grp .. + 3, as f16 pairs per channel
        auto half = [&](int h, u32 (&pp)[3][2]) __attribute__((always_inline)) {
            const int at = h < 0 ? GM_MAIN + lane * 16 : h >= 8 ? GM_MAIN + 1024 + lane * 16 : rd_row + (((kap + 4 * h) & 31) << 4);
            const u32x4 q = *reinterpret_cast<const u32x4*>(buf + at);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                pp[c][0] = __builtin_amdgcn_perm(q.y, q.x, 0x0c040c00u + (u32)c * 0x00010001u);
                pp[c][1] = __builtin_amdgcn_perm(q.w, q.z, 0x0c040c00u + (u32)c * 0x00010001u);
            }
        };
        u32 pm[3][2], p0[3][2], pn[3][2];
        half(-1, pm);
        half(0, p0);
        half(1, pn);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (t < ntile) {
                u32 pf[3][2];          // the half after next, fetched before this tile's results overwrite chunk 4 t + grp
                if (t + 1 < ntile) half(t + 2, pf);
                f32x4 mid[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    f32x4 m = c1v;
                    if constexpr (MODE == 0) {
                        const h8 a01 = __builtin_bit_cast(h8, u32x4{pm[c][0], pm[c][1], p0[c][0], p0[c][1]});
                        const h8 a2 = __builtin_bit_cast(h8, u32x4{p0[c][0], p0[c][1], pn[c][0], pn[c][1]});     // (the first half meets zero weights)
                        m = __builtin_amdgcn_mfma_f32_16x16x32_f16(a01, brow01[1], m, 0, 0, 0);
                        m = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, brow2[1], m, 0, 0, 0);
                        m = __builtin_amdgcn_mfma_f32_16x16x32_f16(a01, brow01[0], m, 0, 0, 0);
                        m = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, brow2[0], m, 0, 0, 0);
                    } else {
                        m = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, u32x2v{pm[c][0], pm[c][1]}), brow[0], m, 0, 0, 0);
                        m = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, u32x2v{p0[c][0], p0[c][1]}), brow[1], m, 0, 0, 0);
                        m = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, u32x2v{pn[c][0], pn[c][1]}), brow[2], m, 0, 0, 0);
                    }
                    mid[c] = m;
                }
                u32 cur[3][2];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float m0 = mid[c][0], m1 = mid[c][1], m2 = mid[c][2], m3 = mid[c][3];
                    if constexpr (MODE == 0) {
                        cur[c][0] = __builtin_bit_cast(u32, __builtin_convertvector(f32x2v{m0, m1}, h2));
                        cur[c][1] = __builtin_bit_cast(u32, __builtin_convertvector(f32x2v{m2, m3}, h2));
                    } else {
                        cur[c][0] = __builtin_amdgcn_perm(__float_as_uint(m1), __float_as_uint(m0), 0x05040100u);
                        cur[c][1] = __builtin_amdgcn_perm(__float_as_uint(m3), __float_as_uint(m2), 0x05040100u);
                    }
                }
                if (s >= 2) {
                    f32x4 o3[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        // upper, middle, lower block in that order whichever way the band is walked: the same operations per pixel
                        f32x4 o = c2v;
                        if constexpr (MODE == 0) {
                            const u32 up0 = rev ? cur[c][0] : w2[t][c][0], up1 = rev ? cur[c][1] : w2[t][c][1];
                            const u32 lo0 = rev ? w2[t][c][0] : cur[c][0], lo1 = rev ? w2[t][c][1] : cur[c][1];
                            const u32x4 b01 = {up0, up1, w1[t][c][0], w1[t][c][1]};
                            const u32x4 b2 = {w1[t][c][0], w1[t][c][1], lo0, lo1};     // (the middle block again, against zero weights)
                            o = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, b01), bcol01[1], o, 0, 0, 0);
                            o = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, b2), bcol2[1], o, 0, 0, 0);
                            o = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, b01), bcol01[0], o, 0, 0, 0);
                            o = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, b2), bcol2[0], o, 0, 0, 0);
                        } else {
                            const u32x2v blk[3] = {{w2[t][c][0], w2[t][c][1]}, {w1[t][c][0], w1[t][c][1]}, {cur[c][0], cur[c][1]}};     // (see bcol)
#pragma unroll
                            for (int d = 0; d < 3; d++) {
                                // the two exact parts of the sums (below 2^13), sum >> 8 and sum & 255: one v_perm_b32 and one v_and_b32 per pair
                                const u32x2v ah = {__builtin_amdgcn_perm(blk[d][0], blk[d][0], 0x0c030c01u), __builtin_amdgcn_perm(blk[d][1], blk[d][1], 0x0c030c01u)};
                                const u32x2v al = {blk[d][0] & 0x00ff00ffu, blk[d][1] & 0x00ff00ffu};
                                o = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, ah), bcol[1][d], o, 0, 0, 0);
                                o = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4, al), bcol[0][d], o, 0, 0, 0);
                            }
                        }
                        o3[c] = o;
                    }
                    u32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        u32 v = 0xFF000000u;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const float f = MODE == 0 ? o3[c][i] : __builtin_fmaf(o3[c][i], gm.rcp, gm.d);
                            v = __builtin_amdgcn_cvt_pk_u8_f32(f, c, v);
                        }
                        o[i] = v;
                    }
                    // output row r16 of the block, pixels 16 t + 4 grp .. + 3: chunk 4 t + grp -- this lane's own, already read
                    *reinterpret_cast<u32x4*>(buf + rd_row + (((kap + 4 * t) & 31) << 4)) = o;
                }
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        w2[t][c][q] = w1[t][c][q]; w1[t][c][q] = cur[c][q];
                        pm[c][q] = p0[c][q]; p0[c][q] = pn[c][q]; pn[c][q] = pf[c][q];
                    }
            }
        }
        if (s >= 2) {
            const int y0 = 16 * (rev ? ob0 + n + 1 - s : ob0 + s - 2);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int r = 2 * i + hi;
                u32x4 v = *reinterpret_cast<const u32x4*>(buf + r * 512 + (((slot + r) & 31) << 4));
                const int y = y0 + r;
                const bool rowok = y >= 1 && y <= g.H - 2;
                v.x = (rowok && !st_first) ? v.x : 0u;
                v.y = rowok ? v.y : 0u;
                v.z = rowok ? v.z : 0u;
                v.w = (rowok && !st_last) ? v.w : 0u;
                __builtin_amdgcn_raw_buffer_store_b128(v, rout, st_lane + (y0 + 2 * i - ya) * pitch, 0, 2);
            }
        }
        gm_wait_lds();
        if (s + 2 <= n + 1) request(s + 2, b);
    }
#endif
}

// fp32 -> f16 bit pattern, round to nearest even (host side; positive finite values below 65504)
static unsigned short f16_bits(float f)
{
    u32 x;
    __builtin_memcpy(&x, &f, 4);
    const int e = (int)((x >> 23) & 0xff) - 127 + 15;
    u32 m = x & 0x7fffffu;
    if (e >= 31) return 0x7bff;
    if (e <= 0) {                         // subnormal half
        if (e < -10) return 0;
        m |= 0x800000u;
        const int shift = 14 - e;
        u32 hm = m >> shift;
        const u32 rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (unsigned short)hm;
    }
    u32 h = ((u32)e << 10) | (m >> 13);
    const u32 rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (unsigned short)h;
}
static float f16_value(unsigned short h)
{
    const int e = (h >> 10) & 31;
    const u32 m = h & 0x3ffu;
    if (e == 0) return (float)m * (1.0f / 16777216.0f);          // m * 2^-24
    u32 x = ((u32)(e - 15 + 127) << 23) | (m << 13);
    float f;
    __builtin_memcpy(&f, &x, 4);
    return f;
}

// f16 weights for gauss_mfma_kernel (NW = 17) and gauss_mfma_wide_kernel (NW = 31); false when the kernel does not apply
template <class GM, int NW>
static bool gauss_mfma_weights(int k, float sigma, GM* gm)
{
    const int R = k / 2;
    if (k % 2 == 0 || R < 1 || 2 * R + 1 > NW) return false;
    float w[MI_MAX_KERNEL_SIZE];
    gaussian_kernel_1d(k, sigma, w);
    double others = 0.0;
    for (int t = 0; t < NW; t++) gm->w[t] = gm->wl[t] = 0.f;
    for (int t = 0; t < k; t++) {
        if (t == R) continue;
        gm->w[t] = f16_value(f16_bits(w[t]));
        gm->wl[t] = f16_value(f16_bits(w[t] - gm->w[t]));
        others += (double)gm->w[t] + (double)gm->wl[t];
    }
    // centre weight: its remainder is the smallest f16 that makes the sum at least 1
    const double want = 1.0 - others;
    if (!(want > 0.0)) return false;
    gm->w[R] = f16_value(f16_bits((float)want));
    const double rest = want - (double)gm->w[R];
    unsigned short cb = f16_bits((float)fabs(rest));
    float cl = f16_value(cb);
    if (rest < 0.0) { if ((double)cl > -rest) cl = f16_value((unsigned short)(cb - 1)); cl = -cl; }
    else if ((double)cl < rest) cl = f16_value((unsigned short)(cb + 1));
    gm->wl[R] = cl;
    const double S = others + (double)gm->w[R] + (double)cl;
    gm->c1 = -0.25f;      // the row pass works in units of 512 levels: -128 / 512
    gm->c2 = (float)(128.0 * S + MI_GM_BIAS - 0.5);
    gm->rcp = 1.f; gm->d = 0.f;
    gm->R = R;
    return true;
}

// Geometry shared by both modes.  0: launched; 1: not applicable; < 0 or a HIP error otherwise
template <int MODE>
static int launch_mfma(GaussMfma& gm, hipStream_t s, const u32* in, u32* out, Geo g)
{
    const int W = g.W;
    // output blocks that intersect [y_begin, y_end): block b holds rows 16 b + 8 .. 16 b + 23
    const int ob_first = (g.y_begin + 8) / 16 - 1, ob_last = (g.y_end + 7) / 16 - 1;
    gm.ob_first = ob_first;
    gm.nob = ob_last - ob_first + 1;
    if (gm.nob <= 0) return 0;
    const bool narrow = tuning().gm_strip == 64, broad = tuning().gm_strip == 256;
    g.strips = narrow ? (W + 63) / 64 : broad ? (W + 255) / 256 : (W + 127) / 128;
    int bpb = tuning().rows_per_band > 0 ? std::max(1, tuning().rows_per_band / 16) : 4;   // 64-row bands measured best (16 x 4K and 32768^2)
    if (tuning().rows_per_band <= 0)
        while (bpb > 1 && (i64)g.batch * g.strips * ((gm.nob + bpb - 1) / bpb) < 4096) bpb >>= 1;
    // a band's rows are addressed through 32-bit buffer offsets
    while (bpb > 1 && (i64)(bpb + 2) * 16 * W * 4 > (1LL << 30)) bpb >>= 1;
    if ((i64)(bpb + 2) * 16 * W * 4 > (1LL << 30)) return 1;
    gm.bpb = bpb;
    g.rows_per_band = 16 * bpb;
    g.bands = (gm.nob + bpb - 1) / bpb;
    const i64 total = (i64)g.batch * g.bands * g.strips;
    if (total > 0x7fffffff) return fail(MI_ERR_UNSUPPORTED, "launch too large (%lld waves)", (long long)total);
    g.total_waves = (int)total;
    g.xcd_remap = 1;
    if (narrow) gauss_mfma_kernel<MODE, 4><<<(int)total, 64, 0, s>>>(in, out, g, gm);
    else if (broad) gauss_mfma_kernel<MODE, 16><<<(int)total, 64, 0, s>>>(in, out, g, gm);
    else gauss_mfma_kernel<MODE, 8><<<(int)total, 64, 0, s>>>(in, out, g, gm);
    return hip_status(hipGetLastError(), "matrix-core filter kernel launch");
}

// The wide form: output block b holds rows 16 b .. 16 b + 15
template <int MODE>
static int launch_mfma_wide(GaussMfmaWide& gm, hipStream_t s, const u32* in, u32* out, Geo g)
{
    const int W = g.W;
    const int ob_first = g.y_begin / 16, ob_last = (g.y_end - 1) / 16;
    gm.ob_first = ob_first;
    gm.nob = ob_last - ob_first + 1;
    if (gm.nob <= 0) return 0;
    g.strips = (W + 127) / 128;
    // measured per 16 x 4K (tools/ab_knob.py): the mean 0.2352 ms with 128-row bands against 0.2368 with 256, the Gaussian 0.2503 against 0.2453
    int bpb = tuning().rows_per_band > 0 ? std::max(1, tuning().rows_per_band / 16) : (MODE == 0 ? 16 : 8);
    if (tuning().rows_per_band <= 0)
        while (bpb > 1 && (i64)g.batch * g.strips * ((gm.nob + bpb - 1) / bpb) < 4096) bpb >>= 1;
    while (bpb > 1 && (i64)(bpb + 3) * 16 * W * 4 > (1LL << 30)) bpb >>= 1;
    if ((i64)(bpb + 3) * 16 * W * 4 > (1LL << 30)) return 1;
    gm.bpb = bpb;
    gm.serp = tuning().mfma_serpentine ? 1 : 0;
    g.rows_per_band = 16 * bpb;
    g.bands = (gm.nob + bpb - 1) / bpb;
    const i64 total = (i64)g.batch * g.bands * g.strips;
    if (total > 0x7fffffff) return fail(MI_ERR_UNSUPPORTED, "launch too large (%lld waves)", (long long)total);
    g.total_waves = (int)total;
    g.xcd_remap = 1;
    gauss_mfma_wide_kernel<MODE><<<(int)total, 64, 0, s>>>(in, out, g, gm);
    return hip_status(hipGetLastError(), "matrix-core filter kernel launch");
}

int launch_gauss_mfma(int k, float sigma, hipStream_t s, const u32* in, u32* out, Geo g)
{
    if (k > 17) {
        GaussMfmaWide gw;
        if (!gauss_mfma_weights<GaussMfmaWide, 31>(k, sigma, &gw)) return 1;
        return launch_mfma_wide<0>(gw, s, in, out, g);
    }
    GaussMfma gm;
    if (!gauss_mfma_weights<GaussMfma, 17>(k, sigma, &gm)) return 1;
    return launch_mfma<0>(gm, s, in, out, g);
}

// Mean blur k x k, 5 <= k <= 31, exact.  rcp and d: nearest(sum * rcp + d) == sum / (k * k) for every sum a window of
// bytes can reach, verified here once per size with the device's own operations (one fused multiply-add in fp32,
// round to nearest even); 1 when no such pair is found (the caller falls back).
int launch_mean_mfma(int k, hipStream_t s, const u32* in, u32* out, Geo g)
{
    const int R = k / 2;
    if (k % 2 == 0 || R < 1 || R > 15) return 1;
    static std::mutex mu;
    static int state[32];            // 0 unknown, 1 verified, -1 failed
    const float kk = (float)(k * k);
    const float unit = 1.0f / 131072.0f;      // the column pass leaves sum * 2^-17
    const float rcp = 1.0f / (kk * unit), d = 0.5f / kk - 0.5f;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (state[k] == 0) {
            bool ok = true;
            for (int sum = 0; sum <= 255 * k * k && ok; sum++) {
                const float f = __builtin_fmaf((float)sum * unit, rcp, d);
                ok = (int)nearbyintf(f) == sum / (k * k);
            }
            state[k] = ok ? 1 : -1;
        }
        if (state[k] < 0) return 1;
    }
    if (k > 17) {
        GaussMfmaWide gw;
        for (int t = 0; t < 31; t++) { gw.w[t] = t < k ? 1.0f : 0.f; gw.wl[t] = t < k ? 256.0f : 0.f; }
        gw.c1 = 16384.0f; gw.c2 = 0.f; gw.rcp = rcp; gw.d = d; gw.R = R;
        return launch_mfma_wide<1>(gw, s, in, out, g);
    }
    GaussMfma gm;
    for (int t = 0; t < 17; t++) { gm.w[t] = t < k ? 1.0f : 0.f; gm.wl[t] = t < k ? 256.0f : 0.f; }
    gm.c1 = 16384.0f;      // 2^14, whose last place is the row pass's unit 2^-9: the row sum is the low 16 bits of the accumulator
    gm.c2 = 0.f;
    gm.rcp = rcp; gm.d = d;
    gm.R = R;
    return launch_mfma<1>(gm, s, in, out, g);
}

}  // namespace mi
