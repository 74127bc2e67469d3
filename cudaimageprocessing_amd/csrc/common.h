// common.h -- shared device helpers for the gfx950 image-filter kernels.
//
// Execution scheme of the 2-D stencil kernels ("wave march"): one 64-lane wave owns a column
// strip of 64*4 pixels and marches down a band of rows.  Every lane holds 4 adjacent pixels
// (one 16-byte load per row), keeps the rows of the stencil window in registers, and gets its
// left/right neighbours' pixels from the adjacent lanes with DPP wave shifts -- no LDS, no
// barriers.  All 64 lanes produce output (a wave-row is one aligned 1-KiB store); the pixels
// beyond the strip's edges come from a second, two-lane "halo" load per row whose value enters
// the DPP shift as the edge lanes' own operand.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int u32;
typedef long long i64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x3 __attribute__((ext_vector_type(3)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 __attribute__((aligned(4))) u32x4_u;  // a 16-byte vector at 4-byte alignment
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

#define MI_DEV __device__ __forceinline__

// Geometry of one launch.  Coordinates are GLOBAL image coordinates: a strip launch passes a
// buffer that holds only some rows of the image (in_lo/in_hi/in_origin describe which).
struct Geo {
    int W, H;             // image size
    int batch;            // frames
    int y_begin, y_end;   // output rows [y_begin, y_end)
    int rows_per_band;    // rows one wave marches over
    int bands, strips;    // per frame
    int total_waves;      // batch * bands * strips
    int xcd_remap;        // 1: give each XCD a contiguous run of blocks
    i64 N;                // W*H
    i64 in_lo, in_hi;     // flat pixel indices [in_lo, in_hi) present in `in`
    i64 in_origin;        // flat index of in[0]
    i64 out_origin;       // flat index of out[0]
    i64 in_frame_stride;  // pixels between frames
    i64 out_frame_stride;
};

// lane i <- lane i-1; lane 0, which has no left neighbour in the wave, keeps its own `edge`
MI_DEV u32 from_left(u32 v, u32 edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false); }
// lane i <- lane i+1; lane 63 keeps its own `edge`
MI_DEV u32 from_right(u32 v, u32 edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false); }
// the same with 0 for the lane without a neighbour (bound_ctrl: no register to initialise)
MI_DEV u32 from_left(u32 v) { return __builtin_amdgcn_update_dpp(0u, v, 0x138, 0xf, 0xf, true); }
MI_DEV u32 from_right(u32 v) { return __builtin_amdgcn_update_dpp(0u, v, 0x130, 0xf, 0xf, true); }

MI_DEV u32 pk_min(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a),
                                                              __builtin_bit_cast(u16x2, b)));
}
MI_DEV u32 pk_max(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a),
                                                              __builtin_bit_cast(u16x2, b)));
}
MI_DEV u32 pk_add(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));
}
MI_DEV u32 pk_sub(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b));
}

// Four pixels at flat index f..f+3 of the global image; pixels outside the image (or outside
// the rows present in `in`) read 0.  ALIGNED: every f passed is a multiple of 4 and `in` is
// 16-byte aligned, so the fast path is one aligned dwordx4 load.
template <int POL>
MI_DEV u32x4 load_px4(const u32* __restrict__ in, const Geo& g, i64 f)
{
    constexpr bool ALIGNED = POL & 1;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (f >= g.in_lo && f + 4 <= g.in_hi) {
        const u32* p = in + (f - g.in_origin);
        if (ALIGNED) v = *reinterpret_cast<const u32x4*>(p);
        else v = *reinterpret_cast<const u32x4_u*>(p);
    } else if (f + 4 > g.in_lo && f < g.in_hi) {
        const u32* p = in + (f - g.in_origin);
        if (f + 0 >= g.in_lo && f + 0 < g.in_hi) v.x = p[0];
        if (f + 1 >= g.in_lo && f + 1 < g.in_hi) v.y = p[1];
        if (f + 2 >= g.in_lo && f + 2 < g.in_hi) v.z = p[2];
        if (f + 3 >= g.in_lo && f + 3 < g.in_hi) v.w = p[3];
    }
    return v;
}

// Store four output pixels of row y at columns cx..cx+3 (columns >= W are dropped).
template <int POL>
MI_DEV void store_px4(u32* __restrict__ out, const Geo& g, int y, int cx, u32x4 v)
{
    constexpr bool ALIGNED = POL & 1;
    u32* p = out + ((i64)y * g.W + cx - g.out_origin);
    if (ALIGNED) {   // W%4==0 and cx%4==0: cx<W implies cx+4<=W
        // streaming store: the output is not read again by this kernel, keep the caches for
        // the input rows that neighbouring waves share
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    } else if (cx + 4 <= g.W) {
        *reinterpret_cast<u32x4_u*>(p) = v;
    } else {
        if (cx + 0 < g.W) p[0] = v.x;
        if (cx + 1 < g.W) p[1] = v.y;
        if (cx + 2 < g.W) p[2] = v.z;
    }
}

// Work assignment of a wave.  Blocks are dealt round-robin over the 8 XCDs (blocks b and b+8
// share an L2), so logical block ids are remapped to give each XCD a contiguous run of
// strips/bands: neighbouring waves, which re-read each other's halo rows and margin columns,
// then hit in the same L2.  The remap is a bijection for every grid size.
struct WaveJob {
    int frame, band, strip, lane;
    bool valid;
};

MI_DEV WaveJob wave_job(const Geo& g)
{
    const int waves_per_block = blockDim.x >> 6;
    const u32 nblk = gridDim.x, b = blockIdx.x;
    const u32 q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    const u32 logical = g.xcd_remap ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx : b;
    int wid = (int)logical * waves_per_block + (int)(threadIdx.x >> 6);
    wid = __builtin_amdgcn_readfirstlane(wid);
    WaveJob j;
    j.valid = wid < g.total_waves;
    j.strip = wid % g.strips;
    const int t = wid / g.strips;
    j.band = t % g.bands;
    j.frame = t / g.bands;
    j.lane = threadIdx.x & 63;
    return j;
}

// x-mask helper: 0xFFFFFFFF where column x lies in the reference's write set [1, W-2].
MI_DEV u32 xmask(int x, int W) { return (x >= 1 && x <= W - 2) ? 0xFFFFFFFFu : 0u; }

// Rows of a wave's band in marching order.  Even bands march down, odd bands march up, so two
// vertically adjacent bands reach their common boundary at the same time (both at the start
// or both at the end of their march): the halo rows they both need are then fetched from HBM
// once and hit in the XCD's L2 for the second wave.  step i in [0, n) is output row row(i);
// the stencil needs input rows row(i-R) .. row(i+R).
struct March {
    int ys, dir, n;
    MI_DEV int row(int i) const { return ys + dir * i; }
};

MI_DEV March make_march(const Geo& g, const WaveJob& j)
{
    const int y0 = g.y_begin + j.band * g.rows_per_band;
    const int y1 = min(y0 + g.rows_per_band, g.y_end);
    March m;
    m.n = y1 - y0;
    m.dir = (j.band & 1) ? -1 : 1;
    m.ys = (j.band & 1) ? y1 - 1 : y0;
    return m;
}

// Row I/O of one wave.  POL 0 (any width/alignment): per-lane 64-bit addresses with explicit
// range checks (load_px4/store_px4).  POL 1 (width % 4 == 0, 16-byte-aligned buffers): buffer
// instructions through two wave-uniform resource descriptors.  The hardware range check then
// does the work of the reference-UB conventions for free -- a tap whose flat index lies outside
// the image (or outside the rows present) reads 0 -- and of store predication: a lane that must
// not write is given an out-of-range offset.  With no branches around the memory operations the
// compiler counts them exactly (s_waitcnt vmcnt(1) in the row loop), so a wave never waits for
// its own stores.  Everything is a multiple of 4 pixels in this mode, so a 16-byte access is
// never partly in range.
template <int POL>
struct RowIO;

template <>
struct RowIO<0> {
    const u32* in;
    u32* out;
    const Geo* g;
    int cx, lane;
    bool productive;
    // margin: lanes at each edge of the wave that only feed their neighbours (overlapping strips)
    MI_DEV RowIO(const u32* in_, u32* out_, const Geo& g_, const WaveJob& j, const March&, int cx0, int margin = 0)
        : in(in_ + (i64)j.frame * g_.in_frame_stride), out(out_ + (i64)j.frame * g_.out_frame_stride), g(&g_)
    {
        lane = j.lane;
        cx = cx0 + 4 * j.lane;
        productive = cx < g_.W && j.lane >= margin && j.lane < 64 - margin;
    }
    MI_DEV u32x4 load(int row) const { return load_px4<0>(in, *g, (i64)row * g->W + cx); }
    // the 4 pixels `hop` groups left of the strip (lane 0) / right of it (lane 63); 0 elsewhere
    MI_DEV u32x4 load_halo(int row, int hop) const
    {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (lane == 0) v = load_px4<0>(in, *g, (i64)row * g->W + cx - 4 * hop);
        else if (lane == 63) v = load_px4<0>(in, *g, (i64)row * g->W + cx + 4 * hop);
        return v;
    }
    template <int NPX>
    MI_DEV u32x4 load_halo_px(int row, int hop) const
    {
        const u32x4 h = load_halo(row, hop);
        if (lane != 0 || NPX == 4) return h;
        const u32 a[4] = {h.x, h.y, h.z, h.w};
        u32x4 v = {0u, 0u, 0u, 0u};
        v.x = a[4 - NPX];
        if (NPX > 1) v.y = a[(5 - NPX) & 3];
        if (NPX > 2) v.z = a[(6 - NPX) & 3];
        return v;
    }
    // the pixel just left of the strip (lane 0) / just right of it (lane 63); 0 elsewhere
    MI_DEV u32 load_halo1(int row) const
    {
        const u32x4 h = load_halo(row, 1);
        return lane == 0 ? h.w : h.x;
    }
    MI_DEV void store(int y, u32x4 v) const { if (productive) store_px4<0>(out, *g, y, cx, v); }
    MI_DEV void store_in_band(int y, u32x4 v) const { store(y, v); }
};

template <>
struct RowIO<1> {
    __amdgpu_buffer_rsrc_t rin, rout;
    int in_off;    // byte offset of this lane's 4 pixels in row `row0`, relative to rin's base
    int halo_off;  // the same for the halo group one hop away (out of range for lanes 1..62)
    int halo_dir;  // -16 for lane 0, +16 for lane 63: byte step per extra hop
    int halo1_off; // the single pixel next to the strip: in_off - 4 (lane 0), in_off + 16 (lane 63)
    bool first_lane;
    u32 out_off;   // same for rout; 0x80000000 (out of range) for lanes that must not store
    int row0, ylo, pitch;
    MI_DEV RowIO(const u32* in_, u32* out_, const Geo& g, const WaveJob& j, const March& m, int cx0, int margin = 0)
    {
        const u32* in = in_ + (i64)j.frame * g.in_frame_stride;
        u32* out = out_ + (i64)j.frame * g.out_frame_stride;
        ylo = m.dir > 0 ? m.ys : m.ys - (m.n - 1);
        row0 = ylo - 16;                       // any row at or above the first one loaded
        pitch = g.W * 4;
        const i64 fmin = (i64)row0 * g.W + cx0 - 8;      // lowest flat index this wave can touch
        const i64 base = fmin > g.in_lo ? fmin : g.in_lo;
        i64 bytes = (g.in_hi - base) * 4;
        bytes = bytes < 0 ? 0 : (bytes > 0x7FFFFF00LL ? 0x7FFFFF00LL : bytes);
        rin = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (base - g.in_origin)), 0, (int)bytes, 0x00020000);
        in_off = (int)((fmin + 8 - base) * 4) + 16 * j.lane;
        halo_dir = j.lane == 0 ? -16 : 16;
        first_lane = j.lane == 0;
        halo_off = (j.lane == 0 || j.lane == 63) ? in_off + halo_dir : (int)0x80000000;
        halo1_off = j.lane == 0 ? in_off - 4 : (j.lane == 63 ? in_off + 16 : (int)0x80000000);
        const i64 obase = (i64)ylo * g.W;
        rout = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (obase - g.out_origin)), 0, m.n * pitch, 0x00020000);
        const int cx = cx0 + 4 * j.lane;
        out_off = (cx < g.W && j.lane >= margin && j.lane < 64 - margin) ? (u32)(cx * 4) : 0x80000000u;
    }
    MI_DEV u32x4 load(int row) const
    {
        return __builtin_amdgcn_raw_buffer_load_b128(rin, (row - row0) * pitch + in_off, 0, 0);
    }
    MI_DEV u32x4 load_halo(int row, int hop) const
    {
        return __builtin_amdgcn_raw_buffer_load_b128(rin, (row - row0) * pitch + halo_off + (hop - 1) * halo_dir, 0, 0);
    }
    // The NPX pixels of halo group `hop` nearest to the strip: lane 0 gets the group's last NPX
    // pixels, lane 63 its first NPX, in elements 0..NPX-1 (the rest is 0).  Loading exactly what
    // is used matters: unused destination registers of a wider load get reused for temporaries
    // and the wave then waits for the load right after issuing it.
    template <int NPX>
    MI_DEV u32x4 load_halo_px(int row, int hop) const
    {
        const int off = (row - row0) * pitch + halo_off + (hop - 1) * halo_dir + (first_lane ? (4 - NPX) * 4 : 0);
        u32x4 v = {0u, 0u, 0u, 0u};
        if constexpr (NPX == 4) {
            v = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
        } else if constexpr (NPX == 3) {
            const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rin, off, 0, 0);
            v.x = t.x; v.y = t.y; v.z = t.z;
        } else if constexpr (NPX == 2) {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rin, off, 0, 0);
            v.x = t.x; v.y = t.y;
        } else {
            v.x = __builtin_amdgcn_raw_buffer_load_b32(rin, off, 0, 0);
        }
        return v;
    }
    // One-pixel halo as a single dword load: a 3x3 filter uses nothing else of the halo group, and
    // unused destination registers of a wider load would be reused by the compiler for
    // temporaries, which makes the wave wait for the load right after issuing it.
    MI_DEV u32 load_halo1(int row) const
    {
        return __builtin_amdgcn_raw_buffer_load_b32(rin, (row - row0) * pitch + halo1_off, 0, 0);
    }
    MI_DEV void store(int y, u32x4 v) const
    {   // aux 2 = nt: streaming store, the output is not read again by this kernel
        __builtin_amdgcn_raw_buffer_store_b128(v, rout, (u32)((y - ylo) * pitch) + out_off, 0, 2);
    }
    // The same with the row's offset in the instruction's scalar offset (no vector add per row).  The
    // hardware leaves the scalar offset out of its range check: only for rows INSIDE the band (a lane
    // that must not store is still dropped through its out-of-range out_off).
    MI_DEV void store_in_band(int y, u32x4 v) const
    {
        __builtin_amdgcn_raw_buffer_store_b128(v, rout, out_off, (y - ylo) * pitch, 2);
    }
};

// ---------------------------------------------------------------------------------------------
// RowFeed: the input rows of a marching wave, requested in BURSTS.
//
// A wave that asks for one row per step presents the memory system with thousands of slow, widely
// spaced row streams; asking for K consecutive rows of the strip at one time runs 3..10 % faster as
// a pure copy (tools/marchbench.hip, profiles/r02_marchbench.txt).  For aligned images (POL 1) the
// rows go straight into a per-wave ring of 2K rows in LDS (buffer_load ... lds: rows in flight hold
// no registers, so occupancy is unchanged) and come back one per step with a ds_read; every K steps
// the wave requests the rows of the NEXT K steps.  hipcc does not order a ds_read behind an LDS-DMA
// write, so the wait is explicit.  Vector-memory operations complete in order: after a burst of L
// loads is issued, the only operations younger than the PREVIOUS burst are the stores of the K
// steps in between (one per step once the kernel has primed, none before) and the L loads just
// issued, so s_waitcnt vmcnt(K + L) -- vmcnt(L) while fewer than K stores can lie in between --
// means "the rows of the next K steps have landed".  Waiting for more than necessary is always safe.
//
// HALO: what a row carries beside the strip's own pixels, in a second, 256-byte part of the ring slot,
// fetched by ONE more 4-byte-per-lane request: 1 = the one pixel next to the strip on either side
// (3x3 filters; a dword in each edge lane), 4 or 8 = the one or two 4-pixel groups on either side
// (lanes 0..HALO-1 fetch the left pixels, lanes HALO..2*HALO-1 the right ones; Row::near and Row::far
// then hold, in lane 0, the nearer and the farther group to the left, in lane 63 those to the right).
// Usage: feed.start(first_row_index) once, then feed.next(c, prime) for consumption c = 0, 1, 2, ...
// (rows first, first+1, ... in marching order); `prime` = consumptions before the kernel's first store.
// For POL 0 the feed degenerates to the one-row-ahead register prefetch.
// MI_ROWFEED_FULL_WAIT=1 (the guard build, libmi_imagetools_guard.so) replaces every counted wait by
// s_waitcnt vmcnt(0).  The counts above are only right while a step issues exactly the memory operations
// this comment lists; tests/test_gpu_guard.py compares the product library with the guard build bit for
// bit, so a compiler that adds or drops one (a spill, an elided store) is caught as a difference.
#ifndef MI_ROWFEED_FULL_WAIT
#define MI_ROWFEED_FULL_WAIT 0
#endif

template <int POL, int K_, int HALO>
struct RowFeed {
    static constexpr bool HALO1 = HALO != 0;
    static constexpr bool LDSQ = POL == 1 && K_ > 0;   // K_ = 0: the register prefetch for every policy
    static constexpr int K = K_ > 0 ? K_ : 1;
    static constexpr int SLOT_BYTES = HALO1 ? 1280 : 1024;
    static constexpr int WAVE_BYTES = 2 * K * SLOT_BYTES;
    static constexpr int BLOCK_BYTES = LDSQ ? 4 * WAVE_BYTES : 16;    // the __shared__ ring a 256-lane block declares
    static constexpr int L = HALO1 ? 2 * K : K;           // loads per burst
    struct Row { u32x4 p; u32 h; u32x4 near, far; };

    const RowIO<POL>& io;
    const March& mch;
    unsigned char* ring;       // this wave's 2K slots
    int first, lane;
    int phase = 0, slot = 0;
    int group_off = 0;         // HALO >= 4: this lane's pixel of the halo groups (out of range for lanes >= 2 * HALO)
    Row nxt{};                 // POL 0 only

    MI_DEV RowFeed(const RowIO<POL>& io_, const March& m_, unsigned char* block_ring, int lane_)
        : io(io_), mch(m_), ring(block_ring + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * WAVE_BYTES), first(0), lane(lane_)
    {
        if constexpr (LDSQ && HALO >= 4) {
            const int base0 = io.in_off - 16 * lane_;        // lane 0's pixels
            group_off = lane_ < HALO ? base0 - 4 * HALO + 4 * lane_
                                     : (lane_ < 2 * HALO ? base0 + 1024 + 4 * (lane_ - HALO) : (int)0x80000000);
        }
    }

    MI_DEV void request(int row, int s)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (LDSQ) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(io.rin, (__attribute__((address_space(3))) void*)(ring + s * SLOT_BYTES), 16,
                                                     (row - io.row0) * io.pitch + io.in_off, 0, 0, 0);
            if constexpr (HALO == 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(io.rin, (__attribute__((address_space(3))) void*)(ring + s * SLOT_BYTES + 1024), 4,
                                                         (row - io.row0) * io.pitch + io.halo1_off, 0, 0, 0);
            if constexpr (HALO >= 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(io.rin, (__attribute__((address_space(3))) void*)(ring + s * SLOT_BYTES + 1024), 4,
                                                         (row - io.row0) * io.pitch + group_off, 0, 0, 0);
        }
#endif
    }
    template <int N>
    static MI_DEV void wait_vm()
    {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
    }
    MI_DEV Row load_direct(int row) const
    {
        Row r{};
        r.p = io.load(row);
        if constexpr (HALO == 1) r.h = io.load_halo1(row);
        if constexpr (HALO >= 4) r.near = io.load_halo(row, 1);
        if constexpr (HALO >= 8) r.far = io.load_halo(row, 2);
        return r;
    }
    // requests rows first .. first+K-1 (marching order)
    MI_DEV void start(int first_index)
    {
        first = first_index;
        if constexpr (LDSQ) {
#pragma unroll
            for (int p = 0; p < K; p++) request(mch.row(first + p), p);
        } else {
            nxt = load_direct(mch.row(first));
        }
    }
    // the row of consumption c (= row index first + c); c must count 0, 1, 2, ...
    MI_DEV Row next(int c, int prime)
    {
        Row r{};
        if constexpr (LDSQ) {
            if (phase == 0) {
                const int half = slot >= K ? 0 : K;      // the half of the ring the last K consumptions have read
#pragma unroll
                for (int p = 0; p < K; p++) request(mch.row(first + c + K + p), half + p);
                if (MI_ROWFEED_FULL_WAIT) wait_vm<0>();
                else if (c < prime + K) wait_vm<L>();
                else wait_vm<K + L>();
            }
            const unsigned char* sl = ring + slot * SLOT_BYTES;
            r.p = *reinterpret_cast<const u32x4*>(sl + 16 * lane);
            r.h = HALO == 1 ? *reinterpret_cast<const u32*>(sl + 1024 + 4 * lane) : 0u;
            if constexpr (HALO >= 4) {
                // left pixels at bytes [0, 4 HALO), right ones at [4 HALO, 8 HALO); only lanes 0 and 63 use theirs
                const bool left = lane == 0;
                r.near = *reinterpret_cast<const u32x4*>(sl + 1024 + (left ? 4 * HALO - 16 : 4 * HALO));
                if constexpr (HALO >= 8) r.far = *reinterpret_cast<const u32x4*>(sl + 1024 + (left ? 0 : 4 * HALO + 16));
            }
            phase = phase + 1 == K ? 0 : phase + 1;
            slot = slot + 1 == 2 * K ? 0 : slot + 1;
        } else {
            r = nxt;
            nxt = load_direct(mch.row(first + c + 1));
        }
        return r;
    }
};

