// spectral.hip -- the pointwise kernels of the reference's FFT demo path
// (imageFFTImageGenerate, kernels/imageTools.cu:633-791) and the normalise helpers (:416-471).
// The 2-D transform itself lives in fft.hip.
#include <mutex>
#include <vector>

#include "common.h"
#include "internal.h"

namespace mi {

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

static int flat_grid(i64 n, int per_thread = 1)
{
    const i64 blocks = (n + 256LL * per_thread - 1) / (256LL * per_thread);
    return (int)std::max<i64>(1, std::min<i64>(blocks, 256 * 16));
}

// uchar4ToCufftComplex, imageTools.cu:506-515: plane_c[idx] = (float(c), 0).
__global__ __launch_bounds__(256) void u8_to_complex_kernel(const u32* __restrict__ in,
                                                            f32x2v* __restrict__ r,
                                                            f32x2v* __restrict__ g,
                                                            f32x2v* __restrict__ b, i64 n)
{
    // blockIdx.y = image of a batch: images back to back, and so are the plane triples (r, g, b of image k
    // at 3 k n, (3 k + 1) n, (3 k + 2) n from `r` when g = r + n, b = r + 2 n)
    in += (i64)blockIdx.y * n;
    r += (i64)blockIdx.y * 3 * n; g += (i64)blockIdx.y * 3 * n; b += (i64)blockIdx.y * 3 * n;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u32 p = in[i];
        r[i] = f32x2v{(float)(p & 0xffu), 0.0f};
        g[i] = f32x2v{(float)((p >> 8) & 0xffu), 0.0f};
        b[i] = f32x2v{(float)((p >> 16) & 0xffu), 0.0f};
    }
}

// ComplexRGBToUchar, imageTools.cu:517-527; the undefined float->uchar cast of out-of-range
// values is defined as saturate (NaN -> 0).
MI_DEV u32 sat_u8(float v)
{
    if (!(v > 0.0f)) return 0u;
    if (v >= 255.0f) return 255u;
    return (u32)v;
}

__global__ __launch_bounds__(256) void complex_to_u8_kernel(const f32x2v* __restrict__ r,
                                                            const f32x2v* __restrict__ g,
                                                            const f32x2v* __restrict__ b,
                                                            u32* __restrict__ out, i64 n)
{
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = sat_u8(r[i].x) | (sat_u8(g[i].x) << 8) | (sat_u8(b[i].x) << 16) | 0xFF000000u;
}

// fftShift, imageTools.cu:564-583: swap quadrants 1<->3 and 2<->4 for x<W/2, y<H/2.
__global__ __launch_bounds__(256) void fftshift_kernel(f32x2v* __restrict__ d, int w, int h)
{
    const int hw = w / 2, hh = h / 2;
    const i64 n = (i64)hw * hh, stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / hw), x = (int)(i % hw);
        const i64 i1 = (i64)y * w + x, i2 = (i64)(y + hh) * w + (x + hw);
        const i64 i3 = (i64)y * w + (x + hw), i4 = (i64)(y + hh) * w + x;
        const f32x2v a = d[i1], b = d[i2], c = d[i3], e = d[i4];
        d[i1] = b; d[i2] = a; d[i3] = e; d[i4] = c;
    }
}

// highpassFilter, imageTools.cu:530-549.
__global__ __launch_bounds__(256) void highpass_kernel(f32x2v* __restrict__ d, int w, int h,
                                                       float cutoff)
{
    const i64 n = (i64)w * h, stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / w), x = (int)(i % w);
        const float cx = (float)x - (float)w / 2.0f;
        const float cy = (float)y - (float)h / 2.0f;
        const float distance = sqrtf(cx * cx + cy * cy);
        const float gain = 1.0f - expf(-(distance * distance) / (2.0f * cutoff * cutoff));
        f32x2v v = d[i];
        v.x *= gain; v.y *= gain;
        d[i] = v;
    }
}

// fftShift -> highpassFilter -> fftShift (imageTools.cu:704-737) as one pass over `planes` back-to-back
// planes: the shift is an involution that swaps pairs of elements, so the three steps multiply the
// element at p by the gain evaluated at the position p is swapped with (at p itself for the middle
// row / column of an odd size, which the shift leaves alone).  Same float expression for the gain,
// one multiplication per component: bit-identical to the three kernels.
__global__ __launch_bounds__(256) void highpass_unshifted_kernel(f32x2v* __restrict__ d, int w, int h,
                                                                 float cutoff, i64 total)
{
    const int hw = w / 2, hh = h / 2;
    const i64 plane = (i64)w * h, stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const i64 p = i % plane;
        int y = (int)(p / w), x = (int)(p % w);
        if (x < 2 * hw && y < 2 * hh) {
            x = x < hw ? x + hw : x - hw;
            y = y < hh ? y + hh : y - hh;
        }
        const float cx = (float)x - (float)w / 2.0f;
        const float cy = (float)y - (float)h / 2.0f;
        const float distance = sqrtf(cx * cx + cy * cy);
        const float gain = 1.0f - expf(-(distance * distance) / (2.0f * cutoff * cutoff));
        f32x2v v = d[i];
        v.x *= gain; v.y *= gain;
        d[i] = v;
    }
}

// NormalizeAndZeroImaginary + ComplexRGBToUchar (imageTools.cu:551-562, :517-527) in one pass.
__global__ __launch_bounds__(256) void complex_to_u8_normalized_kernel(const f32x2v* __restrict__ r,
                                                                       const f32x2v* __restrict__ g,
                                                                       const f32x2v* __restrict__ b,
                                                                       u32* __restrict__ out, i64 n, float div)
{
    out += (i64)blockIdx.y * n;        // blockIdx.y = image of a batch (see u8_to_complex_kernel)
    r += (i64)blockIdx.y * 3 * n; g += (i64)blockIdx.y * 3 * n; b += (i64)blockIdx.y * 3 * n;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = sat_u8(r[i].x / div) | (sat_u8(g[i].x / div) << 8) | (sat_u8(b[i].x / div) << 16) | 0xFF000000u;
}

// NormalizeAndZeroImaginary, imageTools.cu:551-562: re /= (float)(W*H), im = 0.
__global__ __launch_bounds__(256) void normalize_zero_imag_kernel(f32x2v* __restrict__ d, i64 n,
                                                                  float div)
{
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        d[i] = f32x2v{d[i].x / div, 0.0f};
}

// findMinMax, imageTools.cu:416-454: global min/max of a float image.  Wave reduce by DPP-free
// shuffles, one block result in LDS, then one ordered-int atomic per block: for IEEE floats the
// signed-int order of the bit patterns matches the float order for non-negative values and is
// reversed for negative ones, so min/max are taken on a monotone integer key.  NaNs are skipped
// (the reference's CAS loop would spin forever on NaN).
MI_DEV int f2key(float f)
{
    const int i = __float_as_int(f);
    return i >= 0 ? i : (int)(0x80000000u - (u32)i);
}
MI_DEV float key2f(int k)
{
    return __int_as_float(k >= 0 ? k : (int)(0x80000000u - (u32)k));
}

__global__ void minmax_init_kernel(int* keys)
{
    keys[0] = 0x7fffffff;
    keys[1] = (int)0x80000000;
}

// Each lane reads 16-byte groups (the ragged head and tail of an unaligned buffer go to lane 0
// of block 0), wave reduction by DPP-free shuffles, one pair of atomics per block.
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ img, i64 n,
                                                     int* __restrict__ keys)
{
    float mn = __builtin_inff(), mx = -__builtin_inff();
    auto take = [&](float v) { if (v == v) { mn = fminf(mn, v); mx = fmaxf(mx, v); } };
    // elements before the first 16-byte boundary
    i64 head = (i64)((16 - ((uintptr_t)img & 15)) & 15) / 4;
    if (head > n) head = n;
    const i64 groups = (n - head) / 4;
    const f32x4v* g4 = reinterpret_cast<const f32x4v*>(img + head);
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += stride) {
        const f32x4v v = g4[i];
        take(v.x); take(v.y); take(v.z); take(v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (i64 i = 0; i < head; i++) take(img[i]);
        for (i64 i = head + 4 * groups; i < n; i++) take(img[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    __shared__ float smn[4], smx[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smn[wave] = mn; smx[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) { mn = fminf(mn, smn[k]); mx = fmaxf(mx, smx[k]); }
        if (mn <= mx) {
            atomicMin(&keys[0], f2key(mn));
            atomicMax(&keys[1], f2key(mx));
        }
    }
}

__global__ void minmax_final_kernel(const int* keys, float* out)
{
    out[0] = key2f(keys[0]);
    out[1] = key2f(keys[1]);
}

// floatToUchar4, imageTools.cu:458-471, after the scaling the reference never wrote:
// v' = (v-min)/(max-min)*255 (0 when max==min), saturating truncation, out=(v',v',v',255).
__global__ __launch_bounds__(256) void normalize_to_u8_kernel(const float* __restrict__ img,
                                                              u32* __restrict__ out, i64 n,
                                                              const float* __restrict__ minmax)
{
    const float mn = minmax[0], range = minmax[1] - minmax[0];
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = range > 0.0f ? (img[i] - mn) / range * 255.0f : 0.0f;
        out[i] = sat_u8(v) * 0x010101u | 0xFF000000u;
    }
}

int launch_u8_to_complex(const void* in, void* r, void* g, void* b, int w, int h, hipStream_t s, int batch)
{
    const i64 n = (i64)w * h;
    u8_to_complex_kernel<<<dim3(flat_grid(n), batch), 256, 0, s>>>((const u32*)in, (f32x2v*)r, (f32x2v*)g, (f32x2v*)b, n);
    return hip_status(hipGetLastError(), "u8_to_complex launch");
}

int launch_complex_to_u8(const void* r, const void* g, const void* b, void* out, int w, int h, hipStream_t s)
{
    const i64 n = (i64)w * h;
    complex_to_u8_kernel<<<flat_grid(n), 256, 0, s>>>((const f32x2v*)r, (const f32x2v*)g, (const f32x2v*)b, (u32*)out, n);
    return hip_status(hipGetLastError(), "complex_to_u8 launch");
}

int launch_fftshift(void* d, int w, int h, hipStream_t s)
{
    const i64 n = (i64)(w / 2) * (h / 2);
    if (n == 0) return MI_OK;
    fftshift_kernel<<<flat_grid(n), 256, 0, s>>>((f32x2v*)d, w, h);
    return hip_status(hipGetLastError(), "fftshift launch");
}

int launch_highpass(void* d, int w, int h, float cutoff, hipStream_t s)
{
    highpass_kernel<<<flat_grid((i64)w * h), 256, 0, s>>>((f32x2v*)d, w, h, cutoff);
    return hip_status(hipGetLastError(), "highpass launch");
}

int launch_highpass_unshifted(void* d, int w, int h, int planes, float cutoff, hipStream_t s)
{
    const i64 total = (i64)w * h * planes;
    highpass_unshifted_kernel<<<flat_grid(total), 256, 0, s>>>((f32x2v*)d, w, h, cutoff, total);
    return hip_status(hipGetLastError(), "highpass launch");
}

int launch_complex_to_u8_normalized(const void* r, const void* g, const void* b, void* out, int w, int h, int batch, hipStream_t s)
{
    const i64 n = (i64)w * h;
    complex_to_u8_normalized_kernel<<<dim3(flat_grid(n), batch), 256, 0, s>>>((const f32x2v*)r, (const f32x2v*)g, (const f32x2v*)b,
                                                                 (u32*)out, n, (float)(w * h));
    return hip_status(hipGetLastError(), "complex_to_u8 launch");
}

int launch_normalize_zero_imag(void* d, int w, int h, hipStream_t s)
{
    normalize_zero_imag_kernel<<<flat_grid((i64)w * h), 256, 0, s>>>((f32x2v*)d, (i64)w * h, (float)(w * h));
    return hip_status(hipGetLastError(), "normalize_zero_imag launch");
}

// The {min,max} pair is reduced in place in the caller's 2-float result buffer, reinterpreted
// as two ordered-int keys and converted back by the last kernel: no allocation.
int launch_minmax(const void* img, size_t n, void* minmax, hipStream_t s)
{
    minmax_init_kernel<<<1, 1, 0, s>>>((int*)minmax);
    // few blocks: every block ends with two atomics on the same two words
    minmax_kernel<<<std::min(flat_grid((i64)n, 16), 512), 256, 0, s>>>((const float*)img, (i64)n, (int*)minmax);
    minmax_final_kernel<<<1, 1, 0, s>>>((const int*)minmax, (float*)minmax);
    return hip_status(hipGetLastError(), "minmax launch");
}

int launch_normalize_to_u8(const void* img, void* out, size_t n, const void* minmax, hipStream_t s)
{
    normalize_to_u8_kernel<<<flat_grid((i64)n), 256, 0, s>>>((const float*)img, (u32*)out, (i64)n, (const float*)minmax);
    return hip_status(hipGetLastError(), "normalize_to_u8 launch");
}


// ---------------------------------------------------------------- compressImage support
// Keep the `keep` coefficients of largest |X|^2 of a spectrum and zero the rest: the threshold
// is the keep-th largest value, found exactly by a three-level radix select on the bit pattern of
// |X|^2 (non-negative floats order like unsigned integers; bits 30..20, 19..9, 8..0): each level
// histograms, in LDS per block, the next digit of the values that share the digits found so far.
// Coefficients equal to the threshold are all kept, so slightly more than `keep` can survive
// when values tie.
MI_DEV u32 mag2_bits(f32x2v v) { return __float_as_uint(v.x * v.x + v.y * v.y); }

constexpr int SEL_BINS = 2048;
constexpr int SEL_MAX_PLANES = 4;
struct SelPrefix { u32 v[SEL_MAX_PLANES]; };

// blockIdx.y = plane (planes are back to back, n values each).
// hist[plane][digit] += 1 for every value of the plane whose bits above `shift + width` equal the plane's prefix
__global__ __launch_bounds__(256) void mag_hist_kernel(const f32x2v* __restrict__ d, i64 n, SelPrefix prefix,
                                                       int prefix_shift, int shift, u32 digit_mask,
                                                       u32* __restrict__ hist)
{
    __shared__ u32 local[SEL_BINS];
    for (int b = threadIdx.x; b < SEL_BINS; b += 256) local[b] = 0;
    __syncthreads();
    const int plane = blockIdx.y;
    const u32 want = prefix.v[plane];
    d += (i64)plane * n;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u32 b = mag2_bits(d[i]);
        if (prefix_shift >= 32 || (b >> prefix_shift) == want) atomicAdd(&local[(b >> shift) & digit_mask], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < SEL_BINS; b += 256)
        if (local[b]) atomicAdd(&hist[plane * SEL_BINS + b], local[b]);
}

__global__ __launch_bounds__(256) void mag_mask_kernel(f32x2v* __restrict__ d, i64 n, SelPrefix threshold)
{
    const u32 thr = threshold.v[blockIdx.y];
    d += (i64)blockIdx.y * n;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (mag2_bits(d[i]) < thr) d[i] = f32x2v{0.f, 0.f};
}

// The histograms live in a static device array (32 KiB per GPU), so the selection allocates
// nothing and never pays hipFree's device-wide synchronisation.  Calls on one device take turns
// on it (SelHist holds that device's mutex for the length of the call).
static __device__ __align__(256) u32 g_sel_hist[SEL_MAX_PLANES * SEL_BINS];

struct SelHist {
    std::unique_lock<std::mutex> lock;
    u32* ptr = nullptr;
    int acquire()
    {
        static std::mutex mu[64];
        static u32* dev_ptr[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return hip_status(e, "hipGetDevice");
        if (dev < 0 || dev >= 64) return fail(MI_ERR_UNSUPPORTED, "device ordinal %d out of range", dev);
        lock = std::unique_lock<std::mutex>(mu[dev]);
        if (!dev_ptr[dev]) {
            void* p = nullptr;
            e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_sel_hist));
            if (e != hipSuccess) return hip_status(e, "hipGetSymbolAddress(g_sel_hist)");
            dev_ptr[dev] = (u32*)p;
        }
        ptr = dev_ptr[dev];
        return MI_OK;
    }
};

// `planes` spectra of n values each, back to back, each keeping its own `keep` largest.
// Blocking: reads three histograms per plane back on the host (one read-back per level for all planes).
int spectrum_keep_top(void* data, size_t n, size_t keep, int planes, hipStream_t s)
{
    if (keep >= n || planes <= 0) return MI_OK;
    int rc = MI_OK;
    SelHist sel;
    if (keep != 0 && (rc = sel.acquire()) != MI_OK) return rc;
    for (int p0 = 0; p0 < planes && rc == MI_OK; p0 += SEL_MAX_PLANES) {
        const int np = std::min(SEL_MAX_PLANES, planes - p0);
        f32x2v* base = (f32x2v*)data + (size_t)p0 * n;
        const dim3 grid(std::min(flat_grid((i64)n, 4), 1024), np);
        SelPrefix prefix{};
        if (keep == 0) {
            for (int p = 0; p < np; p++) prefix.v[p] = 0xffffffffu;        // nothing survives
            mag_mask_kernel<<<grid, 256, 0, s>>>(base, (i64)n, prefix);
            rc = hip_status(hipGetLastError(), "spectrum_keep_top");
            continue;
        }
        u32* d_hist = sel.ptr;
        hipError_t e = hipSuccess;
        std::vector<u32> hist((size_t)np * SEL_BINS);
        // digits, most significant first: (shift, width)
        const int shifts[3] = {20, 9, 0}, widths[3] = {11, 11, 9};
        size_t want[SEL_MAX_PLANES];     // rank of the threshold among the values sharing the prefix, from the top
        for (int p = 0; p < np; p++) want[p] = keep;
        for (int level = 0; level < 3 && e == hipSuccess; level++) {
            const int shift = shifts[level], prefix_shift = level == 0 ? 32 : shifts[level - 1];
            e = hipMemsetAsync(d_hist, 0, hist.size() * sizeof(u32), s);
            if (e != hipSuccess) break;
            mag_hist_kernel<<<grid, 256, 0, s>>>(base, (i64)n, prefix, prefix_shift, shift, (1u << widths[level]) - 1u, d_hist);
            e = hipMemcpyAsync(hist.data(), d_hist, hist.size() * sizeof(u32), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) break;
            for (int p = 0; p < np; p++) {
                const u32* h = hist.data() + (size_t)p * SEL_BINS;
                size_t above = 0;
                u32 digit = 0;
                for (int b = (1 << widths[level]) - 1; b >= 0; b--) {
                    if (above + h[b] >= want[p]) { digit = (u32)b; want[p] -= above; break; }
                    above += h[b];
                }
                prefix.v[p] = (prefix.v[p] << widths[level]) | digit;
            }
        }
        if (e == hipSuccess) {
            mag_mask_kernel<<<grid, 256, 0, s>>>(base, (i64)n, prefix);   // after three levels the prefix is the threshold
            e = hipGetLastError();
        }
        rc = hip_status(e, "spectrum_keep_top");
    }
    return rc;
}

}  // namespace mi
