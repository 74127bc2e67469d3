// api.hip -- the extern "C" surface of libmi_imagetools (include/mi_imagetools.h): argument
// validation, error reporting, the blocking host tier and the PNG writer.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <vector>

#include "internal.h"

namespace mi {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int hip_status(hipError_t e, const char* what)
{
    if (e == hipSuccess) return MI_OK;
    snprintf(g_err, sizeof g_err, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return (int)e;
}

// ---------------------------------------------------------------- tuning knobs
namespace {
struct Knob { const char* name; int Tuning::*field; };
const Knob KNOBS[] = {
    {"MI_ROWS_PER_BAND", &Tuning::rows_per_band}, {"MI_CHAIN_ROWS_PER_BAND", &Tuning::chain_rows_per_band},
    {"MI_CHAIN_MARGIN", &Tuning::chain_margin}, {"MI_FFT_WAVES", &Tuning::fft_waves},
    {"MI_FFT_PERSISTENT", &Tuning::fft_persistent}, {"MI_FFT_PREFETCH", &Tuning::fft_prefetch},
    {"MI_FFT_GROUP_MIB", &Tuning::fft_group_mib}, {"MI_FFT_SPLIT", &Tuning::fft_split}, {"MI_FFT_TWOPASS", &Tuning::fft_twopass}, {"MI_FFT_TWOPASS_REMAP", &Tuning::fft_twopass_remap}, {"MI_FFT_TWOPASS_GROUP", &Tuning::fft_twopass_group}, {"MI_FFT_FUSED_SPECTRAL", &Tuning::fft_fused_spectral},
    {"MI_FFT_CHIRPZ_BLOCKS_PER_CU", &Tuning::fft_chirpz_blocks_per_cu},
    {"MI_FFT_FUSED_CHIRPZ", &Tuning::fft_fused_chirpz}, {"MI_CHAIN_LDS_PAD_KIB", &Tuning::chain_lds_pad_kib}, {"MI_LDS_PAD_KIB", &Tuning::lds_pad_kib}, {"MI_SEP_LDS_PAD_KIB", &Tuning::sep_lds_pad_kib},
    {"MI_PNG_ZLIB", &Tuning::png_zlib}, {"MI_GAUSS_WIDE", &Tuning::gauss_wide}, {"MI_GAUSS_MFMA", &Tuning::gauss_mfma}, {"MI_GAUSS_MFMA_MAX", &Tuning::gauss_mfma_max}, {"MI_MEAN_MFMA", &Tuning::mean_mfma}, {"MI_MFMA_SERPENTINE", &Tuning::mfma_serpentine}, {"MI_GM_STRIP", &Tuning::gm_strip}, {"MI_MEAN_MFMA_MAX", &Tuning::mean_mfma_max}, {"MI_GAUSS_DIRECT_WIN", &Tuning::gauss_direct_win}, {"MI_MEAN_PREFIX", &Tuning::mean_prefix}, {"MI_HOST_COPY_THREADS", &Tuning::host_copy_threads},
};
Tuning& tuning_rw()
{
    static Tuning t = [] {                 // the environment is read here, once per process
        Tuning v;
        for (const Knob& k : KNOBS) {
            const char* e = getenv(k.name);
            if (e && *e) v.*(k.field) = atoi(e);
        }
        return v;
    }();
    return t;
}
}  // namespace

const Tuning& tuning() { return tuning_rw(); }

bool set_tuning(const char* name, int value)
{
    for (const Knob& k : KNOBS)
        if (name && strcmp(name, k.name) == 0) {
            // a plain int store: callers change knobs between launches, not concurrently with them
            tuning_rw().*(k.field) = value;
            return true;
        }
    return false;
}

bool get_tuning(const char* name, int* value)
{
    for (const Knob& k : KNOBS)
        if (name && strcmp(name, k.name) == 0) {
            *value = tuning_rw().*(k.field);
            return true;
        }
    return false;
}

// generateGaussianKernal, reference kernels/imageTools.cu:362-385: expf of a float argument,
// division by the double 2*pi*sigma^2, narrowed to float, summed in float in i-major order,
// every entry divided by the sum.
void gaussian_kernel_2d(int k, float sigma, float* out)
{
    float sum = 0.0f;
    const int half = k / 2;
    for (int i = -half; i <= half; i++)
        for (int j = -half; j <= half; j++) {
            const float e = expf((float)(-(i * i + j * j)) / (2 * sigma * sigma));
            const float value = (float)((double)e / (2 * M_PI * (double)sigma * (double)sigma));
            out[(i + half) * k + (j + half)] = value;
            sum += value;
        }
    for (int i = 0; i < k * k; i++) out[i] /= sum;
}

// 1-D factor of the kernel above: w[i] = e_i / sum(e), evaluated in double and rounded once.
void gaussian_kernel_1d(int k, float sigma, float* out)
{
    const int half = k / 2;
    double e[MI_MAX_KERNEL_SIZE], sum = 0.0;
    for (int i = -half; i <= half; i++) {
        e[i + half] = exp(-(double)(i * i) / (2.0 * (double)sigma * (double)sigma));
        sum += e[i + half];
    }
    for (int i = 0; i < k; i++) out[i] = (float)(e[i] / sum);
}

static bool overlap(const void* a, size_t na, const void* b, size_t nb)
{
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a0 < b0 + nb && b0 < a0 + na;
}

static int check_image(const void* in, const void* out, int w, int h, int batch)
{
    if (!in || !out) return fail(MI_ERR_INVALID_ARG, "null image pointer");
    if (w <= 0 || h <= 0 || batch <= 0) return fail(MI_ERR_INVALID_ARG, "non-positive size %dx%d batch %d", w, h, batch);
    if ((int64_t)w * h > (int64_t)1 << 40) return fail(MI_ERR_INVALID_ARG, "image too large");
    return MI_OK;
}

int check_filter(int filter, int w, int h, int ksize, float sigma)
{
    if (filter == MI_FILTER_GRAYSCALE) return MI_OK;
    if (w < 3 || h < 3) return fail(MI_ERR_INVALID_ARG, "stencil filters need w>=3 and h>=3 (got %dx%d)", w, h);
    switch (filter) {
    case MI_FILTER_SOBEL: return MI_OK;
    case MI_FILTER_GAUSSIAN:
    case MI_FILTER_GAUSSIAN_DIRECT:
        if (ksize < 1 || ksize % 2 == 0 || ksize > MI_MAX_KERNEL_SIZE_ANY)
            return fail(MI_ERR_BAD_KERNEL_SIZE, "Gaussian kernel size must be odd, in [1,%d] (got %d)", MI_MAX_KERNEL_SIZE_ANY, ksize);
        if (!(sigma > 0.0f) || !isfinite(sigma)) return fail(MI_ERR_INVALID_ARG, "sigma must be positive and finite");
        return MI_OK;
    case MI_FILTER_MEAN:
        if (ksize < 1 || ksize % 2 == 0 || ksize > MI_MAX_KERNEL_SIZE_ANY)
            return fail(MI_ERR_BAD_KERNEL_SIZE, "mean kernel size must be odd, in [1,%d] (got %d)", MI_MAX_KERNEL_SIZE_ANY, ksize);
        return MI_OK;
    case MI_FILTER_MEDIAN:
        if (ksize != 3 && ksize != 5) return fail(MI_ERR_BAD_KERNEL_SIZE, "median kernel size must be 3 or 5 (got %d)", ksize);
        return MI_OK;
    default: return fail(MI_ERR_INVALID_ARG, "unknown filter %d", filter);
    }
}

static int run_frames(int filter, const void* in, void* out, int w, int h, int batch, int ksize,
                      float sigma, void* stream)
{
    int rc = check_image(in, out, w, h, batch);
    if (rc) return rc;
    rc = check_filter(filter, w, h, ksize, sigma);
    if (rc) return rc;
    const size_t bytes = (size_t)w * h * 4 * batch;
    if (overlap(in, bytes, out, bytes)) return fail(MI_ERR_OVERLAP, "device-tier in/out buffers overlap");
    // kernel size 1 is the identity on the write set; the generic paths handle it
    FilterArgs a;
    a.filter = filter;
    if (filter == MI_FILTER_GAUSSIAN && (ksize == 1 || ksize > 31)) a.filter = MI_FILTER_GAUSSIAN_DIRECT;
    a.in = in; a.out = out; a.w = w; a.h_total = h; a.batch = batch;
    a.in_row0 = 0; a.in_rows = h; a.y_begin = 0; a.y_end = h;
    a.ksize = ksize; a.sigma = sigma; a.stream = (hipStream_t)stream;
    return launch_filter(a);
}

// Host tier: the reference wrapper's shape (kernels/imageTools.cu:233-253): H2D, kernel(s), D2H,
// blocking until `out` holds the result -- run as the chunked pipeline of host.hip.
static int run_host_ops(const uint8_t* in, uint8_t* out, int w, int h, const mi_op* ops, int n_ops, int chunk_rows)
{
    int rc = check_image(in, out, w, h, 1);
    if (rc) return rc;
    if (!ops || n_ops <= 0 || n_ops > MI_MAX_OPS) return fail(MI_ERR_INVALID_ARG, "need 1..%d ops (got %d)", MI_MAX_OPS, n_ops);
    for (int k = 0; k < n_ops; k++) {
        rc = check_filter(ops[k].filter, w, h, ops[k].ksize, ops[k].sigma);
        if (rc) return rc;
    }
    const size_t bytes = (size_t)w * h * 4;
    if (in != out && overlap(in, bytes, out, bytes)) return fail(MI_ERR_OVERLAP, "partially overlapping host buffers");
    return host_run_ops(in, out, w, h, ops, n_ops, chunk_rows);
}

static int run_host(int filter, const uint8_t* in, uint8_t* out, int w, int h, int ksize, float sigma)
{
    const mi_op op = {filter, ksize, sigma};
    return run_host_ops(in, out, w, h, &op, 1, 0);
}

// ---------------------------------------------------------------- PNG (RGBA8, zlib deflate)
static void put_be32(std::vector<uint8_t>& v, uint32_t x)
{
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}

static void png_chunk(std::vector<uint8_t>& png, const char* tag, const uint8_t* data, size_t n)
{
    put_be32(png, (uint32_t)n);
    const size_t start = png.size();
    png.insert(png.end(), tag, tag + 4);
    if (n) png.insert(png.end(), data, data + n);
    put_be32(png, (uint32_t)crc32(0L, png.data() + start, (uInt)(n + 4)));
}

static int write_png_rgba(const char* filename, const uint8_t* px, int w, int h)
{
    const size_t pitch = (size_t)w * 4;
    std::vector<uint8_t> raw((pitch + 1) * h);
    for (int y = 0; y < h; y++) {
        raw[(pitch + 1) * y] = 0;  // filter type None
        memcpy(&raw[(pitch + 1) * y + 1], px + pitch * y, pitch);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK)
        return fail(MI_ERR_IO, "zlib compress failed");
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(png, "IHDR", ihdr.data(), ihdr.size());
    png_chunk(png, "IDAT", z.data(), zn);
    png_chunk(png, "IEND", nullptr, 0);
    FILE* f = fopen(filename, "wb");
    if (!f) return fail(MI_ERR_IO, "cannot open %s for writing", filename);
    const bool ok = fwrite(png.data(), 1, png.size(), f) == png.size();
    fclose(f);
    return ok ? MI_OK : fail(MI_ERR_IO, "short write to %s", filename);
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_version(void) { return 100; }
const char* mi_last_error(void) { return g_err; }

int mi_debug_set_tuning(const char* name, int value)
{
    return set_tuning(name, value) ? MI_OK : fail(MI_ERR_INVALID_ARG, "unknown tuning knob %s", name ? name : "(null)");
}

int mi_debug_get_tuning(const char* name, int* value)
{
    if (!value) return fail(MI_ERR_INVALID_ARG, "null pointer");
    return get_tuning(name, value) ? MI_OK : fail(MI_ERR_INVALID_ARG, "unknown tuning knob %s", name ? name : "(null)");
}

int mi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// ---------------------------------------------------------------- device tier
int mi_image_load_dev(const void* d_bytes, void* d_pixels, size_t n_pixels, void* stream)
{
    if (!d_bytes || !d_pixels) return fail(MI_ERR_INVALID_ARG, "null pointer");
    if (overlap(d_bytes, n_pixels * 4, d_pixels, n_pixels * 4)) return fail(MI_ERR_OVERLAP, "buffers overlap");
    return launch_copy(d_bytes, d_pixels, n_pixels * 4, (hipStream_t)stream);
}

int mi_image_write_dev(void* d_bytes, const void* d_pixels, int w, int h, void* stream)
{
    if (!d_bytes || !d_pixels || w <= 0 || h <= 0) return fail(MI_ERR_INVALID_ARG, "bad argument");
    const size_t n = (size_t)w * h * 4;
    if (overlap(d_bytes, n, d_pixels, n)) return fail(MI_ERR_OVERLAP, "buffers overlap");
    return launch_copy(d_pixels, d_bytes, n, (hipStream_t)stream);
}

int mi_grayscale_u8x4_dev(const void* d_in, void* d_out, int w, int h, int batch, void* stream)
{
    return run_frames(MI_FILTER_GRAYSCALE, d_in, d_out, w, h, batch, 0, 0.f, stream);
}

int mi_sobel_u8x4_dev(const void* d_in, void* d_out, int w, int h, int batch, void* stream)
{
    return run_frames(MI_FILTER_SOBEL, d_in, d_out, w, h, batch, 3, 0.f, stream);
}

int mi_gaussian_u8x4_dev(const void* d_in, void* d_out, int w, int h, int batch, int kernel_size,
                         float sigma, int mode, void* stream)
{
    if (mode != MI_GAUSS_SEPARABLE && mode != MI_GAUSS_DIRECT) return fail(MI_ERR_INVALID_ARG, "unknown Gaussian mode %d", mode);
    return run_frames(mode == MI_GAUSS_DIRECT ? MI_FILTER_GAUSSIAN_DIRECT : MI_FILTER_GAUSSIAN, d_in,
                      d_out, w, h, batch, kernel_size, sigma, stream);
}

int mi_meanblur_u8x4_dev(const void* d_in, void* d_out, int w, int h, int batch, int kernel_size, void* stream)
{
    return run_frames(MI_FILTER_MEAN, d_in, d_out, w, h, batch, kernel_size, 0.f, stream);
}

int mi_medianblur_u8x4_dev(const void* d_in, void* d_out, int w, int h, int batch, int kernel_size, void* stream)
{
    return run_frames(MI_FILTER_MEDIAN, d_in, d_out, w, h, batch, kernel_size, 0.f, stream);
}

int mi_chain_gray_gauss5_median3_sobel_u8x4_dev(const void* d_in, void* d_out, int w, int h, int batch, float sigma, void* stream)
{
    int rc = check_image(d_in, d_out, w, h, batch);
    if (rc) return rc;
    rc = check_filter(MI_FILTER_GAUSSIAN, w, h, 5, sigma);
    if (rc) return rc;
    const size_t bytes = (size_t)w * h * 4 * batch;
    if (overlap(d_in, bytes, d_out, bytes)) return fail(MI_ERR_OVERLAP, "device-tier in/out buffers overlap");
    return launch_chain(d_in, d_out, w, h, batch, sigma, (hipStream_t)stream);
}

int mi_filter_halo_rows(int filter, int param_i) { return filter_halo_rows(filter, param_i); }

int mi_filter_strip_u8x4_dev(int filter, const void* d_in, void* d_out, int w, int h_total,
                             int in_row0, int in_rows, int y_begin, int y_end, int param_i,
                             float param_f, void* stream)
{
    int rc = check_image(d_in, d_out, w, h_total, 1);
    if (rc) return rc;
    rc = check_filter(filter, w, h_total, param_i, param_f);
    if (rc) return rc;
    if (y_begin < 0 || y_end > h_total || y_begin >= y_end || in_rows <= 0)
        return fail(MI_ERR_INVALID_ARG, "bad row range [%d,%d) of %d", y_begin, y_end, h_total);
    const int halo = filter_halo_rows(filter, param_i, w);
    const int need0 = std::max(0, y_begin - halo), need1 = std::min(h_total, y_end + halo);
    if (in_row0 < 0 || in_row0 > need0 || in_row0 + in_rows < need1 || in_row0 + in_rows > h_total)
        return fail(MI_ERR_INVALID_ARG, "input rows [%d,%d) do not cover the needed rows [%d,%d)",
                    in_row0, in_row0 + in_rows, need0, need1);
    if (overlap(d_in, (size_t)in_rows * w * 4, d_out, (size_t)(y_end - y_begin) * w * 4))
        return fail(MI_ERR_OVERLAP, "device-tier in/out buffers overlap");
    FilterArgs a;
    a.filter = filter;
    if (filter == MI_FILTER_GAUSSIAN && (param_i == 1 || param_i > 31)) a.filter = MI_FILTER_GAUSSIAN_DIRECT;
    a.in = d_in; a.out = d_out; a.w = w; a.h_total = h_total; a.batch = 1;
    a.in_row0 = in_row0; a.in_rows = in_rows; a.y_begin = y_begin; a.y_end = y_end;
    a.ksize = param_i; a.sigma = param_f; a.stream = (hipStream_t)stream;
    return launch_filter(a);
}

int mi_synth_u8x4_dev(void* d_out, uint64_t seed, uint64_t first_idx, uint64_t count, void* stream)
{
    if (!d_out) return fail(MI_ERR_INVALID_ARG, "null pointer");
    return launch_synth(d_out, seed, first_idx, count, (hipStream_t)stream);
}

int mi_debug_sobel_mag_table_dev(void* d_out_u8, uint32_t count, void* stream)
{
    if (!d_out_u8) return fail(MI_ERR_INVALID_ARG, "null pointer");
    return launch_sobel_table(d_out_u8, count, (hipStream_t)stream);
}

int mi_debug_sobel_mag_pairs_dev(void* d_out_u8, int gmax, int variant, void* stream)
{
    if (!d_out_u8) return fail(MI_ERR_INVALID_ARG, "null pointer");
    if (gmax < 0 || gmax > 1020) return fail(MI_ERR_INVALID_ARG, "gmax must be in [0, 1020]");
    return launch_sobel_pairs(d_out_u8, gmax, variant, (hipStream_t)stream);
}

// ---------------------------------------------------------------- FFT path, device tier
static int check_plane(const void* p, int w, int h)
{
    if (!p) return fail(MI_ERR_INVALID_ARG, "null pointer");
    if (w <= 0 || h <= 0) return fail(MI_ERR_INVALID_ARG, "non-positive size %dx%d", w, h);
    return MI_OK;
}

int mi_u8_to_complex_dev(const void* d_in, void* d_red, void* d_green, void* d_blue, int w, int h, void* stream)
{
    if (!d_red || !d_green || !d_blue) return fail(MI_ERR_INVALID_ARG, "null pointer");
    int rc = check_plane(d_in, w, h);
    if (rc) return rc;
    return launch_u8_to_complex(d_in, d_red, d_green, d_blue, w, h, (hipStream_t)stream);
}

int mi_fft2d_c2c_f32_dev(const void* d_in, void* d_out, int rows, int cols, int batch, int direction, void* stream)
{
    if (!d_in || !d_out) return fail(MI_ERR_INVALID_ARG, "null pointer");
    if (batch <= 0) return fail(MI_ERR_INVALID_ARG, "non-positive batch");
    if (direction != MI_FFT_FORWARD && direction != MI_FFT_INVERSE) return fail(MI_ERR_INVALID_ARG, "direction must be -1 or +1");
    auto pow2 = [](int n) { return n >= 2 && (n & (n - 1)) == 0; };
    if (!pow2(rows) || !pow2(cols) || !fft2d_size_supported(rows, cols))
        return fail(MI_ERR_UNSUPPORTED, "fft2d needs power-of-two sizes in [2,%d] and fewer than 2^28 points a plane (got %dx%d)", MI_FFT_MAX_POW2, rows, cols);
    const size_t bytes = (size_t)rows * cols * 8 * batch;
    if (d_in != d_out && overlap(d_in, bytes, d_out, bytes)) return fail(MI_ERR_OVERLAP, "partially overlapping buffers");
    return launch_fft2d(d_in, d_out, rows, cols, batch, direction, (hipStream_t)stream);
}

int mi_fft2d_workspace_bytes(int rows, int cols, size_t* bytes)
{
    if (!bytes) return fail(MI_ERR_INVALID_ARG, "null pointer");
    if (!fft2d_size_supported(rows, cols))
        return fail(MI_ERR_UNSUPPORTED, "fft2d sizes must be in [2,%d] or a power of two up to %d, fewer than 2^28 points a plane (got %dx%d)",
                    MI_FFT_MAX_ANY, MI_FFT_MAX_POW2, rows, cols);
    *bytes = fft2d_workspace_bytes(rows, cols);
    return MI_OK;
}

int mi_fft2d_c2c_f32_ws_dev(const void* d_in, void* d_out, int rows, int cols, int batch, int direction,
                            void* d_workspace, size_t workspace_bytes, void* stream)
{
    if (!d_in || !d_out) return fail(MI_ERR_INVALID_ARG, "null pointer");
    if (batch <= 0) return fail(MI_ERR_INVALID_ARG, "non-positive batch");
    if (direction != MI_FFT_FORWARD && direction != MI_FFT_INVERSE) return fail(MI_ERR_INVALID_ARG, "direction must be -1 or +1");
    size_t need = 0;
    int rc = mi_fft2d_workspace_bytes(rows, cols, &need);
    if (rc) return rc;
    const bool pow2 = (rows & (rows - 1)) == 0 && (cols & (cols - 1)) == 0;
    if (pow2) {
        // powers of two: the workspace is optional (one plane of it lets a tall plane be transformed in
        // place at full speed, and lets the two-pass plan run its rows pass first when asked to)
        const size_t bytes = (size_t)rows * cols * 8 * batch;
        if (d_in != d_out && overlap(d_in, bytes, d_out, bytes)) return fail(MI_ERR_OVERLAP, "partially overlapping buffers");
        void* tmp = (d_workspace && need && workspace_bytes >= need) ? d_workspace : nullptr;
        if (tmp && (overlap(tmp, need, d_in, bytes) || overlap(tmp, need, d_out, bytes)))
            return fail(MI_ERR_OVERLAP, "workspace overlaps the data");
        return launch_fft2d(d_in, d_out, rows, cols, batch, direction, (hipStream_t)stream, tmp);
    }
    if (need && (!d_workspace || workspace_bytes < need))
        return fail(MI_ERR_INVALID_ARG, "fft2d %dx%d needs a workspace of %zu bytes (got %zu)", rows, cols, need, d_workspace ? workspace_bytes : (size_t)0);
    const size_t bytes = (size_t)rows * cols * 8 * batch;
    if (d_in != d_out && overlap(d_in, bytes, d_out, bytes)) return fail(MI_ERR_OVERLAP, "partially overlapping buffers");
    if (need && (overlap(d_workspace, need, d_in, bytes) || overlap(d_workspace, need, d_out, bytes)))
        return fail(MI_ERR_OVERLAP, "workspace overlaps the data");
    return launch_fft2d_any(d_in, d_out, rows, cols, batch, direction, d_workspace, (hipStream_t)stream);
}

int mi_fftshift_dev(void* d_data, int w, int h, void* stream)
{
    int rc = check_plane(d_data, w, h);
    return rc ? rc : launch_fftshift(d_data, w, h, (hipStream_t)stream);
}

int mi_highpass_dev(void* d_data, int w, int h, float cutoff, void* stream)
{
    int rc = check_plane(d_data, w, h);
    if (rc) return rc;
    if (!(cutoff > 0.0f)) return fail(MI_ERR_INVALID_ARG, "cutoff must be positive");
    return launch_highpass(d_data, w, h, cutoff, (hipStream_t)stream);
}

int mi_normalize_zero_imag_dev(void* d_data, int w, int h, void* stream)
{
    int rc = check_plane(d_data, w, h);
    return rc ? rc : launch_normalize_zero_imag(d_data, w, h, (hipStream_t)stream);
}

int mi_complex_to_u8_dev(const void* d_red, const void* d_green, const void* d_blue, void* d_out, int w, int h, void* stream)
{
    if (!d_red || !d_green || !d_blue) return fail(MI_ERR_INVALID_ARG, "null pointer");
    int rc = check_plane(d_out, w, h);
    return rc ? rc : launch_complex_to_u8(d_red, d_green, d_blue, d_out, w, h, (hipStream_t)stream);
}

int mi_minmax_f32_dev(const void* d_img, size_t n, void* d_minmax, void* stream)
{
    if (!d_img || !d_minmax || n == 0) return fail(MI_ERR_INVALID_ARG, "bad argument");
    return launch_minmax(d_img, n, d_minmax, (hipStream_t)stream);
}

int mi_normalize_to_u8_dev(const void* d_img, void* d_out, size_t n, const void* d_minmax, void* stream)
{
    if (!d_img || !d_out || !d_minmax || n == 0) return fail(MI_ERR_INVALID_ARG, "bad argument");
    return launch_normalize_to_u8(d_img, d_out, n, d_minmax, (hipStream_t)stream);
}

// ---------------------------------------------------------------- host tier
int mi_image_load(const uint8_t* bytes, uint8_t* pixels, size_t n_pixels)
{
    if (!bytes || !pixels) return fail(MI_ERR_INVALID_ARG, "null pointer");
    if (n_pixels == 0) return MI_OK;
    if (bytes != pixels && overlap(bytes, n_pixels * 4, pixels, n_pixels * 4)) return fail(MI_ERR_OVERLAP, "partially overlapping host buffers");
    // (the same buffer twice is fine: a chunk is read before it is written)
    return host_repack(bytes, pixels, n_pixels * 4);
}

int mi_image_write(const char* filename, const uint8_t* pixels, int w, int h)
{
    if (!filename || !pixels || w <= 0 || h <= 0) return fail(MI_ERR_INVALID_ARG, "bad argument");
    // The reference round-trips the pixels through the device (imageWrite kernel, a copy) before
    // encoding; the bytes are identical, so the encoder reads the caller's buffer directly.  The encoder is
    // the reference's own, stb_image_write v1.16 (imageTools.cu:152): the same file, byte for byte.
    // MI_PNG_ZLIB=1 selects the zlib encoder instead (smaller files, 3x faster on large images).
    if (tuning().png_zlib) return write_png_rgba(filename, pixels, w, h);
    if ((int64_t)w * 4 > 0x7fffffff) return fail(MI_ERR_INVALID_ARG, "image too wide for the PNG encoder");
    return png_write_rgba_stb(filename, pixels, w, h) ? MI_OK : fail(MI_ERR_IO, "cannot write %s", filename);
}

int mi_grayscale_u8x4(const uint8_t* in, uint8_t* out, int w, int h)
{
    return run_host(MI_FILTER_GRAYSCALE, in, out, w, h, 0, 0.f);
}

int mi_sobel_u8x4(const uint8_t* in, uint8_t* out, int w, int h)
{
    return run_host(MI_FILTER_SOBEL, in, out, w, h, 3, 0.f);
}

int mi_gaussian_u8x4(const uint8_t* in, uint8_t* out, int w, int h, int kernel_size, float sigma, int mode)
{
    if (mode != MI_GAUSS_SEPARABLE && mode != MI_GAUSS_DIRECT) return fail(MI_ERR_INVALID_ARG, "unknown Gaussian mode %d", mode);
    return run_host(mode == MI_GAUSS_DIRECT ? MI_FILTER_GAUSSIAN_DIRECT : MI_FILTER_GAUSSIAN, in, out, w, h, kernel_size, sigma);
}

int mi_meanblur_u8x4(const uint8_t* in, uint8_t* out, int w, int h, int kernel_size)
{
    return run_host(MI_FILTER_MEAN, in, out, w, h, kernel_size, 0.f);
}

int mi_medianblur_u8x4(const uint8_t* in, uint8_t* out, int w, int h, int kernel_size)
{
    return run_host(MI_FILTER_MEDIAN, in, out, w, h, kernel_size, 0.f);
}

int mi_run_ops_u8x4(const uint8_t* in, uint8_t* out, int w, int h, const mi_op* ops, int n_ops)
{
    return run_host_ops(in, out, w, h, ops, n_ops, 0);
}

int mi_run_ops_chunked_u8x4(const uint8_t* in, uint8_t* out, int w, int h, const mi_op* ops, int n_ops, int chunk_rows)
{
    if (chunk_rows <= 0) return fail(MI_ERR_INVALID_ARG, "chunk_rows must be positive");
    return run_host_ops(in, out, w, h, ops, n_ops, chunk_rows);
}

static int check_devices(const int* devices, int n_devices)
{
    if (!devices || n_devices <= 0 || n_devices > 64) return fail(MI_ERR_INVALID_ARG, "need 1..64 devices");
    const int count = mi_device_count();
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= count) return fail(MI_ERR_INVALID_ARG, "device %d of %d visible", devices[i], count);
    return MI_OK;
}

static int check_ops(int w, int h, const mi_op* ops, int n_ops)
{
    if (!ops || n_ops <= 0 || n_ops > MI_MAX_OPS) return fail(MI_ERR_INVALID_ARG, "need 1..%d ops (got %d)", MI_MAX_OPS, n_ops);
    for (int k = 0; k < n_ops; k++) {
        const int rc = check_filter(ops[k].filter, w, h, ops[k].ksize, ops[k].sigma);
        if (rc) return rc;
    }
    return MI_OK;
}

int mi_batch_run_ops_u8x4(const uint8_t* in, uint8_t* out, int w, int h, int n_frames, const mi_op* ops, int n_ops,
                          const int* devices, int n_devices)
{
    int rc = check_image(in, out, w, h, n_frames);
    if (rc == MI_OK) rc = check_ops(w, h, ops, n_ops);
    if (rc == MI_OK) rc = check_devices(devices, n_devices);
    if (rc) return rc;
    const size_t bytes = (size_t)w * h * 4 * n_frames;
    if (in != out && overlap(in, bytes, out, bytes)) return fail(MI_ERR_OVERLAP, "partially overlapping host buffers");
    return host_batch_run(in, out, w, h, n_frames, ops, n_ops, devices, n_devices);
}

int mi_strip_run_ops_u8x4(const uint8_t* in, uint8_t* out, int w, int h, const mi_op* ops, int n_ops, const int* devices,
                          int n_devices)
{
    int rc = check_image(in, out, w, h, 1);
    if (rc == MI_OK) rc = check_ops(w, h, ops, n_ops);
    if (rc == MI_OK) rc = check_devices(devices, n_devices);
    if (rc) return rc;
    const size_t bytes = (size_t)w * h * 4;
    // strips run concurrently and read each other's rows of `in` as halo: in place only on one device
    if (overlap(in, bytes, out, bytes) && !(in == out && n_devices == 1))
        return fail(MI_ERR_OVERLAP, "mi_strip_run_ops_u8x4 on several devices needs out != in");
    return host_strip_run(in, out, w, h, ops, n_ops, devices, n_devices);
}

int mi_host_release(void) { return host_release(); }

// imageFFTImageGenerate's scaffold (imageTools.cu:633-791): u8 -> three complex planes -> forward
// transform -> `edit` on the three spectra -> inverse -> /N, imaginary part dropped -> u8.
// `rows` x `cols` is the geometry the transforms see (rows * cols == w * h).
// Device core: d_in / d_out are device images (they may be the same buffer), `work` is device scratch
// laid out as [three planes | three spectra | transform workspace]; nothing is allocated, everything
// goes to `s`.  tables: the host tier's record of the chirp-z tables in `work` (null: rebuild them).
static size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

static int spectral_work_bytes(int w, int h, int rows, int cols, int batch, size_t* bytes)
{
    size_t ws_bytes = 0;
    const int rc = mi_fft2d_workspace_bytes(rows, cols, &ws_bytes);
    if (rc) return rc;
    *bytes = 2 * up256((size_t)w * h * 8 * 3 * batch) + ws_bytes;
    return MI_OK;
}

extern "C++" template <class Edit>
static int spectral_roundtrip_dev(const void* d_in, void* d_out, int w, int h, int batch, int rows, int cols, Edit edit,
                                  void* work, HostScratch::TableKey* tables, hipStream_t s)
{
    const int planes = 3 * batch;      // image k's colour planes are planes 3k, 3k+1, 3k+2
    const size_t n = (size_t)w * h;
    size_t ws_bytes = 0;
    int rc = mi_fft2d_workspace_bytes(rows, cols, &ws_bytes);
    if (rc) return rc;
    const size_t planes_bytes = up256(n * 8 * planes);
    char* pl = (char*)work;
    char* spec = pl + planes_bytes;
    void* ws = ws_bytes ? spec + planes_bytes : nullptr;
    rc = launch_u8_to_complex(d_in, pl, pl + n * 8, pl + 2 * n * 8, w, h, s, batch);
    // chirp-z sizes: the tables in a cached workspace survive from the forward transform to the inverse
    // one, and from one call to the next of the same geometry (nothing else writes there)
    const bool chirpz = ws_bytes && !(((rows & (rows - 1)) == 0) && ((cols & (cols - 1)) == 0));
    HostScratch::TableKey none{nullptr, 0, 0};
    HostScratch::TableKey& key = tables ? *tables : none;
    // any other geometry lays the buffer out differently and may have overwritten the tables
    if (!chirpz || !(key.at == ws && key.rows == rows && key.cols == cols)) key = HostScratch::TableKey{nullptr, 0, 0};
    auto transform = [&](const void* src, void* dst, int dir) -> int {
        if (!chirpz) return mi_fft2d_c2c_f32_ws_dev(src, dst, rows, cols, planes, dir, ws, ws_bytes, s);
        const bool ready = tables && key.at != nullptr;
        const int r = launch_fft2d_any(src, dst, rows, cols, planes, dir, ws, s, ready);
        key = r == MI_OK ? HostScratch::TableKey{ws, rows, cols} : HostScratch::TableKey{nullptr, 0, 0};
        return r;
    };
    if (rc == MI_OK) rc = transform(pl, spec, MI_FFT_FORWARD);
    if (rc == MI_OK) rc = edit(spec);                       // the three spectra, back to back
    if (rc == MI_OK) rc = transform(spec, pl, MI_FFT_INVERSE);
    // NormalizeAndZeroImaginary and ComplexRGBToUchar in one pass
    if (rc == MI_OK) rc = launch_complex_to_u8_normalized(pl, pl + n * 8, pl + 2 * n * 8, d_out, w, h, batch, s);
    return rc;
}

// Host tier: one cached device buffer (the image, then the device core's scratch), blocking copies.
extern "C++" template <class Edit>
static int spectral_roundtrip(const uint8_t* in, uint8_t* out, int w, int h, int rows, int cols, Edit edit)
{
    int rc = check_image(in, out, w, h, 1);
    if (rc) return rc;
    const size_t n = (size_t)w * h;
    size_t work_bytes = 0;
    rc = spectral_work_bytes(w, h, rows, cols, 1, &work_bytes);
    if (rc) return rc;
    const size_t img_bytes = up256(n * 4);
    HostScratch scratch;
    rc = host_scratch(img_bytes + work_bytes, &scratch);
    if (rc) return rc;
    char* d_img = (char*)scratch.ptr;
    rc = host_upload(d_img, in, n * 4);
    if (rc == MI_OK) rc = spectral_roundtrip_dev(d_img, d_img, w, h, 1, rows, cols, edit, d_img + img_bytes, scratch.tables, nullptr);
    // (the round trip runs on the null stream, which the staging streams do not wait for)
    if (rc == MI_OK) rc = hip_status(hipStreamSynchronize(nullptr), "spectral round trip");
    if (rc == MI_OK) rc = host_download(out, d_img, n * 4);
    return rc;
}

int mi_fft_highpass_batch_workspace_bytes(int w, int h, int batch, size_t* bytes)
{
    if (!bytes) return fail(MI_ERR_INVALID_ARG, "null pointer");
    if (w <= 0 || h <= 0 || batch <= 0) return fail(MI_ERR_INVALID_ARG, "non-positive size %dx%d or batch %d", w, h, batch);
    if ((uint64_t)batch * 3 > 0x3fffffffu) return fail(MI_ERR_UNSUPPORTED, "batch too large");
    return spectral_work_bytes(w, h, /*rows=*/w, /*cols=*/h, batch, bytes);
}

int mi_fft_highpass_workspace_bytes(int w, int h, size_t* bytes)
{
    return mi_fft_highpass_batch_workspace_bytes(w, h, 1, bytes);
}

int mi_fft_highpass_batch_u8x4_dev(const void* d_in, void* d_out, int w, int h, int batch, float cutoff,
                                   void* d_workspace, size_t workspace_bytes, void* stream)
{
    if (!d_in || !d_out || !d_workspace) return fail(MI_ERR_INVALID_ARG, "null pointer");
    if (!(cutoff > 0.0f)) return fail(MI_ERR_INVALID_ARG, "cutoff must be positive");
    size_t need = 0;
    int rc = mi_fft_highpass_batch_workspace_bytes(w, h, batch, &need);
    if (rc) return rc;
    if (batch > 65535) return fail(MI_ERR_UNSUPPORTED, "at most 65535 images per call (got %d)", batch);
    if (workspace_bytes < need)
        return fail(MI_ERR_INVALID_ARG, "fft_highpass %dx%d x%d needs a workspace of %zu bytes (got %zu)", w, h, batch, need, workspace_bytes);
    const size_t img = (size_t)w * h * 4 * batch;
    if (d_in != d_out && overlap(d_in, img, d_out, img)) return fail(MI_ERR_OVERLAP, "partially overlapping images");
    if (overlap(d_workspace, need, d_in, img) || overlap(d_workspace, need, d_out, img))
        return fail(MI_ERR_OVERLAP, "workspace overlaps an image");
    hipStream_t s = (hipStream_t)stream;
    if (w == h && fft_highpass_fused_supported(/*rows=*/w, /*cols=*/h)) {
        // the two plane areas of the workspace; conversions and the filter ride on the transform's own passes
        const size_t planes_bytes = up256((size_t)w * h * 8 * 3 * batch);
        return launch_fft_highpass_fused(d_in, d_out, w, h, batch, w, h, cutoff, d_workspace, (char*)d_workspace + planes_bytes, s);
    }
    return spectral_roundtrip_dev(d_in, d_out, w, h, batch, /*rows=*/w, /*cols=*/h, [&](char* spectra) {
        return launch_highpass_unshifted(spectra, w, h, 3 * batch, cutoff, s);
    }, d_workspace, nullptr, s);
}

int mi_fft_highpass_u8x4_dev(const void* d_in, void* d_out, int w, int h, float cutoff, void* d_workspace,
                             size_t workspace_bytes, void* stream)
{
    return mi_fft_highpass_batch_u8x4_dev(d_in, d_out, w, h, 1, cutoff, d_workspace, workspace_bytes, stream);
}

int mi_fft_highpass_u8x4(const uint8_t* in, uint8_t* out, int w, int h, float cutoff)
{
    if (!(cutoff > 0.0f)) return fail(MI_ERR_INVALID_ARG, "cutoff must be positive");
    // The reference plans its transform as cufftPlan2d(&plan, width, height, ...) (imageTools.cu:688), and
    // cufftPlan2d's first size is the slowest-varying dimension: the buffer is transformed as `width` rows
    // of `height` points, while the shift and the filter index it as `height` rows of `width`.  Square
    // images cannot tell; the reference's own 675x900 sample output is reproduced only this way.
    if (w == h && fft_highpass_fused_supported(/*rows=*/w, /*cols=*/h)) {
        int rc = check_image(in, out, w, h, 1);
        if (rc) return rc;
        const size_t n = (size_t)w * h, img_bytes = up256(n * 4), planes_bytes = up256(n * 8 * 3);
        HostScratch scratch;
        rc = host_scratch(img_bytes + 2 * planes_bytes, &scratch);
        if (rc) return rc;
        if (scratch.tables) *scratch.tables = HostScratch::TableKey{nullptr, 0, 0};      // this layout overwrites any chirp tables
        char* d_img = (char*)scratch.ptr;
        rc = host_upload(d_img, in, n * 4);
        if (rc == MI_OK) rc = launch_fft_highpass_fused(d_img, d_img, w, h, 1, w, h, cutoff, d_img + img_bytes, d_img + img_bytes + planes_bytes, nullptr);
        if (rc == MI_OK) rc = hip_status(hipStreamSynchronize(nullptr), "spectral round trip");
        if (rc == MI_OK) rc = host_download(out, d_img, n * 4);
        return rc;
    }
    return spectral_roundtrip(in, out, w, h, /*rows=*/w, /*cols=*/h, [&](char* spectra) {
        // shift, filter, shift (imageTools.cu:704-737) as one pass over the three planes
        return launch_highpass_unshifted(spectra, w, h, 3, cutoff, nullptr);
    });
}

static size_t keep_count(size_t n, float keep_fraction)
{
    const double k = ceil((double)keep_fraction * (double)n);
    return (size_t)std::min<double>((double)n, std::max<double>(1.0, k));
}

int mi_compress_u8x4(const uint8_t* in, uint8_t* out, int w, int h, float keep_fraction)
{
    if (!(keep_fraction > 0.0f) || !(keep_fraction <= 1.0f))
        return fail(MI_ERR_INVALID_ARG, "keep_fraction must be in (0, 1] (got %g)", (double)keep_fraction);
    const size_t n = (size_t)w * h;
    if (w == h && fft_highpass_fused_supported(h, w)) {
        // conversions on the transform's own first loads and last stores; the selection works on the spectra between
        int rc = check_image(in, out, w, h, 1);
        if (rc) return rc;
        const size_t img_bytes = up256(n * 4), planes_bytes = up256(n * 8 * 3);
        HostScratch scratch;
        rc = host_scratch(img_bytes + 2 * planes_bytes, &scratch);
        if (rc) return rc;
        if (scratch.tables) *scratch.tables = HostScratch::TableKey{nullptr, 0, 0};
        char* d_img = (char*)scratch.ptr;
        char *a = d_img + img_bytes, *b = a + planes_bytes;
        rc = host_upload(d_img, in, n * 4);
        if (rc == MI_OK) rc = launch_spectral_fused_forward(d_img, w, h, 1, h, w, 0.0f, a, b, nullptr);
        if (rc == MI_OK) rc = spectrum_keep_top(b, n, keep_count(n, keep_fraction), 3, nullptr);
        if (rc == MI_OK) rc = launch_spectral_fused_inverse(d_img, w, h, 1, h, w, a, b, nullptr);
        if (rc == MI_OK) rc = hip_status(hipStreamSynchronize(nullptr), "spectral round trip");
        if (rc == MI_OK) rc = host_download(out, d_img, n * 4);
        return rc;
    }
    return spectral_roundtrip(in, out, w, h, /*rows=*/h, /*cols=*/w, [&](char* spectra) {
        return spectrum_keep_top(spectra, n, keep_count(n, keep_fraction), 3, nullptr);
    });
}

int mi_spectrum_keep_top_dev(void* d_data, size_t n, size_t keep, void* stream)
{
    if (!d_data || n == 0) return fail(MI_ERR_INVALID_ARG, "bad argument");
    return spectrum_keep_top(d_data, n, keep, 1, (hipStream_t)stream);
}

int mi_gaussian_kernel(int kernel_size, float sigma, float* out)
{
    if (!out) return fail(MI_ERR_INVALID_ARG, "null pointer");
    if (kernel_size < 1 || kernel_size % 2 == 0 || kernel_size > MI_MAX_KERNEL_SIZE_ANY)
        return fail(MI_ERR_BAD_KERNEL_SIZE, "kernel size must be odd, in [1,%d] (got %d)", MI_MAX_KERNEL_SIZE_ANY, kernel_size);
    if (!(sigma > 0.0f) || !isfinite(sigma)) return fail(MI_ERR_INVALID_ARG, "sigma must be positive and finite");
    gaussian_kernel_2d(kernel_size, sigma, out);
    return MI_OK;
}

}  // extern "C"
