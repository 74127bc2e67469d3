// internal.h -- host-side declarations shared by the translation units of libmi_imagetools.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "../../include/mi_imagetools.h"

namespace mi {

// Records `msg` as the calling thread's last error and returns `code`.
int fail(int code, const char* fmt, ...);
// Maps a hipError_t to the ABI's return value (0 or the positive error code), recording it.
int hip_status(hipError_t e, const char* what);

// Tuning knobs.  The environment (MI_ROWS_PER_BAND, ...) is read ONCE per process, on first use; a
// launch only reads these integers.  mi_debug_set_tuning changes one at run time (tests, sweeps).
struct Tuning {
    int rows_per_band = 0;        // MI_ROWS_PER_BAND: rows a wave marches over (0 = per-filter default)
    int chain_rows_per_band = 0;  // MI_CHAIN_ROWS_PER_BAND
    int chain_margin = 0;         // MI_CHAIN_MARGIN: margin lanes per side of the fused chain's strips
    int fft_waves = 1;            // MI_FFT_WAVES
    int fft_persistent = 1;       // MI_FFT_PERSISTENT
    int fft_prefetch = 1;         // MI_FFT_PREFETCH
    int fft_group_mib = 300;      // MI_FFT_GROUP_MIB: planes per Infinity-Cache group
    int fft_split = 1;            // MI_FFT_SPLIT
    int fft_twopass = 1;          // MI_FFT_TWOPASS: 4096x4096 planes in two passes per direction (fft_twopass.h); 2 = rows pass first where a third buffer allows
    int fft_fused_spectral = 1;   // MI_FFT_FUSED_SPECTRAL: imageFFTImageGenerate's pointwise steps fused into the two-pass kernels
    int fft_twopass_group = 8;    // MI_FFT_TWOPASS_GROUP: planes per launch of the two-pass kernels
    int fft_twopass_remap = 1;    // MI_FFT_TWOPASS_REMAP: XCD-contiguous tiles
    int fft_chirpz_blocks_per_cu = 2;   // MI_FFT_CHIRPZ_BLOCKS_PER_CU
    int fft_fused_chirpz = 1;     // MI_FFT_FUSED_CHIRPZ
    int png_zlib = 0;             // MI_PNG_ZLIB: mi_image_write encodes with zlib instead of the reference's stb_image_write
    int mean_prefix = 1;          // MI_MEAN_PREFIX: mean 17x17..31x31 as a wave march on prefix sums (0: LDS tiles)
    int host_copy_threads = 4;    // MI_HOST_COPY_THREADS: host threads per direction that stage a chunk between pageable and pinned memory
    int gauss_direct_win = 1;     // MI_GAUSS_DIRECT_WIN: direct Gaussians 5x5 / 7x7 from gauss_direct.hip (0: round 2's march kernel; same bits)
    int gauss_wide = 1;           // MI_GAUSS_WIDE: 7x7..15x15 separable Gaussians in 16-bit fixed point (0: the fp32 kernel)
    int gauss_mfma = 1;           // MI_GAUSS_MFMA: 7x7..15x15 separable Gaussians on the matrix cores (gauss_mfma.hip; 0: gauss_wide_kernel)
    int gauss_mfma_max = 31;      // MI_GAUSS_MFMA_MAX: largest separable Gaussian on the matrix cores (15: sizes 17..31 run the direct kernels, as before round 3's wide kernel)
    int gm_strip = 128;           // MI_GM_STRIP: strip width of the radius-8 matrix-core kernel in pixels (128; 64: three waves per SIMD, 256: one; both measured slower; same bytes)
    int mfma_serpentine = 1;      // MI_MFMA_SERPENTINE: odd bands of the matrix-core kernels walk upwards (0: every band downwards; same bytes)
    int mean_mfma = 11;           // MI_MEAN_MFMA: smallest mean-blur size that runs on the matrix cores (sizes up to 31: 11 and above gain, measured; 0: none)
    int mean_mfma_max = 31;       // MI_MEAN_MFMA_MAX: largest one (17: the sizes above run the prefix-sum march, as before the wide matrix-core kernel)
    int lds_pad_kib = 24;         // MI_LDS_PAD_KIB: extra LDS per block of the 3x3 march kernels (20 + 24 KiB: three blocks per CU, +2.5 %)
    int sep_lds_pad_kib = -1;     // MI_SEP_LDS_PAD_KIB: the same for the separable Gaussian (5x5) and mean kernels (-1: by kernel size)
    int chain_lds_pad_kib = 8;    // MI_CHAIN_LDS_PAD_KIB: extra LDS per block of the fused chain (24 + 8 KiB: five blocks per CU, +2 %)
};
const Tuning& tuning();
// name: the environment variable's name; returns false for an unknown name
bool set_tuning(const char* name, int value);
bool get_tuning(const char* name, int* value);

// One filter launch over a (strip of a) batch of frames; see Geo in common.h for the meaning
// of the row arguments.  All pointers are device pointers.
struct FilterArgs {
    int filter;            // MI_FILTER_*
    const void* in;
    void* out;
    int w, h_total;
    int batch;
    int in_row0, in_rows;  // rows present in `in` (per frame)
    int y_begin, y_end;    // output rows, out row 0 == y_begin
    int ksize;
    float sigma;
    hipStream_t stream;
};
int launch_filter(const FilterArgs& a);
// argument check shared by every entry that takes a filter (kernel sizes, sigma, minimum image size)
int check_filter(int filter, int w, int h, int ksize, float sigma);
// rows of halo per side; w: image width in pixels (matters only for images narrower than the radius)
int filter_halo_rows(int filter, int ksize, int w = 1 << 30);
// gray -> Gaussian 5x5 -> median 3x3 -> Sobel in one kernel (whole frames).
int launch_chain(const void* in, void* out, int w, int h, int batch, float sigma, hipStream_t s);

// generateGaussianKernal restated for the library (host code; imageTools.cu:362-385).
void gaussian_kernel_2d(int k, float sigma, float* out);
// Normalised 1-D weights whose outer product is the 2-D kernel (up to rounding).
void gaussian_kernel_1d(int k, float sigma, float* out);

int png_write_rgba_stb(const char* filename, const unsigned char* px, int w, int h);     // stb_write.cpp; 0 on failure
int launch_copy(const void* src, void* dst, size_t bytes, hipStream_t s);
int launch_synth(void* out, uint64_t seed, uint64_t first, uint64_t count, hipStream_t s);
int launch_sobel_table(void* out, uint32_t count, hipStream_t s);
int launch_sobel_pairs(void* out, int gmax, int variant, hipStream_t s);

// Host tier (host.hip): run `n_ops` filters over a host image through the staging pipeline;
// chunk_rows_hint <= 0 picks the chunk height.  host_release frees the cached buffers.
int host_run_ops(const uint8_t* in, uint8_t* out, int w, int h, const mi_op* ops, int n_ops,
                 int chunk_rows_hint);
int host_run_ops_rows(const uint8_t* in, uint8_t* out, int w, int h, const mi_op* ops, int n_ops,
                      int chunk_rows_hint, int y0, int y1);
// one process, several devices (one host thread each): frames / row strips of a host image
int host_batch_run(const uint8_t* in, uint8_t* out, int w, int h, int n_frames, const mi_op* ops, int n_ops,
                   const int* devices, int n_devices);
int host_strip_run(const uint8_t* in, uint8_t* out, int w, int h, const mi_op* ops, int n_ops, const int* devices,
                   int n_devices);
int host_release();
// Blocking whole-buffer copies between pageable host memory and the device, staged through pinned chunks.
int host_repack(const void* src, void* dst, size_t bytes);      // host -> device -> repack kernel -> host, chunked
int host_upload(void* dev, const void* host, size_t bytes);
int host_download(void* host, const void* dev, size_t bytes);
// Device scratch of at least `bytes` for a blocking host-tier call, kept between calls (freed by
// host_release).  The holder of `lock` owns the buffer: host-tier spectral calls serialise on it.
struct HostScratch {
    std::unique_lock<std::mutex> lock;
    void* ptr = nullptr;
    // What the chirp-z tables at `tables_at` inside the buffer were built for (all zero: nothing): the
    // host-tier spectral calls are the only writers of the buffer, so tables built by one call are
    // still there for the next one of the same geometry.  Points into the cache; valid while `lock` is held.
    struct TableKey { const void* at; int rows, cols; };
    TableKey* tables = nullptr;
};
int host_scratch(size_t bytes, HostScratch* out);

// FFT path
// host-tier fusions of the pointwise spectral steps (bit-identical to the separate kernels)
int launch_highpass_unshifted(void* d, int w, int h, int planes, float cutoff, hipStream_t s);
int launch_complex_to_u8_normalized(const void* r, const void* g, const void* b, void* out, int w, int h, int batch, hipStream_t s);
int launch_u8_to_complex(const void* in, void* r, void* g, void* b, int w, int h, hipStream_t s, int batch = 1);
// tmp: optional scratch of one plane (rows*cols*8 bytes) that lets a tall plane (rows >= 1024) be
// transformed in place at full speed; may be null.
int launch_fft2d(const void* in, void* out, int rows, int cols, int batch, int dir, hipStream_t s,
                 void* tmp = nullptr);
// Any sizes (powers of two up to MI_FFT_MAX_POW2, anything else up to MI_FFT_MAX_ANY) with a caller workspace of
// fft2d_workspace_bytes(rows, cols) bytes (0 when both are powers of two).
int fft2d_size_supported(int rows, int cols);
size_t fft2d_workspace_bytes(int rows, int cols);
// tables_ready: the workspace still holds the chirp tables an earlier call built for these sizes
int launch_fft2d_any(const void* in, void* out, int rows, int cols, int batch, int dir, void* ws,
                     hipStream_t s, bool tables_ready = false);
// the whole of imageFFTImageGenerate's device core in four passes over HBM (fft.hip; 4096 x 4096 today)
bool fft_highpass_fused_supported(int rows, int cols);
int launch_fft_highpass_fused(const void* d_in, void* d_out, int w, int h, int batch, int rows, int cols, float cutoff,
                              void* a, void* b, hipStream_t s);
// the two halves separately (an edit of the spectra in `b` goes between them; cutoff <= 0: no filter)
int launch_spectral_fused_forward(const void* d_in, int w, int h, int batch, int rows, int cols, float cutoff,
                                  void* a, void* b, hipStream_t s);
int launch_spectral_fused_inverse(void* d_out, int w, int h, int batch, int rows, int cols, void* a, void* b, hipStream_t s);
int launch_fftshift(void* d, int w, int h, hipStream_t s);
int launch_highpass(void* d, int w, int h, float cutoff, hipStream_t s);
int launch_normalize_zero_imag(void* d, int w, int h, hipStream_t s);
int launch_complex_to_u8(const void* r, const void* g, const void* b, void* out, int w, int h,
                         hipStream_t s);
// Zero all but the `keep` coefficients of largest magnitude (blocking; ties at the threshold stay).
int spectrum_keep_top(void* data, size_t n, size_t keep, int planes, hipStream_t s);
int launch_minmax(const void* img, size_t n, void* minmax, hipStream_t s);
int launch_normalize_to_u8(const void* img, void* out, size_t n, const void* minmax,
                           hipStream_t s);

}  // namespace mi
