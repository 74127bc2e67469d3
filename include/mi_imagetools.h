/*
 * mi_imagetools.h -- C ABI of the MI355X (gfx950) image-filter library.
 *
 * This is the drop-in boundary for ctaylo41/CudaImageProcessing's L2 wrapper API
 * (reference kernels/imageTools.h:10-19): every entry point below names the reference
 * function it replaces.  The C++ header include/imageTools.h keeps the reference's own
 * signatures as thin shims over these functions.
 *
 * Pixel format everywhere: RGBA, 4 bytes per pixel, row-major, pitch = width*4, no padding
 * (the reference's uchar4 image, kernels/imageTools.cu:107-119; main.cpp:14).
 *
 * Two tiers:
 *   host tier    mi_*(...)       host pointers, blocking, reference wrapper semantics
 *                                (allocate, H2D, kernel, D2H -- kernels/imageTools.cu:233-253);
 *                                `out` may alias `in` (main.cpp:38-43 always passes (img,img)).
 *   device tier  mi_*_dev(...)   device pointers, asynchronous on `stream` (a hipStream_t
 *                                passed as void*, NULL = the default stream); `out` must not
 *                                overlap `in`.  This is the tier the benchmarks time.
 *
 * Return value: 0 (MI_OK) on success; negative MI_ERR_* for argument errors (nothing is
 * written); positive = the hipError_t that failed.  mi_last_error() describes the most recent
 * failure on the calling thread.  (The reference returns void and asserts on CUDA errors,
 * kernels/imageTools.cu:97-105; the C++ shims keep that behaviour.)
 *
 * Border convention: the stencil filters write their result only on
 * {1<=x<=W-2, 1<=y<=H-2} like the reference (imageTools.cu:260,291,317) and write RGBA
 * (0,0,0,0) elsewhere (the reference leaves fresh device memory there); neighbours are
 * addressed by the reference's flat index (y+ky)*W+(x+kx); a flat index outside the image
 * reads 0.
 */
#ifndef MI_IMAGETOOLS_H
#define MI_IMAGETOOLS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_ERR_INVALID_ARG (-1)   /* null pointer, non-positive size, W<3 or H<3 for a stencil */
#define MI_ERR_BAD_KERNEL_SIZE (-2) /* even, non-positive or too large kernel size */
#define MI_ERR_UNSUPPORTED (-3)   /* a shape this build has no kernel for */
#define MI_ERR_OVERLAP (-4)       /* device-tier in/out buffers overlap */
#define MI_ERR_IO (-5)            /* file could not be written */
#define MI_ERR_SELFTEST (-6)      /* mi_selftest found a wrong result */

#define MI_MAX_KERNEL_SIZE 31      /* the sizes with specialised kernels */
/* The reference takes any odd kernalSize (imageTools.cu:313-340, :478-502); so do the Gaussian and the mean
 * here, up to this bound.  Above MI_MAX_KERNEL_SIZE they run on generic kernels (mean: LDS tiles up to 49,
 * then one lane per pixel; Gaussian: one lane per pixel, and the call allocates a k*k weight table,
 * synchronises its stream and cannot be captured in a graph): correct, bit-exact, slow. */
#define MI_MAX_KERNEL_SIZE_ANY 255
/* 2-D transforms: each of rows, cols is in [2, MI_FFT_MAX_ANY], or a power of two up to
 * MI_FFT_MAX_POW2; a plane holds fewer than 2^28 points (2 GiB of interleaved fp32). */
#define MI_FFT_MAX_ANY 8192
#define MI_FFT_MAX_POW2 16384

/* ---------------------------------------------------------------- library */

int mi_version(void);
const char *mi_last_error(void);
/* Number of visible HIP devices (0 when there is none); never fails. */
int mi_device_count(void);

/* Tuning knobs (rows per band, FFT plane groups, ...; the list is in INTEGRATION.md).  Each has an
 * environment variable of the same name, read once per process on first use; this call changes
 * one at run time (tests and tuning sweeps; not meant to race with launches on other threads).
 * Debug aid, not in the reference. */
int mi_debug_set_tuning(const char *name, int value);
/* The knob's current value (whatever set it: the default, the environment, an earlier set). */
int mi_debug_get_tuning(const char *name, int *value);

/* ---------------------------------------------------------------- device tier
 * All device-tier filters process `batch` frames of w x h pixels stored back to back. */

/* imageLoad / imageWrite kernels (imageTools.cu:107-132): byte stream <-> uchar4, a copy. */
int mi_image_load_dev(const void *d_bytes, void *d_pixels, size_t n_pixels, void *stream);
int mi_image_write_dev(void *d_bytes, const void *d_pixels, int w, int h, void *stream);

/* imageGrayScale (imageTools.cu:217-231): every pixel, out=(g,g,g,A_in),
 * g = trunc(fma(.114,B, fma(.299,R, .587*G))) in fp32. */
int mi_grayscale_u8x4_dev(const void *d_in, void *d_out, int w, int h, int batch, void *stream);

/* imageSobelEdge (imageTools.cu:255-285): 3x3 Sobel on the .x channel, out=(m,m,m,255). */
int mi_sobel_u8x4_dev(const void *d_in, void *d_out, int w, int h, int batch, void *stream);

/* imageGaussianBlur + generateGaussianKernal (imageTools.cu:287-311, 362-385).
 * mode MI_GAUSS_SEPARABLE: row pass then column pass with 1-D weights (max |delta| <= 1 LSB
 *   against the reference's direct 2-D sum, which truncates float->u8, imageTools.cu:306-308);
 * mode MI_GAUSS_DIRECT: the reference's k x k sum in its own order, bit-exact, slower.
 * kernel_size: odd, 1..MI_MAX_KERNEL_SIZE_ANY; sizes above 31 always take the direct form, and so do separable
 * sizes 17..31 on images the matrix-core kernels do not take (width not a multiple of 4, buffers not 16-byte aligned).
 * (Separable sizes 7..31 on aligned images run both passes on the matrix cores, csrc/gauss_mfma.hip;
 * the same bytes for any band, strip or batch decomposition.  Round 2 ran 17..31 in the direct form whatever the
 * mode: MI_GAUSS_MFMA_MAX=15 brings that back.) */
#define MI_GAUSS_SEPARABLE 0
#define MI_GAUSS_DIRECT 1
int mi_gaussian_u8x4_dev(const void *d_in, void *d_out, int w, int h, int batch, int kernel_size,
                         float sigma, int mode, void *stream);

/* imageMeanBlur (imageTools.cu:313-340); the reference wrapper fixes kernel_size=3 (:806).  Any odd kernel_size up to
 * MI_MAX_KERNEL_SIZE_ANY, bit-exact (integer sums, exact division) whichever kernel runs: sizes 11..31 on aligned images
 * take the matrix cores (csrc/gauss_mfma.hip), the others the vector ALUs. */
int mi_meanblur_u8x4_dev(const void *d_in, void *d_out, int w, int h, int batch, int kernel_size,
                         void *stream);

/* Median blur, kernel_size 3 or 5.  README.md:3 of the reference advertises it; the reference
 * has no code for it, so its semantics are this build's (per-channel median of the k x k flat
 * taps on the same write set, alpha 255). */
int mi_medianblur_u8x4_dev(const void *d_in, void *d_out, int w, int h, int batch,
                           int kernel_size, void *stream);

/* The filter chain of BASELINE.json's headline config -- grayscale, Gaussian 5x5 (sigma), median
 * 3x3, Sobel -- fused into one pass over the frames.  Output is bit-identical to calling
 * mi_grayscale_u8x4_dev, mi_gaussian_u8x4_dev(5, sigma, MI_GAUSS_SEPARABLE),
 * mi_medianblur_u8x4_dev(3) and mi_sobel_u8x4_dev in turn (i.e. to chaining the reference's
 * imageGrayScaleWrapper, imageGaussianBlurWrapper, [median], imageSobelEdgeWrapper,
 * main.cpp:38-42, with the Gaussian's +-1 LSB caveat); the intermediate images are never
 * written.  Needs w >= 3 and h >= 3. */
int mi_chain_gray_gauss5_median3_sobel_u8x4_dev(const void *d_in, void *d_out, int w, int h,
                                                int batch, float sigma, void *stream);

/* Row-strip variants for one image of w x h_total pixels split over several devices.
 * d_in holds global rows [in_row0, in_row0+in_rows) (the strip plus its halo), d_out receives
 * global rows [y_begin, y_end) starting at its row 0.  The halo must cover rows
 * [y_begin-r-1, y_end+r+1) clipped to the image (r = kernel_size/2; the extra row is the
 * flat-index wrap, needed only for r>1).  filter: one of MI_FILTER_*.  param_i = kernel size
 * (ignored by gray/Sobel), param_f = sigma (Gaussian only). */
#define MI_FILTER_GRAYSCALE 0
#define MI_FILTER_SOBEL 1
#define MI_FILTER_GAUSSIAN 2
#define MI_FILTER_MEAN 3
#define MI_FILTER_MEDIAN 4
#define MI_FILTER_GAUSSIAN_DIRECT 5
int mi_filter_strip_u8x4_dev(int filter, const void *d_in, void *d_out, int w, int h_total,
                             int in_row0, int in_rows, int y_begin, int y_end, int param_i,
                             float param_f, void *stream);
/* Rows of halo a strip needs above and below for `filter` (0 for grayscale; r + 1 for a stencil of radius r > 1,
 * whose flat-index taps reach into the next row).  For an image narrower than the radius (w < r) the taps reach
 * ceil(r / w) rows further instead of one: such a strip needs r + ceil(r / w) rows, and mi_filter_strip_u8x4_dev
 * rejects a strip whose input rows do not cover them. */
int mi_filter_halo_rows(int filter, int param_i);

/* Synthetic input generator used by the benchmarks and tests (not in the reference):
 * pixel i of stream `seed` = low 32 bits of splitmix64(seed*0x9E3779B97F4A7C15 + i), alpha 255. */
int mi_synth_u8x4_dev(void *d_out, uint64_t seed, uint64_t first_idx, uint64_t count,
                      void *stream);

/* Sobel magnitude of s = Gx^2+Gy^2 for s in [0,count) as the device kernel computes it
 * (uint8 per entry); lets a test sweep every reachable s.  Debug aid, not in the reference. */
int mi_debug_sobel_mag_table_dev(void *d_out_u8, uint32_t count, void *stream);

/* The same for the fused chain's form of the magnitude, which takes (Gx, Gy) as a pair of f16:
 * out[(gy + gmax) * (2 gmax + 1) + gx + gmax] for gx, gy in [-gmax, gmax], gmax <= 1020 (the
 * largest |Gx|, |Gy| a Sobel of bytes can reach).  variant 0 = the one the chain uses.
 * Debug aid, not in the reference. */
int mi_debug_sobel_mag_pairs_dev(void *d_out_u8, int gmax, int variant, void *stream);

/* ---------------------------------------------------------------- FFT path (device tier)
 * Complex planes are interleaved fp32 (re,im), `rows` x `cols`, row-major (cufftComplex). */

/* uchar4ToCufftComplex (imageTools.cu:506-515): three planes (R,G,B), each (float(c), 0). */
int mi_u8_to_complex_dev(const void *d_in, void *d_red, void *d_green, void *d_blue, int w, int h,
                         void *stream);
/* cufftExecC2C (imageTools.cu:690-701, 741-751): unnormalised 2-D complex DFT, out of place or
 * in place, `batch` planes back to back.  direction -1 forward, +1 inverse.  rows and cols must
 * be powers of two in [2, MI_FFT_MAX_POW2] for this entry point (no workspace). */
#define MI_FFT_FORWARD (-1)
#define MI_FFT_INVERSE (+1)
int mi_fft2d_c2c_f32_dev(const void *d_in, void *d_out, int rows, int cols, int batch,
                         int direction, void *stream);
/* cufftPlan2d + cufftExecC2C for any size (the reference plans whatever the image measures; its
 * demo input is 675x900, main.cpp:34-45): each of rows, cols is in [2, MI_FFT_MAX_ANY], or a power
 * of two up to MI_FFT_MAX_POW2.  Lengths that are not powers of two go through Bluestein's chirp-z transform and
 * need device scratch: mi_fft2d_workspace_bytes gives the size, the caller owns the buffer (cuFFT
 * keeps its work area inside the plan).  For powers of two the workspace is optional: the size is
 * 0 below 1024 rows and one plane (rows*cols*8) from 1024 rows on, where it lets an IN-PLACE
 * transform run as fast as an out-of-place one (without it, in place, tall planes take a slower
 * column pass).  Calls sharing one workspace must be stream-ordered. */
int mi_fft2d_workspace_bytes(int rows, int cols, size_t *bytes);
int mi_fft2d_c2c_f32_ws_dev(const void *d_in, void *d_out, int rows, int cols, int batch,
                            int direction, void *d_workspace, size_t workspace_bytes, void *stream);
/* fftShift (imageTools.cu:564-583), in place. */
int mi_fftshift_dev(void *d_data, int w, int h, void *stream);
/* highpassFilter (imageTools.cu:530-549), in place, on a shifted spectrum. */
int mi_highpass_dev(void *d_data, int w, int h, float cutoff, void *stream);
/* NormalizeAndZeroImaginary (imageTools.cu:551-562), in place. */
int mi_normalize_zero_imag_dev(void *d_data, int w, int h, void *stream);
/* ComplexRGBToUchar (imageTools.cu:517-527) with the cast defined as saturate. */
int mi_complex_to_u8_dev(const void *d_red, const void *d_green, const void *d_blue, void *d_out,
                         int w, int h, void *stream);
/* findMinMax (imageTools.cu:416-454): d_minmax receives {min, max} (2 floats). */
int mi_minmax_f32_dev(const void *d_img, size_t n, void *d_minmax, void *stream);
/* floatToUchar4 (imageTools.cu:458-471) after scaling by (v-min)/(max-min)*255. */
int mi_normalize_to_u8_dev(const void *d_img, void *d_out, size_t n, const void *d_minmax,
                           void *stream);

/* ---------------------------------------------------------------- host tier
 * Host pointers, blocking; replaces the reference wrapper of the same name. */

/* imageLoadWrapper (imageTools.cu:158-183) */
int mi_image_load(const uint8_t *bytes, uint8_t *pixels, size_t n_pixels);
/* imageWriteWrapper (imageTools.cu:134-156): always writes a PNG, whatever the extension. */
int mi_image_write(const char *filename, const uint8_t *pixels, int w, int h);
/* imageGrayScaleWrapper (imageTools.cu:233-253) */
int mi_grayscale_u8x4(const uint8_t *in, uint8_t *out, int w, int h);
/* imageSobelEdgeWrapper (imageTools.cu:342-360) */
int mi_sobel_u8x4(const uint8_t *in, uint8_t *out, int w, int h);
/* imageGaussianBlurWrapper (imageTools.cu:478-502) */
int mi_gaussian_u8x4(const uint8_t *in, uint8_t *out, int w, int h, int kernel_size, float sigma,
                     int mode);
/* imageMeanBlurWrapper (imageTools.cu:794-812) with its hard-coded 3 as a parameter */
int mi_meanblur_u8x4(const uint8_t *in, uint8_t *out, int w, int h, int kernel_size);
/* Median blur (README.md:3; no reference code) */
int mi_medianblur_u8x4(const uint8_t *in, uint8_t *out, int w, int h, int kernel_size);
/* A list of filters run back to back on one host image -- what the reference's demo does with
 * consecutive wrapper calls on (img, img) (main.cpp:38-43) -- in ONE trip over PCIe: the image is
 * cut into row chunks that are staged through pinned memory, copied, filtered by every stage on
 * the device and copied back on alternating streams, overlapped with the host-side staging of
 * the neighbouring chunks.  `out` may be `in`.  The result is bit-identical to calling the
 * single-filter wrappers one after the other.  sigma is read by MI_FILTER_GAUSSIAN(_DIRECT)
 * only, ksize by the Gaussian, mean and median.  The single-filter wrappers above run through
 * the same pipeline.  Staging buffers are cached between calls (host-tier calls of one process
 * serialise on them); mi_host_release frees them. */
#define MI_MAX_OPS 16
typedef struct mi_op {
    int filter;  /* MI_FILTER_* */
    int ksize;
    float sigma;
} mi_op;
int mi_run_ops_u8x4(const uint8_t *in, uint8_t *out, int w, int h, const mi_op *ops, int n_ops);
/* The same with the chunk height chosen by the caller (tests; tuning). */
int mi_run_ops_chunked_u8x4(const uint8_t *in, uint8_t *out, int w, int h, const mi_op *ops,
                            int n_ops, int chunk_rows);
/* Several GPUs behind the C ABI, one process (no counterpart upstream: the reference is single-GPU;
 * BASELINE.json configs 3 and 4).  One host thread per entry of `devices` (HIP ordinals; an ordinal
 * may repeat), each driving its device's own staging pipeline; nothing is exchanged between devices
 * because every halo row comes from the host image.  The pixels are those of mi_run_ops_u8x4.
 *   mi_batch_run_ops_u8x4  n_frames images of w x h back to back; device i takes the contiguous
 *                          frames [i*ceil(n/G), (i+1)*ceil(n/G)); out may be in.
 *   mi_strip_run_ops_u8x4  one image; device i filters rows [h*i/G, h*(i+1)/G) (its chunks read
 *                          the accumulated halo of the list from `in`); out must not be in when G > 1.
 * The calling thread's current device is unchanged on return.  (Device-resident strips with an RCCL
 * halo exchange are cudaimageprocessing_amd/shard.py over mi_filter_strip_u8x4_dev.) */
/* Device-RESIDENT row strips, one process (no counterpart upstream; BASELINE.json config 4 for a C++ caller):
 * strip i of a w x h image -- rows [y0, y1) = mi_strip_rows(h, G, i) -- lives on devices[i] in the middle of a
 * device buffer of (pad_rows + rows + pad_rows) image rows, its first row at row pad_rows.  One call = one filter
 * over every strip: each device's copy stream brings the neighbours' edge rows straight into the pad rows
 * (hipMemcpyPeerAsync, point to point, no host copy), its main stream filters the interior rows meanwhile, then
 * waits for the copies on an event and filters the two edge bands.  d_in[i] / d_out[i] are the strip buffers
 * on devices[i] (they must differ; swap them between the stages of a chain); pad_rows >=
 * mi_filter_halo_rows(filter, param_i).  Blocking: on return every device has finished and d_out[i] holds the
 * filtered strip (its pad rows are undefined).  Output == the single-device call on the whole image, bit for
 * bit.  An ordinal may repeat (its strips then take turns on that device).  The RCCL form of the same
 * decomposition, one process per GPU, is cudaimageprocessing_amd/shard.py. */
int mi_strip_rows(int h, int n_devices, int i, int *y0, int *y1);
int mi_strip_stage_u8x4_dev(void *const *d_in, void *const *d_out, int w, int h, const int *devices,
                            int n_devices, int pad_rows, int filter, int param_i, float param_f);
int mi_batch_run_ops_u8x4(const uint8_t *in, uint8_t *out, int w, int h, int n_frames, const mi_op *ops,
                          int n_ops, const int *devices, int n_devices);
int mi_strip_run_ops_u8x4(const uint8_t *in, uint8_t *out, int w, int h, const mi_op *ops, int n_ops,
                          const int *devices, int n_devices);
int mi_host_release(void);
/* imageFFTImageGenerate (imageTools.cu:633-791) without its debug means; cutoff is the
 * reference's hard-coded 10.0f (:716).  The transforms use the reference's plan geometry,
 * cufftPlan2d(&plan, width, height, ...) (:688): the buffer is transformed as `w` rows of `h`
 * points while the shift and the filter index it as `h` rows of `w` -- the same thing for a
 * square image, and what reproduces the reference's own sample output for the 675x900 one
 * (tests/golden/).  Sizes: w, h in [2, MI_FFT_MAX_ANY], or powers of two up to MI_FFT_MAX_POW2. */
int mi_fft_highpass_u8x4(const uint8_t *in, uint8_t *out, int w, int h, float cutoff);
/* The same on device images (d_in and d_out may be the same buffer), asynchronous on `stream`,
 * with the caller's scratch: three planes, three spectra and the transform workspace,
 * mi_fft_highpass_workspace_bytes(w, h) bytes (about 13 images' worth for a power-of-two size,
 * 20 for the chirp-z sizes).  Allocates nothing: usable under stream capture like the filters. */
int mi_fft_highpass_workspace_bytes(int w, int h, size_t *bytes);
int mi_fft_highpass_u8x4_dev(const void *d_in, void *d_out, int w, int h, float cutoff,
                             void *d_workspace, size_t workspace_bytes, void *stream);
/* No counterpart in the reference: mi_fft_highpass_u8x4_dev for callers that cannot allocate (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc) and so have no
 * workspace to bring.  The scratch is a static device array inside the library, MI_STATIC_SCRATCH_BYTES
 * per GPU, held until the process ends; the call allocates nothing (no hipMalloc / hipFree, no
 * allocating library routine) and its output is byte-identical to mi_fft_highpass_u8x4_dev's.
 * Sizes whose workspace (mi_fft_highpass_workspace_bytes) exceeds the static buffer return
 * MI_ERR_UNSUPPORTED: the limit covers the reference's 675x900 demo image and powers of two up to
 * 1024x1024.  All calls on one GPU share that GPU's buffer: calls that overlap in time on one GPU
 * (two streams, two threads) race; stream-ordered calls are safe.  The first call on a device looks
 * the buffer's address up (and fills the FFT twiddle table): make it outside stream capture. */
#define MI_STATIC_SCRATCH_BYTES (64u << 20)
int mi_fft_highpass_u8x4_static_dev(const void *d_in, void *d_out, int w, int h, float cutoff,
                                    void *stream);
/* A batch of `batch` images of one size, back to back (the reference handles one image per call:
 * no counterpart upstream; each image comes out as mi_fft_highpass_u8x4_dev gives it).  All 3*batch
 * colour planes go through every step together, which is what makes small images cheap. */
int mi_fft_highpass_batch_workspace_bytes(int w, int h, int batch, size_t *bytes);
int mi_fft_highpass_batch_u8x4_dev(const void *d_in, void *d_out, int w, int h, int batch, float cutoff,
                                   void *d_workspace, size_t workspace_bytes, void *stream);
/* compressImage (declared at kernels/imageTools.h:17 and commented out at main.cpp:44, never
 * defined upstream; defined by this library on the round-trip scaffold of imageFFTImageGenerate):
 * per colour channel, forward 2-D DFT, keep the ceil(keep_fraction * w * h) coefficients of
 * largest magnitude (ties at the threshold all stay) and zero the rest, inverse DFT, /N, real
 * part saturated to u8, alpha 255.  keep_fraction in (0, 1]; 1 reproduces the input. */
#define MI_COMPRESS_DEFAULT_KEEP 0.10f
int mi_compress_u8x4(const uint8_t *in, uint8_t *out, int w, int h, float keep_fraction);
/* The selection step of mi_compress_u8x4 on a device spectrum of n complex values, in place.
 * Blocking (the threshold search reads three histograms back to the host and waits for `stream`
 * each time); allocates nothing (its histograms are a static 32 KiB device array; calls on one
 * device take turns on it). */
int mi_spectrum_keep_top_dev(void *d_data, size_t n, size_t keep, void *stream);
/* bool test() (declared at kernels/imageTools.h:19, never defined upstream): the library's
 * self-test -- known-answer checks of every filter, the transforms and the host pipeline on
 * the current device.  Returns MI_OK, or MI_ERR_SELFTEST with the first failure in
 * mi_last_error(); every failure is also printed to stderr. */
int mi_selftest(void);
/* generateGaussianKernal (imageTools.cu:362-385): fills out[kernel_size^2] on the host. */
int mi_gaussian_kernel(int kernel_size, float sigma, float *out);

#ifdef __cplusplus
}
#endif
#endif /* MI_IMAGETOOLS_H */
