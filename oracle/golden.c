/*
 * oracle/golden.c -- CPU restatement of ctaylo41/CudaImageProcessing's image filters.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (cudaimageprocessing_amd/, include/,
 * the C-ABI library) links, imports or calls this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for any of these
 * functions (its test/test_imageTools.cu is a 0-byte file) and it is CUDA-only, so it can be
 * neither compiled nor run in the authoring container (no nvcc, no GPU; its one translation
 * unit includes cuda_runtime.h and cufft.h).  This file therefore transcribes the arithmetic
 * of kernels/imageTools.cu line by line (citations below) and is pinned only by hand-computed
 * known-answer tests in tests/test_oracle.py.
 *
 * Pixel format: RGBA, 4 bytes per pixel (.x=R .y=G .z=B .w=A), row-major, pitch = width*4,
 * idx = y*width + x  (kernels/imageTools.cu:107-119, main.cpp:14).
 *
 * Conventions fixed here where the reference leaves behaviour undefined:
 *  - pixels outside the reference's write set {1<=x<=W-2, 1<=y<=H-2} (imageTools.cu:260,291,317)
 *    are written as RGBA (0,0,0,0): the reference leaves them as fresh device memory;
 *  - neighbours are addressed by the reference's flat index (y+ky)*W + (x+kx) with no clamping
 *    (imageTools.cu:272,298,324), so for radius > 1 edge columns read the adjacent row;
 *  - a tap whose flat index falls outside [0, W*H) (undefined behaviour upstream) reads RGBA 0;
 *  - all index math is 64-bit.
 *
 * Every row-parallel function takes a row range [y0, y1): it writes exactly those output rows
 * of the full-size output, reading the full-size input.  Results are independent of how the
 * rows are split, so callers may run bands on several threads.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  -ffp-contract=off matters:
 * every fused multiply-add below is spelled fmaf() explicitly and nothing else may be fused.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { uint8_t x, y, z, w; } px4;

static inline px4 tap(const uint8_t *in, int64_t f, int64_t n)
{
    px4 p = {0, 0, 0, 0};
    if (f >= 0 && f < n) memcpy(&p, in + 4 * f, 4);
    return p;
}

static inline void put(uint8_t *out, int64_t f, uint8_t x, uint8_t y, uint8_t z, uint8_t w)
{
    out[4 * f + 0] = x; out[4 * f + 1] = y; out[4 * f + 2] = z; out[4 * f + 3] = w;
}

static inline int in_write_set(int x, int y, int w, int h)
{   /* imageTools.cu:260 / :291 / :317 -- a 1-pixel guard whatever the kernel size */
    return x > 0 && y > 0 && x < w - 1 && y < h - 1;
}

/* ------------------------------------------------------------------ synthetic input */

/* Counter-based generator shared by the oracle, the numpy mirror and the device generator
 * (SURVEY.md 8d): pixel idx of stream `seed` = low 32 bits of splitmix64(seed*GOLDEN + idx),
 * alpha forced to 255. */
static inline uint32_t synth_px(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ULL + idx + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (uint32_t)z | 0xFF000000u;
}

void gold_synth_u8x4(uint8_t *out, uint64_t seed, uint64_t first_idx, uint64_t count)
{
    for (uint64_t i = 0; i < count; i++) {
        uint32_t p = synth_px(seed, first_idx + i);
        memcpy(out + 4 * i, &p, 4);
    }
}

/* ------------------------------------------------------------------ a6: load / write repack */

/* imageLoad, imageTools.cu:107-119: out[i] = {in[4i], in[4i+1], in[4i+2], in[4i+3]}.
 * imageWrite, imageTools.cu:121-132: the inverse.  Both are byte-identical copies. */
void gold_image_load(const uint8_t *bytes, uint8_t *pixels, uint64_t n_px)
{
    for (uint64_t i = 0; i < n_px; i++) {
        pixels[4 * i + 0] = bytes[i * 4 + 0];
        pixels[4 * i + 1] = bytes[i * 4 + 1];
        pixels[4 * i + 2] = bytes[i * 4 + 2];
        pixels[4 * i + 3] = bytes[i * 4 + 3];
    }
}

void gold_image_write(uint8_t *bytes, const uint8_t *pixels, int w, int h)
{
    uint64_t n = (uint64_t)w * (uint64_t)h;
    for (uint64_t i = 0; i < n; i++) {
        bytes[i * 4 + 0] = pixels[4 * i + 0];
        bytes[i * 4 + 1] = pixels[4 * i + 1];
        bytes[i * 4 + 2] = pixels[4 * i + 2];
        bytes[i * 4 + 3] = pixels[4 * i + 3];
    }
}

/* ------------------------------------------------------------------ a1: grayscale */

/* imageGrayScale, imageTools.cu:217-231.  Every pixel (no border skip, :221).
 * gray = (unsigned char)(0.299f*R + 0.587f*G + 0.114f*B), fp32, left to right, C truncation.
 * nvcc's default -fmad=true may contract the expression; which contraction it picks cannot be
 * observed without nvcc.  Pinned form: t = 0.587f*G; u = fmaf(0.299f,R,t);
 * v = fmaf(0.114f,B,u) -- LLVM's contraction of fadd(fmul,fmul) fuses the left multiply.
 * Output (g,g,g,A_in), :226-229. */
static inline uint8_t gray_fused(uint8_t r, uint8_t g, uint8_t b)
{
    float t = 0.587f * (float)g;
    float u = fmaf(0.299f, (float)r, t);
    float v = fmaf(0.114f, (float)b, u);
    return (uint8_t)v;
}

static inline uint8_t gray_unfused(uint8_t r, uint8_t g, uint8_t b)
{   /* the same expression with no contraction at all (what -fmad=false would give) */
    float v = (0.299f * (float)r + 0.587f * (float)g) + 0.114f * (float)b;
    return (uint8_t)v;
}

void gold_grayscale(const uint8_t *in, uint8_t *out, int w, int h, int y0, int y1)
{
    (void)h;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < w; x++) {
            int64_t f = (int64_t)y * w + x;
            const uint8_t *p = in + 4 * f;
            uint8_t g = gray_fused(p[0], p[1], p[2]);
            put(out, f, g, g, g, p[3]);
        }
}

/* Exhaustive table over all 2^24 (R,G,B): out[(r<<16)|(g<<8)|b].  fused!=0 selects the pinned
 * form, 0 the un-contracted one (used to document how far the two can be apart). */
void gold_grayscale_table(uint8_t *out, int fused)
{
    for (uint32_t r = 0; r < 256; r++)
        for (uint32_t g = 0; g < 256; g++)
            for (uint32_t b = 0; b < 256; b++)
                out[(r << 16) | (g << 8) | b] =
                    fused ? gray_fused((uint8_t)r, (uint8_t)g, (uint8_t)b)
                          : gray_unfused((uint8_t)r, (uint8_t)g, (uint8_t)b);
}

/* ------------------------------------------------------------------ a2: Sobel */

/* imageSobelEdge, imageTools.cu:255-285.  Write set :260.  Uses the .x channel only (:274-275).
 * Gx, Gy accumulate in float (exact: integers < 2^24), magnitude = sqrt(Gx*Gx+Gy*Gy) (:278),
 * clamp to [0,255] (:279), implicit float -> unsigned char truncation (:280-282), alpha 255. */
void gold_sobel(const uint8_t *in, uint8_t *out, int w, int h, int y0, int y1)
{
    static const int sobelX[3][3] = {{-1, 0, 1}, {-2, 0, 2}, {-1, 0, 1}};
    static const int sobelY[3][3] = {{-1, -2, -1}, {0, 0, 0}, {1, 2, 1}};
    int64_t n = (int64_t)w * h;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < w; x++) {
            int64_t o = (int64_t)y * w + x;
            if (!in_write_set(x, y, w, h)) { put(out, o, 0, 0, 0, 0); continue; }
            float Gx = 0, Gy = 0;
            for (int ky = -1; ky <= 1; ky++)
                for (int kx = -1; kx <= 1; kx++) {
                    px4 p = tap(in, (int64_t)(y + ky) * w + (x + kx), n);
                    Gx += (float)(p.x * sobelX[ky + 1][kx + 1]);
                    Gy += (float)(p.x * sobelY[ky + 1][kx + 1]);
                }
            float m = sqrtf(Gx * Gx + Gy * Gy);
            m = fmaxf(0.0f, fminf(255.0f, m));
            uint8_t v = (uint8_t)m;
            put(out, o, v, v, v, 255);
        }
}

/* trunc(clamp(sqrtf(s))) for one s = Gx^2+Gy^2; lets a test prove the integer form
 * min(255, isqrt(s)) used by the device kernel equal over the whole reachable range. */
int gold_sobel_mag(uint32_t s)
{
    float m = sqrtf((float)s);
    m = fmaxf(0.0f, fminf(255.0f, m));
    return (int)(uint8_t)m;
}

/* ------------------------------------------------------------------ a3: Gaussian weights */

/* generateGaussianKernal, imageTools.cu:362-385.
 * value = exp(-(i*i+j*j)/(2*sigma*sigma)) / (2*M_PI*sigma*sigma), :371 -- the numerator is an
 * int converted to float, 2*sigma*sigma is float, exp resolves to the float overload under
 * nvcc (expf); the denominator is promoted to double by M_PI; the quotient narrows to float.
 * sum accumulates in float in i-major order (:373); every entry is then divided by sum (:381).
 * Only odd sizes are valid upstream (an even size overruns the buffer); callers reject them. */
void gold_gaussian_kernel(int size, float sigma, float *kernal)
{
    float sum = 0.0f;
    int half = size / 2;
    for (int i = -half; i <= half; i++)
        for (int j = -half; j <= half; j++) {
            float e = expf((float)(-(i * i + j * j)) / (2 * sigma * sigma));
            float value = (float)((double)e / (2 * M_PI * (double)sigma * (double)sigma));
            kernal[(i + half) * size + (j + half)] = value;
            sum += value;
        }
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++)
            kernal[i * size + j] /= sum;
}

/* ------------------------------------------------------------------ a4: Gaussian blur */

/* imageGaussianBlur, imageTools.cu:287-311.  Write set :291 (1-pixel guard for every k).
 * ky outer, kx inner, ascending (:294-296); sum.c += pixel.c * kernalValue (:301-303) in fp32
 * from 0, spelled fmaf(px, kv, sum) (nvcc contracts a*b+c); implicit float -> unsigned char
 * truncation on store (:306-308); alpha 255 (:309).  Flat taps, out-of-buffer taps read 0. */
void gold_gaussian(const uint8_t *in, uint8_t *out, int w, int h, int k, const float *kernal,
                   int y0, int y1)
{
    int64_t n = (int64_t)w * h;
    int r = k / 2;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < w; x++) {
            int64_t o = (int64_t)y * w + x;
            if (!in_write_set(x, y, w, h)) { put(out, o, 0, 0, 0, 0); continue; }
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;
            for (int ky = -r; ky <= r; ky++)
                for (int kx = -r; kx <= r; kx++) {
                    px4 p = tap(in, (int64_t)(y + ky) * w + (x + kx), n);
                    float kv = kernal[(ky + r) * k + (kx + r)];
                    sx = fmaf((float)p.x, kv, sx);
                    sy = fmaf((float)p.y, kv, sy);
                    sz = fmaf((float)p.z, kv, sz);
                }
            put(out, o, (uint8_t)sx, (uint8_t)sy, (uint8_t)sz, 255);
        }
}

/* ------------------------------------------------------------------ a5: mean (box) blur */

/* imageMeanBlur, imageTools.cu:313-340.  int3 sums over the k x k flat taps (:319-330),
 * C integer division by k*k (:332-334), alpha 255 (:338).  The reference wrapper fixes k=3
 * (:806); k is a parameter here so the kernel is covered for other odd sizes too. */
void gold_meanblur(const uint8_t *in, uint8_t *out, int w, int h, int k, int y0, int y1)
{
    int64_t n = (int64_t)w * h;
    int r = k / 2;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < w; x++) {
            int64_t o = (int64_t)y * w + x;
            if (!in_write_set(x, y, w, h)) { put(out, o, 0, 0, 0, 0); continue; }
            int mx = 0, my = 0, mz = 0;
            for (int ky = -r; ky <= r; ky++)
                for (int kx = -r; kx <= r; kx++) {
                    px4 p = tap(in, (int64_t)(y + ky) * w + (x + kx), n);
                    mx += p.x; my += p.y; mz += p.z;
                }
            put(out, o, (uint8_t)(mx / (k * k)), (uint8_t)(my / (k * k)), (uint8_t)(mz / (k * k)),
                255);
        }
}

/* ------------------------------------------------------------------ a5': median blur (NEW) */

/* The reference only advertises a median blur (README.md:3); there is no code to follow.
 * Defined by this build: on the same write set, per channel c in {x,y,z}, the median of the
 * k x k flat taps (k=3: 5th smallest of 9; k=5: 13th of 25); alpha 255; border 0. */
static void sort_u8(uint8_t *v, int n)
{   /* plain insertion sort: n is 9 or 25 */
    for (int i = 1; i < n; i++) {
        uint8_t key = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > key) { v[j + 1] = v[j]; j--; }
        v[j + 1] = key;
    }
}

void gold_medianblur(const uint8_t *in, uint8_t *out, int w, int h, int k, int y0, int y1)
{
    int64_t n = (int64_t)w * h;
    int r = k / 2, kk = k * k;
    uint8_t *v = (uint8_t *)malloc((size_t)3 * kk);
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < w; x++) {
            int64_t o = (int64_t)y * w + x;
            if (!in_write_set(x, y, w, h)) { put(out, o, 0, 0, 0, 0); continue; }
            int t = 0;
            for (int ky = -r; ky <= r; ky++)
                for (int kx = -r; kx <= r; kx++) {
                    px4 p = tap(in, (int64_t)(y + ky) * w + (x + kx), n);
                    v[t] = p.x; v[kk + t] = p.y; v[2 * kk + t] = p.z; t++;
                }
            sort_u8(v, kk);
            sort_u8(v + kk, kk);
            sort_u8(v + 2 * kk, kk);
            put(out, o, v[kk / 2], v[kk + kk / 2], v[2 * kk + kk / 2], 255);
        }
    free(v);
}

/* ------------------------------------------------------------------ a7..a13: FFT path */

/* uchar4ToCufftComplex, imageTools.cu:506-515 (call :659 swaps green/blue buffers, undone at
 * :781, so the net effect is three independent channel planes): plane_c[idx] = (float(c), 0). */
void gold_u8_to_complex(const uint8_t *in, float *red, float *green, float *blue, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        red[2 * i] = (float)in[4 * i + 0];   red[2 * i + 1] = 0.0f;
        green[2 * i] = (float)in[4 * i + 1]; green[2 * i + 1] = 0.0f;
        blue[2 * i] = (float)in[4 * i + 2];  blue[2 * i + 1] = 0.0f;
    }
}

/* Mixed-radix decimation in time for any n: n = p * m with p the smallest prime factor; the p
 * interleaved sub-sequences are transformed recursively and combined by the definition,
 *   X[k + m q] = sum_r W_n^(r (k + m q)) Y_r[k],   W_n = exp(sign * 2 pi i / n),
 * so a prime length is the O(n^2) definition itself.  (wr, wi)[j] = W_N^j for the top-level N;
 * a sub-transform of length n uses every (N/n)-th entry.  x is read with `stride`, y is dense;
 * `scratch` holds 2n doubles per recursion level (2N in total is enough: n + n/p + ... < 2n). */
static void dft_rec(const double *xr, const double *xi, int n, int stride, double *yr, double *yi,
                    const double *wr, const double *wi, int N, double *scratch)
{
    if (n == 1) { yr[0] = xr[0]; yi[0] = xi[0]; return; }
    int p = n;
    for (int f = 2; (int64_t)f * f <= n; f++) if (n % f == 0) { p = f; break; }
    int m = n / p, step = N / n;
    double *sr = scratch, *si = scratch + n;           /* Y_r[k] at [r*m + k] */
    if (p == n) {
        for (int r = 0; r < n; r++) { sr[r] = xr[(int64_t)r * stride]; si[r] = xi[(int64_t)r * stride]; }
    } else {
        for (int r = 0; r < p; r++)
            dft_rec(xr + (int64_t)r * stride, xi + (int64_t)r * stride, m, stride * p, sr + r * m, si + r * m,
                    wr, wi, N, scratch + 2 * n);
    }
    for (int q = 0; q < p; q++)
        for (int k = 0; k < m; k++) {
            double ar = 0, ai = 0;
            int64_t idx = k + (int64_t)m * q;
            for (int r = 0; r < p; r++) {
                int64_t e = ((int64_t)r * idx) % n * step;
                double cr = wr[e], ci = wi[e], vr = sr[r * m + k], vi = si[r * m + k];
                ar += vr * cr - vi * ci; ai += vr * ci + vi * cr;
            }
            yr[idx] = ar; yi[idx] = ai;
        }
}

/* 1-D DFT in double, in place on a strided line.  sign = -1 forward, +1 inverse, unnormalised --
 * cuFFT's convention (imageTools.cu:690,741).  tw = 2n doubles of twiddles for this n and sign,
 * work = 6n doubles. */
static void dft1d(double *re, double *im, int n, int stride, const double *tw, double *work)
{
    double *yr = work, *yi = work + n;
    dft_rec(re, im, n, stride, yr, yi, tw, tw + n, n, work + 2 * n);
    for (int i = 0; i < n; i++) { re[(int64_t)i * stride] = yr[i]; im[(int64_t)i * stride] = yi[i]; }
}

static double *dft_twiddles(int n, int sign)
{
    double *tw = (double *)malloc(sizeof(double) * 2 * n);
    for (int j = 0; j < n; j++) {
        double ang = sign * 2.0 * M_PI * (double)j / n;
        tw[j] = cos(ang); tw[n + j] = sin(ang);
    }
    return tw;
}

/* 2-D complex DFT of `rows` rows of `cols` elements, interleaved (re,im) double, in place.
 * The caller chooses the geometry: the reference's cufftPlan2d(width,height) (imageTools.cu:688)
 * is rows = width, cols = height (see gold_fft_highpass_image). */
void gold_fft2d_f64(double *data, int rows, int cols, int sign)
{
    int64_t n = (int64_t)rows * cols;
    int m = rows > cols ? rows : cols;
    double *re = (double *)malloc(sizeof(double) * n), *im = (double *)malloc(sizeof(double) * n);
    double *work = (double *)malloc(sizeof(double) * 6 * m);
    double *twc = dft_twiddles(cols, sign), *twr = dft_twiddles(rows, sign);
    for (int64_t i = 0; i < n; i++) { re[i] = data[2 * i]; im[i] = data[2 * i + 1]; }
    for (int y = 0; y < rows; y++) dft1d(re + (int64_t)y * cols, im + (int64_t)y * cols, cols, 1, twc, work);
    for (int x = 0; x < cols; x++) dft1d(re + x, im + x, rows, cols, twr, work);
    for (int64_t i = 0; i < n; i++) { data[2 * i] = re[i]; data[2 * i + 1] = im[i]; }
    free(re); free(im); free(work); free(twc); free(twr);
}

/* fp32 in / fp32 out wrapper: the transform itself is evaluated in double and rounded once. */
void gold_fft2d_c2c(const float *in, float *out, int rows, int cols, int sign)
{
    int64_t n = (int64_t)rows * cols;
    double *d = (double *)malloc(sizeof(double) * 2 * n);
    for (int64_t i = 0; i < 2 * n; i++) d[i] = in[i];
    gold_fft2d_f64(d, rows, cols, sign);
    for (int64_t i = 0; i < 2 * n; i++) out[i] = (float)d[i];
    free(d);
}

/* fftShift, imageTools.cu:564-583: for x<W/2, y<H/2 swap (x,y)<->(x+W/2,y+H/2) and
 * (x+W/2,y)<->(x,y+H/2) in place.  For odd W or H the middle column/row is left alone. */
void gold_fftshift(float *data, int w, int h)
{
    for (int y = 0; y < h / 2; y++)
        for (int x = 0; x < w / 2; x++) {
            int64_t i1 = (int64_t)y * w + x, i2 = (int64_t)(y + h / 2) * w + (x + w / 2);
            int64_t i3 = (int64_t)y * w + (x + w / 2), i4 = (int64_t)(y + h / 2) * w + x;
            float a, b;
            a = data[2 * i1]; b = data[2 * i1 + 1];
            data[2 * i1] = data[2 * i2]; data[2 * i1 + 1] = data[2 * i2 + 1];
            data[2 * i2] = a; data[2 * i2 + 1] = b;
            a = data[2 * i3]; b = data[2 * i3 + 1];
            data[2 * i3] = data[2 * i4]; data[2 * i3 + 1] = data[2 * i4 + 1];
            data[2 * i4] = a; data[2 * i4 + 1] = b;
        }
}

/* highpassFilter, imageTools.cu:530-549: cx = x - W/2.0f; cy = y - H/2.0f;
 * d = sqrtf(cx*cx+cy*cy); h = 1 - expf(-(d*d)/(2*cutoff*cutoff)); re *= h; im *= h. */
float gold_highpass_gain(int x, int y, int w, int h, float cutoff)
{
    float cx = (float)x - (float)w / 2.0f;
    float cy = (float)y - (float)h / 2.0f;
    float distance = sqrtf(cx * cx + cy * cy);
    return 1.0f - expf(-(distance * distance) / (2.0f * cutoff * cutoff));
}

void gold_highpass(float *data, int w, int h, float cutoff)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int64_t i = (int64_t)y * w + x;
            float g = gold_highpass_gain(x, y, w, h, cutoff);
            data[2 * i] *= g; data[2 * i + 1] *= g;
        }
}

/* NormalizeAndZeroImaginary, imageTools.cu:551-562: re = re / (W*H) (int product -> float), im = 0. */
void gold_normalize_zero_imag(float *data, int w, int h)
{
    int64_t n = (int64_t)w * h;
    float div = (float)(w * h);
    for (int64_t i = 0; i < n; i++) { data[2 * i] = data[2 * i] / div; data[2 * i + 1] = 0.0f; }
}

/* ComplexRGBToUchar, imageTools.cu:517-527: out = ((uchar)red.re, (uchar)green.re, (uchar)blue.re, 255)
 * after the buffer-name swap cancels.  The reference's cast of negative or >255 floats is
 * undefined; the build defines it as saturate: clamp(trunc(v), 0, 255) (NaN -> 0). */
static inline uint8_t sat_u8(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

void gold_complex_to_u8(const float *red, const float *green, const float *blue, uint8_t *out,
                        int64_t n)
{
    for (int64_t i = 0; i < n; i++)
        put(out, i, sat_u8(red[2 * i]), sat_u8(green[2 * i]), sat_u8(blue[2 * i]), 255);
}

/* imageFFTImageGenerate, imageTools.cu:633-791, minus the debug means: per channel
 * u8 -> complex -> fft2d -> shift -> high-pass(cutoff) -> shift -> ifft2d -> /N, im=0 -> u8.
 * The plan is cufftPlan2d(&plan, width, height, CUFFT_C2C) (:688), and cufftPlan2d's first size
 * is the slowest-varying dimension: the transform treats the W*H buffer as `width` rows of `height`
 * points, while fftShift and highpassFilter index it as `height` rows of `width` (:535, :569-572).
 * For a square image the two agree; for any other it is this reshaped transform that the
 * reference computes, and its own sample output (image/output.jpg, made from the 675x900
 * image/lord.jpeg) is reproduced only this way (tests/test_oracle.py, tests/test_gpu_golden.py).
 * `real_out` (3 planes of W*H floats, may be NULL) receives the real parts before the cast:
 * that float image is what the device path is compared against (rel-L2). */
void gold_fft_highpass_image(const uint8_t *in, uint8_t *out, float *real_out, int w, int h,
                             float cutoff)
{
    int64_t n = (int64_t)w * h;
    float *pl[3];
    for (int c = 0; c < 3; c++) pl[c] = (float *)malloc(sizeof(float) * 2 * n);
    gold_u8_to_complex(in, pl[0], pl[1], pl[2], n);
    for (int c = 0; c < 3; c++) {
        gold_fft2d_c2c(pl[c], pl[c], w, h, -1);     /* rows = width, cols = height: see above */
        gold_fftshift(pl[c], w, h);
        gold_highpass(pl[c], w, h, cutoff);
        gold_fftshift(pl[c], w, h);
        gold_fft2d_c2c(pl[c], pl[c], w, h, +1);
        gold_normalize_zero_imag(pl[c], w, h);
        if (real_out)
            for (int64_t i = 0; i < n; i++) real_out[c * n + i] = pl[c][2 * i];
    }
    gold_complex_to_u8(pl[0], pl[1], pl[2], out, n);
    for (int c = 0; c < 3; c++) free(pl[c]);
}

/* ------------------------------------------------------------------ a14: normalise */

/* findMinMax, imageTools.cu:416-454 (never launched upstream): global min and max of a float
 * image.  floatToUchar4, :458-471: out = (v,v,v,255) by truncation.  The scaling step between
 * them was never written; the build defines normalize as v' = (v-min)/(max-min)*255 in fp32
 * (0 when max==min), then floatToUchar4's truncating store. */
void gold_minmax(const float *img, int64_t n, float *mn, float *mx)
{
    float a = img[0], b = img[0];
    for (int64_t i = 1; i < n; i++) { a = fminf(a, img[i]); b = fmaxf(b, img[i]); }
    *mn = a; *mx = b;
}

void gold_normalize_to_u8(const float *img, uint8_t *out, int64_t n, float mn, float mx)
{
    float range = mx - mn;
    for (int64_t i = 0; i < n; i++) {
        float v = range > 0.0f ? (img[i] - mn) / range * 255.0f : 0.0f;
        uint8_t q = sat_u8(v);
        put(out, i, q, q, q, 255);
    }
}
