// selftest.cpp -- mi_selftest(): `bool test()` of the reference's header (kernels/imageTools.h:19,
// declared, never defined upstream).  Known-answer checks through the library's own C ABI on the
// current device: small images whose correct output follows from the definition of each filter
// by inspection (constants, a step edge, one outlier, an impulse), so the expected values are
// written down here and nothing but the library under test computes pixels.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <hip/hip_runtime.h>

#include "internal.h"

namespace {

struct Report {
    int failures = 0;
    char first[256] = "";
    void fail(const char* fmt, ...) __attribute__((format(printf, 2, 3)))
    {
        char msg[256];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(msg, sizeof msg, fmt, ap);
        va_end(ap);
        fprintf(stderr, "mi_selftest: FAIL %s\n", msg);
        if (!failures) snprintf(first, sizeof first, "%s", msg);
        failures++;
    }
};

typedef std::vector<uint8_t> Img;

Img constant(int w, int h, uint8_t r, uint8_t g, uint8_t b, uint8_t a = 255)
{
    Img v((size_t)w * h * 4);
    for (size_t i = 0; i < v.size(); i += 4) { v[i] = r; v[i + 1] = g; v[i + 2] = b; v[i + 3] = a; }
    return v;
}

const uint8_t* px(const Img& v, int w, int x, int y) { return &v[((size_t)y * w + x) * 4]; }

bool interior(int x, int y, int w, int h) { return x >= 1 && x <= w - 2 && y >= 1 && y <= h - 2; }

// Every pixel of the write set must be (r,g,b,255) -- `deep` restricts the check to pixels at
// least `deep` away from the border (flat-index taps of wide kernels see zeros near it) --
// and every pixel outside the write set must be (0,0,0,0).
void expect_uniform(Report& rep, const char* what, const Img& out, int w, int h, int deep, int r, int g, int b, int tol = 0)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* p = px(out, w, x, y);
            if (!interior(x, y, w, h)) {
                if (p[0] | p[1] | p[2] | p[3]) { rep.fail("%s: border pixel (%d,%d) not zero", what, x, y); return; }
            } else if (x >= deep && x < w - deep && y >= deep && y < h - deep) {
                if (abs(p[0] - r) > tol || abs(p[1] - g) > tol || abs(p[2] - b) > tol || p[3] != 255) {
                    rep.fail("%s: pixel (%d,%d) = (%d,%d,%d,%d), expected (%d,%d,%d,255)", what, x, y, p[0], p[1], p[2], p[3], r, g, b);
                    return;
                }
            }
        }
}

#define RUN(call)                                                                  \
    do {                                                                           \
        const int rc_ = (call);                                                    \
        if (rc_ != MI_OK) { rep.fail("%s returned %d: %s", #call, rc_, mi_last_error()); return; } \
    } while (0)

void test_filters(Report& rep)
{
    const int w = 40, h = 24;
    Img out((size_t)w * h * 4);
    // grayscale: integer-exact cases of .299 R + .587 G + .114 B
    {
        Img in = constant(w, h, 0, 0, 0, 77);
        RUN(mi_grayscale_u8x4(in.data(), out.data(), w, h));
        for (size_t i = 0; i < out.size(); i += 4)
            if (out[i] | out[i + 1] | out[i + 2] || out[i + 3] != 77) { rep.fail("grayscale of black"); break; }
        in = constant(w, h, 200, 100, 50);      // 59.8 + 58.7 + 5.7 = 124.2 -> 124
        RUN(mi_grayscale_u8x4(in.data(), out.data(), w, h));
        for (size_t i = 0; i < out.size(); i += 4)
            if (out[i] != 124 || out[i + 1] != 124 || out[i + 2] != 124 || out[i + 3] != 255) {
                rep.fail("grayscale of (200,100,50) = %d, expected 124", out[i]);
                break;
            }
    }
    // constant image through every smoothing filter: the interior keeps its value
    {
        const Img in = constant(w, h, 90, 160, 30);
        RUN(mi_meanblur_u8x4(in.data(), out.data(), w, h, 3));
        expect_uniform(rep, "mean 3x3 of a constant", out, w, h, 1, 90, 160, 30);
        RUN(mi_meanblur_u8x4(in.data(), out.data(), w, h, 5));
        expect_uniform(rep, "mean 5x5 of a constant", out, w, h, 3, 90, 160, 30);
        RUN(mi_medianblur_u8x4(in.data(), out.data(), w, h, 3));
        expect_uniform(rep, "median 3x3 of a constant", out, w, h, 1, 90, 160, 30);
        RUN(mi_medianblur_u8x4(in.data(), out.data(), w, h, 5));
        expect_uniform(rep, "median 5x5 of a constant", out, w, h, 3, 90, 160, 30);
        RUN(mi_gaussian_u8x4(in.data(), out.data(), w, h, 5, 1.0f, MI_GAUSS_SEPARABLE));
        expect_uniform(rep, "Gaussian 5x5 of a constant", out, w, h, 3, 90, 160, 30, 1);
        RUN(mi_gaussian_u8x4(in.data(), out.data(), w, h, 15, 2.5f, MI_GAUSS_SEPARABLE));
        expect_uniform(rep, "Gaussian 15x15 of a constant", out, w, h, 8, 90, 160, 30, 1);
        RUN(mi_gaussian_u8x4(in.data(), out.data(), w, h, 3, 7.0f, MI_GAUSS_DIRECT));
        expect_uniform(rep, "direct Gaussian 3x3 of a constant", out, w, h, 1, 90, 160, 30, 1);
        RUN(mi_sobel_u8x4(in.data(), out.data(), w, h));
        expect_uniform(rep, "Sobel of a constant", out, w, h, 2, 0, 0, 0);
    }
    // one outlier disappears under the median and spreads as 255/9 = 28 under the 3x3 mean
    {
        Img in = constant(w, h, 10, 10, 10);
        uint8_t* c = &in[((size_t)12 * w + 20) * 4];
        c[0] = c[1] = c[2] = 255;
        RUN(mi_medianblur_u8x4(in.data(), out.data(), w, h, 3));
        expect_uniform(rep, "median 3x3 of one outlier", out, w, h, 1, 10, 10, 10);
        RUN(mi_medianblur_u8x4(in.data(), out.data(), w, h, 5));
        expect_uniform(rep, "median 5x5 of one outlier", out, w, h, 3, 10, 10, 10);
        RUN(mi_meanblur_u8x4(in.data(), out.data(), w, h, 3));
        for (int y = 10; y <= 14; y++)
            for (int x = 18; x <= 22; x++) {
                const int want = (abs(x - 20) <= 1 && abs(y - 12) <= 1) ? (8 * 10 + 255) / 9 : 10;
                if (px(out, w, x, y)[0] != want) { rep.fail("mean 3x3 near an outlier: (%d,%d) = %d, expected %d", x, y, px(out, w, x, y)[0], want); y = 99; break; }
            }
    }
    // vertical step edge in red: |Gx| = 4 * 200 = 800 on the two columns next to it -> 255
    {
        Img in = constant(w, h, 0, 50, 50);
        for (int y = 0; y < h; y++)
            for (int x = 20; x < w; x++) in[((size_t)y * w + x) * 4] = 200;
        RUN(mi_sobel_u8x4(in.data(), out.data(), w, h));
        for (int y = 1; y < h - 1; y++)
            for (int x = 1; x < w - 1; x++) {
                const int want = (x == 19 || x == 20) ? 255 : 0;
                const uint8_t* p = px(out, w, x, y);
                if (p[0] != want || p[1] != want || p[2] != want || p[3] != 255) {
                    rep.fail("Sobel of a step edge: (%d,%d) = %d, expected %d", x, y, p[0], want);
                    y = h; break;
                }
            }
        // a weaker edge: 4 * 30 = 120
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) in[((size_t)y * w + x) * 4] = x < 20 ? 100 : 130;
        RUN(mi_sobel_u8x4(in.data(), out.data(), w, h));
        if (px(out, w, 20, 12)[0] != 120 || px(out, w, 19, 12)[0] != 120 || px(out, w, 10, 12)[0] != 0)
            rep.fail("Sobel of a 30-level step: %d %d %d, expected 120 120 0", px(out, w, 19, 12)[0], px(out, w, 20, 12)[0], px(out, w, 10, 12)[0]);
    }
    // several filters in one trip == one after the other, in place, chunked
    {
        Img in((size_t)w * h * 4);
        for (size_t i = 0; i < in.size(); i++) in[i] = (uint8_t)((i * 2654435761u) >> 13);
        Img a = in, b((size_t)w * h * 4);
        const mi_op ops[3] = {{MI_FILTER_GAUSSIAN, 5, 1.0f}, {MI_FILTER_MEDIAN, 3, 0.f}, {MI_FILTER_SOBEL, 3, 0.f}};
        RUN(mi_run_ops_chunked_u8x4(a.data(), a.data(), w, h, ops, 3, 5));
        RUN(mi_gaussian_u8x4(in.data(), b.data(), w, h, 5, 1.0f, MI_GAUSS_SEPARABLE));
        RUN(mi_medianblur_u8x4(b.data(), b.data(), w, h, 3));
        RUN(mi_sobel_u8x4(b.data(), b.data(), w, h));
        if (memcmp(a.data(), b.data(), a.size()) != 0) rep.fail("chunked filter list differs from single calls");
        // image load is the identity on bytes
        RUN(mi_image_load(in.data(), b.data(), (size_t)w * h));
        if (memcmp(in.data(), b.data(), in.size()) != 0) rep.fail("image load changed bytes");
    }
    // Gaussian weights sum to one and are symmetric
    {
        float k[25];
        RUN(mi_gaussian_kernel(5, 1.0f, k));
        float sum = 0.f;
        for (float v : k) sum += v;
        if (fabsf(sum - 1.0f) > 1e-5f || k[0] != k[24] || k[1] != k[5] || !(k[12] > k[11]))
            rep.fail("Gaussian kernel 5x5: sum %.7f", sum);
    }
}

void test_transforms(Report& rep)
{
    struct Case { int rows, cols; };
    const Case cases[] = {{16, 32}, {64, 8}, {12, 10}, {15, 64}};
    for (const Case& c : cases) {
        const size_t n = (size_t)c.rows * c.cols;
        std::vector<float> x(2 * n), f(2 * n), back(2 * n);
        for (size_t i = 0; i < n; i++) { x[2 * i] = (float)((i * 37u + 11u) % 256u); x[2 * i + 1] = (float)((i * 91u) % 64u) - 32.f; }
        size_t ws_bytes = 0;
        if (mi_fft2d_workspace_bytes(c.rows, c.cols, &ws_bytes) != MI_OK) {
            rep.fail("fft2d %dx%d: workspace query failed: %s", c.rows, c.cols, mi_last_error());
            continue;
        }
        // device buffers are freed on every path out of this block
        struct DevBuf {
            void* p = nullptr;
            ~DevBuf() { if (p) (void)hipFree(p); }
        } bx, bf, bws;
        if (hipMalloc(&bx.p, n * 8) != hipSuccess || hipMalloc(&bf.p, n * 8) != hipSuccess ||
            (ws_bytes && hipMalloc(&bws.p, ws_bytes) != hipSuccess)) { rep.fail("hipMalloc in transform test"); return; }
        void *d_x = bx.p, *d_f = bf.p, *d_ws = bws.p;
        bool ok = hipMemcpy(d_x, x.data(), n * 8, hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && mi_fft2d_c2c_f32_ws_dev(d_x, d_f, c.rows, c.cols, 1, MI_FFT_FORWARD, d_ws, ws_bytes, nullptr) == MI_OK;
        ok = ok && hipMemcpy(f.data(), d_f, n * 8, hipMemcpyDeviceToHost) == hipSuccess;
        ok = ok && mi_fft2d_c2c_f32_ws_dev(d_f, d_f, c.rows, c.cols, 1, MI_FFT_INVERSE, d_ws, ws_bytes, nullptr) == MI_OK;
        ok = ok && hipMemcpy(back.data(), d_f, n * 8, hipMemcpyDeviceToHost) == hipSuccess;
        if (!ok) { rep.fail("fft2d %dx%d: call failed: %s", c.rows, c.cols, mi_last_error()); continue; }
        // X[0,0] = sum x;  Parseval;  inverse(forward(x)) = N x
        double sr = 0, si = 0, ex = 0, ef = 0, err = 0;
        for (size_t i = 0; i < n; i++) {
            sr += x[2 * i]; si += x[2 * i + 1];
            ex += (double)x[2 * i] * x[2 * i] + (double)x[2 * i + 1] * x[2 * i + 1];
            ef += (double)f[2 * i] * f[2 * i] + (double)f[2 * i + 1] * f[2 * i + 1];
            const double dr = back[2 * i] / (double)n - x[2 * i], di = back[2 * i + 1] / (double)n - x[2 * i + 1];
            err += dr * dr + di * di;
        }
        if (fabs(f[0] - sr) > 1e-5 * fabs(sr) + 1e-2 || fabs(f[1] - si) > 1e-5 * fabs(sr) + 1e-2)
            rep.fail("fft2d %dx%d: DC term (%g,%g), expected (%g,%g)", c.rows, c.cols, f[0], f[1], sr, si);
        if (fabs(ef - ex * (double)n) > 1e-5 * ex * (double)n) rep.fail("fft2d %dx%d: Parseval %g vs %g", c.rows, c.cols, ef, ex * (double)n);
        if (sqrt(err / ex) > 1e-5) rep.fail("fft2d %dx%d: round trip error %g", c.rows, c.cols, sqrt(err / ex));
    }
    // compress with every coefficient kept reproduces the image (within rounding of the cast)
    {
        const int w = 24, h = 20;
        Img in((size_t)w * h * 4), out((size_t)w * h * 4);
        for (size_t i = 0; i < in.size(); i++) in[i] = (i % 4 == 3) ? 255 : (uint8_t)(40 + (i * 7u) % 170u);
        RUN(mi_compress_u8x4(in.data(), out.data(), w, h, 1.0f));
        int worst = 0;
        for (size_t i = 0; i < in.size(); i++) worst = std::max(worst, abs((int)in[i] - (int)out[i]));
        if (worst > 1) rep.fail("compress(keep=1) changed a pixel by %d", worst);
        // a constant image has one non-zero coefficient: keeping a single one is lossless too
        Img flat = constant(w, h, 80, 120, 200);
        RUN(mi_compress_u8x4(flat.data(), out.data(), w, h, 1.0f / (float)(w * h)));
        worst = 0;
        for (size_t i = 0; i < flat.size(); i++) worst = std::max(worst, abs((int)flat[i] - (int)out[i]));
        if (worst > 1) rep.fail("compress(keep=1 coefficient) of a constant changed a pixel by %d", worst);
    }
}

}  // namespace

extern "C" int mi_selftest(void)
{
    Report rep;
    test_filters(rep);
    test_transforms(rep);
    if (rep.failures) return mi::fail(MI_ERR_SELFTEST, "%d self-test failure(s); first: %s", rep.failures, rep.first);
    return MI_OK;
}
