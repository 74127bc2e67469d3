// imgtool -- command-line counterpart of the reference's demo program (main.cpp:12-49), written
// against include/imageTools.h: the same wrapper calls on (img, img), with the file names and
// the list of operations taken from the command line instead of being edited into the source.
//
//   imgtool <in.jpg|in.png|in.ppm|...> <out.png> [--ops gray,gauss:5:1.0,median:3,sobel] [--one-trip]
//   imgtool --decode <in> <out.png>      decode only (stb_image, like the reference's loadImage), no GPU work
//   imgtool --selftest
//
// ops: gray | sobel | gauss:K:SIGMA | mean[:K] | median[:K] | fft_highpass | compress[:KEEP]
// (mean:K with K != 3 and compress:KEEP go through the C ABI, which has the parameters the
// reference's signatures lack).  --one-trip sends runs of consecutive pixel filters through
// mi_run_ops_u8x4 (one PCIe round trip per run instead of one per filter); the pixels are the same.
// The output is always a PNG, as with the reference (imageTools.cu:152).  Inputs are decoded as the
// reference decodes them -- stbi_load(..., STBI_rgb_alpha), main.cpp:14, with the same stb_image v2.30
// (third_party/stb/, unchanged) -- so a JPEG gives the same RGBA bytes as upstream; PNG and PNM inputs
// keep this file's own small decoders (stb would give the same pixels).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <vector>

#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_SIMD   // (the scalar and SIMD paths of stb_image give identical pixels; no SIMD keeps hipcc's host pass simple)
#include "../third_party/stb/stb_image.h"

#include "../include/imageTools.h"
#include "../include/mi_imagetools.h"

namespace {

[[noreturn]] void die(const std::string& msg)
{
    fprintf(stderr, "imgtool: %s\n", msg.c_str());
    exit(1);
}

std::vector<uint8_t> read_file(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) die(std::string("cannot open ") + path);
    std::vector<uint8_t> data;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    return data;
}

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// 8-bit, non-interlaced PNG of colour type 0 (gray), 2 (RGB), 4 (gray+alpha) or 6 (RGBA) -> RGBA.
std::vector<uint8_t> decode_png(const std::vector<uint8_t>& d, int* w, int* h)
{
    size_t pos = 8;
    std::vector<uint8_t> idat;
    int depth = 0, ctype = -1, interlace = 0;
    while (pos + 12 <= d.size()) {
        const uint32_t n = be32(&d[pos]);
        if (pos + 12 + n > d.size()) die("truncated PNG");
        const uint8_t* tag = &d[pos + 4];
        const uint8_t* body = &d[pos + 8];
        if (!memcmp(tag, "IHDR", 4)) {
            *w = (int)be32(body); *h = (int)be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + n);
        } else if (!memcmp(tag, "IEND", 4)) {
            break;
        }
        pos += 12 + n;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (depth != 8 || !ch || interlace || *w <= 0 || *h <= 0) die("unsupported PNG (need 8-bit, non-interlaced, gray/RGB/RGBA)");
    const size_t stride = (size_t)*w * ch;
    std::vector<uint8_t> raw((stride + 1) * *h);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) die("corrupt PNG data");
    std::vector<uint8_t> px((size_t)*w * *h * 4);
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    for (int y = 0; y < *h; y++) {
        const uint8_t* line = &raw[(stride + 1) * y];
        const int ft = line[0];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) / 2;
            else if (ft == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) die("bad PNG filter type");
            cur[i] = (uint8_t)(line[1 + i] + pred);
        }
        for (int x = 0; x < *w; x++) {
            uint8_t* o = &px[((size_t)y * *w + x) * 4];
            const uint8_t* s = &cur[(size_t)x * ch];
            if (ch == 1) { o[0] = o[1] = o[2] = s[0]; o[3] = 255; }
            else if (ch == 2) { o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; }
            else if (ch == 3) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255; }
            else memcpy(o, s, 4);
        }
        prev.swap(cur);
    }
    return px;
}

// binary PGM (P5) / PPM (P6), maxval 255 -> RGBA
std::vector<uint8_t> decode_pnm(const std::vector<uint8_t>& d, int* w, int* h)
{
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < d.size()) {
        if (d[pos] == '#') { while (pos < d.size() && d[pos] != '\n') pos++; }
        else if (d[pos] >= '0' && d[pos] <= '9') { int v = 0; while (pos < d.size() && d[pos] >= '0' && d[pos] <= '9') v = v * 10 + (d[pos++] - '0'); vals[got++] = v; continue; }
        pos++;
    }
    pos++;   // the single whitespace byte after maxval
    const int ch = d[1] == '6' ? 3 : 1;
    *w = vals[0]; *h = vals[1];
    if (got < 3 || vals[2] != 255 || *w <= 0 || *h <= 0 || pos + (size_t)*w * *h * ch > d.size()) die("unsupported or truncated PNM");
    std::vector<uint8_t> px((size_t)*w * *h * 4);
    for (size_t i = 0; i < (size_t)*w * *h; i++) {
        const uint8_t* s = &d[pos + i * ch];
        px[4 * i] = s[0]; px[4 * i + 1] = s[ch == 3 ? 1 : 0]; px[4 * i + 2] = s[ch == 3 ? 2 : 0]; px[4 * i + 3] = 255;
    }
    return px;
}

// File -> RGBA bytes.
std::vector<uint8_t> decode_file(const char* path, int* w, int* h)
{
    const std::vector<uint8_t> d = read_file(path);
    if (d.size() > 8 && !memcmp(d.data(), "\x89PNG\r\n\x1a\n", 8)) return decode_png(d, w, h);
    if (d.size() > 2 && d[0] == 'P' && (d[1] == '5' || d[1] == '6')) return decode_pnm(d, w, h);
    // everything else (JPEG first of all) the way the reference's loadImage does it (main.cpp:12-19)
    int channels = 0;
    unsigned char* img = stbi_load_from_memory(d.data(), (int)d.size(), w, h, &channels, STBI_rgb_alpha);
    if (!img) die(std::string("unknown image format: ") + path + " (" + stbi_failure_reason() + ")");
    std::vector<uint8_t> rgba(img, img + (size_t)*w * *h * 4);
    stbi_image_free(img);
    return rgba;
}

// Decoded file -> the library's pixel layout, through imageLoadWrapper as upstream (main.cpp:20-22).
uchar4* load_image(const char* path, int* w, int* h)
{
    std::vector<uint8_t> rgba = decode_file(path, w, h);
    const size_t n = (size_t)*w * *h;
    uchar4* img = new uchar4[n];
    imageLoadWrapper(rgba.data(), img, n);
    return img;
}

struct Op { std::string name; int k; float v; };

std::vector<Op> parse_ops(const std::string& spec)
{
    std::vector<Op> ops;
    size_t start = 0;
    while (start <= spec.size()) {
        size_t end = spec.find(',', start);
        if (end == std::string::npos) end = spec.size();
        std::string item = spec.substr(start, end - start);
        start = end + 1;
        if (item.empty()) continue;
        std::vector<std::string> parts;
        size_t a = 0;
        while (a <= item.size()) { size_t b = item.find(':', a); if (b == std::string::npos) b = item.size(); parts.push_back(item.substr(a, b - a)); a = b + 1; }
        Op op{parts[0], 3, 1.0f};
        if (op.name == "compress") op.v = parts.size() > 1 ? (float)atof(parts[1].c_str()) : MI_COMPRESS_DEFAULT_KEEP;
        else {
            if (parts.size() > 1) op.k = atoi(parts[1].c_str());
            if (parts.size() > 2) op.v = (float)atof(parts[2].c_str());
        }
        const char* known[] = {"gray", "sobel", "gauss", "mean", "median", "fft_highpass", "compress"};
        bool ok = false;
        for (const char* kname : known) ok |= op.name == kname;
        if (!ok) die("unknown op '" + op.name + "'");
        ops.push_back(op);
    }
    return ops;
}

int pixel_filter_id(const Op& op)
{
    if (op.name == "gray") return MI_FILTER_GRAYSCALE;
    if (op.name == "sobel") return MI_FILTER_SOBEL;
    if (op.name == "gauss") return MI_FILTER_GAUSSIAN_DIRECT;   // the same bytes as imageGaussianBlurWrapper (the reference's summation order)
    if (op.name == "mean") return MI_FILTER_MEAN;
    if (op.name == "median") return MI_FILTER_MEDIAN;
    return -1;
}

void check(int rc, const char* what)
{
    if (rc != MI_OK) die(std::string(what) + ": " + mi_last_error());
}

void apply(const Op& op, uchar4* img, int w, int h)
{
    if (op.name == "gray") imageGrayScaleWrapper(img, img, w, h);
    else if (op.name == "sobel") imageSobelEdgeWrapper(img, img, w, h);
    else if (op.name == "gauss") imageGaussianBlurWrapper(img, img, w, h, op.k, op.v);
    else if (op.name == "mean" && op.k == 3) imageMeanBlurWrapper(img, img, w, h);
    else if (op.name == "mean") check(mi_meanblur_u8x4((uint8_t*)img, (uint8_t*)img, w, h, op.k), "mean");
    else if (op.name == "median") imageMedianBlurWrapper(img, img, w, h, op.k);
    else if (op.name == "fft_highpass") imageFFTImageGenerate(img, img, w, h);
    else if (op.name == "compress" && op.v == MI_COMPRESS_DEFAULT_KEEP) compressImage(img, img, w, h);
    else check(mi_compress_u8x4((uint8_t*)img, (uint8_t*)img, w, h, op.v), "compress");
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc == 2 && !strcmp(argv[1], "--selftest")) {
        const bool ok = test();
        printf("self-test %s\n", ok ? "passed" : "FAILED");
        return ok ? 0 : 1;
    }
    if (argc == 4 && !strcmp(argv[1], "--decode")) {
        int w = 0, h = 0;
        const std::vector<uint8_t> rgba = decode_file(argv[2], &w, &h);
        check(mi_image_write(argv[3], rgba.data(), w, h), "write");
        printf("decoded %s: %d x %d\n", argv[2], w, h);
        return 0;
    }
    const char *in = nullptr, *out = nullptr;
    std::string spec = "fft_highpass";   // what the reference's main.cpp runs as checked in
    bool one_trip = false;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--ops") && i + 1 < argc) spec = argv[++i];
        else if (!strcmp(argv[i], "--one-trip")) one_trip = true;
        else if (!in) in = argv[i];
        else if (!out) out = argv[i];
        else die(std::string("unexpected argument ") + argv[i]);
    }
    if (!in || !out) die("usage: imgtool <in> <out.png> [--ops LIST] [--one-trip] | imgtool --decode <in> <out.png> | imgtool --selftest");
    int w = 0, h = 0;
    uchar4* img = load_image(in, &w, &h);
    printf("loaded %s: %d x %d\n", in, w, h);
    const std::vector<Op> ops = parse_ops(spec);
    for (size_t i = 0; i < ops.size();) {
        size_t j = i;
        std::vector<mi_op> run;
        while (one_trip && j < ops.size() && pixel_filter_id(ops[j]) >= 0 && run.size() < MI_MAX_OPS) {
            run.push_back(mi_op{pixel_filter_id(ops[j]), ops[j].k, ops[j].v});
            j++;
        }
        if (run.size() > 1) {
            check(mi_run_ops_u8x4((uint8_t*)img, (uint8_t*)img, w, h, run.data(), (int)run.size()), "filter list");
            i = j;
        } else {
            apply(ops[i], img, w, h);
            i++;
        }
    }
    imageWriteWrapper(out, img, w, h);
    printf("wrote %s\n", out);
    delete[] img;
    return 0;
}
