// shim.cpp -- the reference's C++ wrapper names (kernels/imageTools.h:10-18) as thin calls
// into the C ABI.  void returns; a failure prints and aborts like the reference's checkCuda
// (kernels/imageTools.cu:97-105).
#include <stdio.h>
#include <stdlib.h>

#include "../../include/imageTools.h"
#include "../../include/mi_imagetools.h"

static void check(int rc, const char* what)
{
    if (rc != MI_OK) {
        fprintf(stderr, "%s failed (%d): %s\n", what, rc, mi_last_error());
        abort();
    }
}

void imageLoadWrapper(unsigned char* image, uchar4* imageLoaded, size_t imgSize)
{
    check(mi_image_load(image, (uint8_t*)imageLoaded, imgSize), "imageLoadWrapper");
}

void imageWriteWrapper(const char* filename, uchar4* pixels, int width, int height)
{
    check(mi_image_write(filename, (const uint8_t*)pixels, width, height), "imageWriteWrapper");
}

void imageGrayScaleWrapper(uchar4* returnImage, uchar4* imageLoaded, int width, int height)
{
    check(mi_grayscale_u8x4((const uint8_t*)imageLoaded, (uint8_t*)returnImage, width, height), "imageGrayScaleWrapper");
}

void imageSobelEdgeWrapper(uchar4* returnImage, uchar4* imageLoaded, int width, int height)
{
    check(mi_sobel_u8x4((const uint8_t*)imageLoaded, (uint8_t*)returnImage, width, height), "imageSobelEdgeWrapper");
}

void imageGaussianBlurWrapper(uchar4* returnImage, uchar4* imageLoaded, int width, int height, int kernalSize, float sigma)
{   // the reference's own k x k summation order: the same bytes as upstream (the separable form, within
    // 1 LSB of it, is mi_gaussian_u8x4(..., MI_GAUSS_SEPARABLE); over PCIe the two take the same time)
    check(mi_gaussian_u8x4((const uint8_t*)imageLoaded, (uint8_t*)returnImage, width, height, kernalSize, sigma, MI_GAUSS_DIRECT), "imageGaussianBlurWrapper");
}

void imageMeanBlurWrapper(uchar4* returnImage, uchar4* imageLoaded, int width, int height)
{   // the reference hard-codes kernel size 3 (kernels/imageTools.cu:806)
    check(mi_meanblur_u8x4((const uint8_t*)imageLoaded, (uint8_t*)returnImage, width, height, 3), "imageMeanBlurWrapper");
}

void imageMedianBlurWrapper(uchar4* returnImage, uchar4* imageLoaded, int width, int height, int kernalSize)
{
    check(mi_medianblur_u8x4((const uint8_t*)imageLoaded, (uint8_t*)returnImage, width, height, kernalSize), "imageMedianBlurWrapper");
}

void imageFFTImageGenerate(uchar4* returnImage, uchar4* imageLoaded, int width, int height)
{   // cutoff 10.0f is the reference's constant (kernels/imageTools.cu:716)
    check(mi_fft_highpass_u8x4((const uint8_t*)imageLoaded, (uint8_t*)returnImage, width, height, 10.0f), "imageFFTImageGenerate");
}

void compressImage(uchar4* outputImage, uchar4* inputImage, int width, int height)
{   // never defined upstream; see mi_compress_u8x4 for the definition used here
    check(mi_compress_u8x4((const uint8_t*)inputImage, (uint8_t*)outputImage, width, height, MI_COMPRESS_DEFAULT_KEEP), "compressImage");
}

bool test()
{   // never defined upstream; the library's self-test
    return mi_selftest() == MI_OK;
}

float* generateGaussianKernal(int kernalSize, float sigma)
{
    float* k = (float*)malloc((size_t)kernalSize * kernalSize * sizeof(float));
    if (!k) abort();
    check(mi_gaussian_kernel(kernalSize, sigma, k), "generateGaussianKernal");
    return k;
}
