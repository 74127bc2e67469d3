// imageTools.h -- drop-in C++ header with the signatures of ctaylo41/CudaImageProcessing's
// kernels/imageTools.h:10-19, implemented by libmi_imagetools.so (HIP, gfx950) over the C ABI
// in mi_imagetools.h.  A program written against the reference's header compiles against this
// one unchanged, provided it includes <hip/hip_runtime.h> (or just hip_vector_types.h, as done
// here) where it used to include <cuda_runtime.h> for the uchar4 type (reference main.cpp:2-3).
//
// Deliberate deviation: the reference header also leaks six __global__ kernel prototypes
// (kernels/imageTools.h:4-9); kernel symbols are not part of a shared-library boundary and are
// not declared here.
//
// Error behaviour follows the reference's checkCuda (kernels/imageTools.cu:97-105): a failing
// call prints a message to stderr and aborts.
#ifndef IMAGETOOLS_H
#define IMAGETOOLS_H
#include <stddef.h>
#include <hip/hip_vector_types.h>

void imageLoadWrapper(unsigned char* image, uchar4* imageLoaded, size_t imgSize);
void imageWriteWrapper(const char* filename, uchar4* pixels, int width, int height);
void imageGrayScaleWrapper(uchar4* returnImage, uchar4* imageLoaded, int width, int height);
void imageSobelEdgeWrapper(uchar4* returnImage, uchar4* imageLoaded, int width, int height);
// The reference's k x k sum in its own order (bit-exact); kernalSize odd, 1..31 (upstream: any odd size).
void imageGaussianBlurWrapper(uchar4* returnImage, uchar4* imageLoaded, int width, int height, int kernalSize, float sigma);
void imageMeanBlurWrapper(uchar4* returnImage, uchar4* imageLoaded, int width, int height);
void imageFFTImageGenerate(uchar4* returnImage, uchar4* imageLoaded, int width, int height);
// Declared but never defined upstream (kernels/imageTools.h:17, call commented out at main.cpp:44).
// Defined here as: keep the 10 % largest DFT coefficients of each channel (mi_compress_u8x4).
void compressImage(uchar4* outputImage, uchar4* inputImage, int width, int height);
// Declared but never defined upstream (kernels/imageTools.h:19): the library's self-test
// (mi_selftest) -- true when every known-answer check passes on the current device.
bool test();
// Returns malloc()ed memory; the caller frees it (kernels/imageTools.cu:364, 501).
float* generateGaussianKernal(int kernalSize, float sigma);
// New: the median blur the reference's README advertises (kernalSize 3 or 5).
void imageMedianBlurWrapper(uchar4* returnImage, uchar4* imageLoaded, int width, int height, int kernalSize);
#endif  // IMAGETOOLS_H
