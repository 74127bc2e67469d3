// stb_write.cpp -- the one translation unit that compiles stb_image_write v1.16 (third_party/stb, the header
// the reference vendors, unchanged).  imageWriteWrapper ends in stbi_write_png(filename, w, h, 4, buf, w*4)
// (kernels/imageTools.cu:152); mi_image_write calls the same encoder through png_write_rgba_stb, so the FILE
// is byte-identical to what the reference writes for the same pixels, not only its decoded content.
#define STB_IMAGE_WRITE_IMPLEMENTATION
#define STBIW_WINDOWS_UTF8 0
#if defined(__clang__)
#pragma clang diagnostic ignored "-Wunused-function"
#pragma clang diagnostic ignored "-Wmissing-field-initializers"
#endif
#include "../../third_party/stb/stb_image_write.h"

namespace mi {
int png_write_rgba_stb(const char* filename, const unsigned char* px, int w, int h)
{
    return stbi_write_png(filename, w, h, 4, px, w * 4);      // 0 on failure
}
}  // namespace mi
