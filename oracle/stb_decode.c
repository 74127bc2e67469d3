/* stb_decode.c -- TEST INFRASTRUCTURE (like golden.c): the reference's image decode step for the CPU
 * baseline of BASELINE.json config 1 and for tests.  The reference's loadImage (main.cpp:12-26) calls
 * stbi_load(filename, &w, &h, &channels, STBI_rgb_alpha) from its vendored stb_image v2.30; this shim
 * does exactly that with the same header (third_party/stb/stb_image.h, byte-identical to the
 * reference's include/stb_image.h).  Nothing in the product path links or loads it. */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_STDIO_WARNINGS
#include "../third_party/stb/stb_image.h"

/* RGBA bytes (w*h*4, free with gold_stb_free) or NULL */
unsigned char* gold_stb_load_rgba(const char* path, int* w, int* h)
{
    int channels = 0;
    return stbi_load(path, w, h, &channels, STBI_rgb_alpha);
}

void gold_stb_free(unsigned char* p) { stbi_image_free(p); }
