// noalloc.hip -- the one entry point of the library that allocates nothing, for callers that
// cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid
// it).  No counterpart in the reference.
//
// mi_fft_highpass_u8x4_static_dev is mi_fft_highpass_u8x4_dev (imageFFTImageGenerate on a device
// image) with the workspace argument replaced by a file-scope __device__ array: the same kernels
// run on the same layout, so the output is the same bytes.  Every GPU the library runs on holds
// its own copy of the array until the process ends.
#include <mutex>

#include "internal.h"

namespace mi {

// three planes + three spectra + transform workspace of one image; 64 MiB covers the reference's
// own 675x900 demo image (48.8 MB) and power-of-two images up to 1024x1024 (56 MB)
static __device__ __align__(256) unsigned char scratch[MI_STATIC_SCRATCH_BYTES];

// In host code `scratch` is only a stand-in for the symbol: the address of a device's copy of the
// array comes from hipGetSymbolAddress with that device current, once per device.
static int static_scratch(void** ptr)
{
    constexpr int MAX_DEV = 64;
    static std::mutex mu;
    static void* dev_ptr[MAX_DEV] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_status(e, "hipGetDevice");
    if (dev < 0 || dev >= MAX_DEV) return fail(MI_ERR_UNSUPPORTED, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (!dev_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return hip_status(e, "hipGetSymbolAddress(scratch)");
        if (!p) return fail(MI_ERR_UNSUPPORTED, "hipGetSymbolAddress returned a null address");
        dev_ptr[dev] = p;
    }
    *ptr = dev_ptr[dev];
    return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" int mi_fft_highpass_u8x4_static_dev(const void* d_in, void* d_out, int w, int h, float cutoff, void* stream)
{
    if (!d_in || !d_out) return fail(MI_ERR_INVALID_ARG, "null pointer");
    size_t need = 0;
    int rc = mi_fft_highpass_workspace_bytes(w, h, &need);
    if (rc) return rc;
    if (need > (size_t)MI_STATIC_SCRATCH_BYTES)
        return fail(MI_ERR_UNSUPPORTED, "fft_highpass %dx%d needs %zu bytes of scratch, the static buffer holds %zu: "
                    "use mi_fft_highpass_u8x4_dev with a workspace", w, h, need, (size_t)MI_STATIC_SCRATCH_BYTES);
    void* ws = nullptr;
    rc = static_scratch(&ws);
    if (rc) return rc;
    // nothing in the buffer has to start at zero: every kernel of the path writes what it later reads
    return mi_fft_highpass_u8x4_dev(d_in, d_out, w, h, cutoff, ws, MI_STATIC_SCRATCH_BYTES, stream);
}
