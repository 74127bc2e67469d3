// fft.hip -- 2-D complex fp32 DFT (replaces the reference's cuFFT calls,
// kernels/imageTools.cu:687-701, 741-751).  Placeholder until the LDS-resident kernels land.
#include "common.h"
#include "internal.h"

namespace mi {

int launch_fft2d(const void*, void*, int, int, int, int, hipStream_t)
{
    return fail(MI_ERR_UNSUPPORTED, "fft2d is not built yet");
}

}  // namespace mi
