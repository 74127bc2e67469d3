// strips.hip -- one large image in row strips that STAY on the devices, behind the C ABI (BASELINE.json
// config 4 for a C++ caller of imageTools.h; the reference is single-GPU, so there is no counterpart
// upstream).  Strip i of a w x h image -- rows [h*i/G, h*(i+1)/G) -- lives on devices[i] in the middle of a
// buffer of (pad + rows + pad) image rows.  One stage = one filter over every strip:
//   1. on each device's copy stream, the neighbours' edge rows travel straight into this strip's pad
//      rows (hipMemcpyPeerAsync: point to point over xGMI, no collective, no host copy);
//   2. meanwhile the device's main stream filters the interior rows, which need no halo;
//   3. the main stream waits for its copies (an event, no host synchronisation) and filters the two
//      edge bands.
// The kernels read strip and halo as one contiguous run of rows, so nothing is concatenated or
// allocated per stage.  The same decomposition as cudaimageprocessing_amd/shard.py's StripPlan (one
// process per GPU over RCCL); this is the one-process form.
#include <mutex>

#include "internal.h"

namespace mi {

namespace {

constexpr int MAX_DEV = 64;

struct StripStreams {
    hipStream_t main = nullptr, copy = nullptr;
    hipEvent_t halos = nullptr;
};

// per device ordinal; created with that device current, kept until the process ends
std::mutex g_mu;
StripStreams g_streams[MAX_DEV];

hipError_t strip_streams(int dev, StripStreams** out)
{
    std::lock_guard<std::mutex> lock(g_mu);
    StripStreams& s = g_streams[dev];
    hipError_t e = hipSuccess;
    if (!s.main) e = hipStreamCreateWithFlags(&s.main, hipStreamNonBlocking);
    if (e == hipSuccess && !s.copy) e = hipStreamCreateWithFlags(&s.copy, hipStreamNonBlocking);
    if (e == hipSuccess && !s.halos) e = hipEventCreateWithFlags(&s.halos, hipEventDisableTiming);
    *out = &s;
    return e;
}

}  // namespace

int strip_rows(int h, int n, int i, int* y0, int* y1)
{
    *y0 = (int)((int64_t)h * i / n);
    *y1 = (int)((int64_t)h * (i + 1) / n);
    return *y1 - *y0;
}

// d_in[i] / d_out[i]: device buffers of (pad + rows_i + pad) rows of w pixels on devices[i]; the strip's rows
// start at row `pad` of each.  Blocking: returns when every device has finished the stage.
int strip_stage_dev(void* const* d_in, void* const* d_out, int w, int h, const int* devices, int n, int pad,
                    int filter, int ksize, float sigma)
{
    const int halo = filter_halo_rows(filter, ksize, w);
    if (halo < 0) return fail(MI_ERR_INVALID_ARG, "unknown filter %d", filter);
    if (halo > pad && n > 1) return fail(MI_ERR_INVALID_ARG, "the stage needs %d halo rows, the buffers have %d pad rows", halo, pad);
    const size_t pitch = (size_t)w * 4;
    int keep = 0;
    const bool have = hipGetDevice(&keep) == hipSuccess;
    int rc = MI_OK;
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && n > 1; i++) {
        int y0, y1;
        if (strip_rows(h, n, i, &y0, &y1) < halo) return fail(MI_ERR_INVALID_ARG, "strip %d has %d rows, fewer than the halo (%d)", i, y1 - y0, halo);
    }
    // Two strips on ONE device (an ordinal listed twice) share that device's streams: stream order then
    // runs them one after the other, still correct.
    for (int i = 0; i < n && rc == MI_OK && e == hipSuccess; i++) {
        int y0, y1;
        const int rows = strip_rows(h, n, i, &y0, &y1);
        if (rows <= 0) continue;
        e = hipSetDevice(devices[i]);
        if (e != hipSuccess) break;
        StripStreams* ss = nullptr;
        e = strip_streams(devices[i], &ss);
        if (e != hipSuccess) break;
        char* src = (char*)d_in[i];
        char* dst = (char*)d_out[i] + (size_t)pad * pitch;            // the strip's rows of the output buffer
        const bool up = i > 0 && halo > 0, down = i < n - 1 && halo > 0;
        if (up) {      // the last `halo` rows of strip i-1 -> my pad rows just above the strip
            int py0, py1;
            const int prow = strip_rows(h, n, i - 1, &py0, &py1);
            e = hipMemcpyPeerAsync(src + (size_t)(pad - halo) * pitch, devices[i],
                                   (const char*)d_in[i - 1] + (size_t)(pad + prow - halo) * pitch, devices[i - 1],
                                   (size_t)halo * pitch, ss->copy);
        }
        if (down && e == hipSuccess)      // the first `halo` rows of strip i+1 -> my pad rows just below the strip
            e = hipMemcpyPeerAsync(src + (size_t)(pad + rows) * pitch, devices[i],
                                   (const char*)d_in[i + 1] + (size_t)pad * pitch, devices[i + 1], (size_t)halo * pitch, ss->copy);
        if (e == hipSuccess) e = hipEventRecord(ss->halos, ss->copy);
        if (e != hipSuccess) break;
        const int a = up ? halo : 0, b = down ? halo : 0;
        auto run = [&](int in_row0, int in_rows, int yb, int ye) {
            FilterArgs f;
            f.filter = filter;
            if (filter == MI_FILTER_GAUSSIAN && (ksize == 1 || ksize > 31)) f.filter = MI_FILTER_GAUSSIAN_DIRECT;
            f.in = src + (size_t)(pad + (in_row0 - y0)) * pitch;
            f.out = dst + (size_t)(yb - y0) * pitch;
            f.w = w; f.h_total = h; f.batch = 1;
            f.in_row0 = in_row0; f.in_rows = in_rows; f.y_begin = yb; f.y_end = ye;
            f.ksize = ksize; f.sigma = sigma; f.stream = ss->main;
            return launch_filter(f);
        };
        if (rows >= 2 * halo && (a || b)) {
            if (rows - a - b > 0) rc = run(y0, rows, y0 + a, y1 - b);        // interior: local rows only
            if (rc == MI_OK) e = hipStreamWaitEvent(ss->main, ss->halos, 0);
            if (rc == MI_OK && e == hipSuccess && a) rc = run(y0 - halo, 3 * halo, y0, y0 + a);
            if (rc == MI_OK && e == hipSuccess && b) rc = run(y1 - 2 * halo, 3 * halo, y1 - b, y1);
        } else {          // no neighbours, or a strip thinner than two halos: one piece, after the halos
            e = hipStreamWaitEvent(ss->main, ss->halos, 0);
            if (e == hipSuccess) rc = run(y0 - a, rows + a + b, y0, y1);
        }
    }
    // wait for everything that was enqueued, whatever failed
    for (int i = 0; i < n; i++) {
        if (hipSetDevice(devices[i]) != hipSuccess) continue;
        StripStreams& s = g_streams[devices[i]];
        if (s.copy) { const hipError_t e2 = hipStreamSynchronize(s.copy); if (e == hipSuccess) e = e2; }
        if (s.main) { const hipError_t e2 = hipStreamSynchronize(s.main); if (e == hipSuccess) e = e2; }
    }
    if (have) (void)hipSetDevice(keep);
    if (rc) return rc;
    return hip_status(e, "device-resident strip stage");
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_strip_rows(int h, int n_devices, int i, int* y0, int* y1)
{
    if (!y0 || !y1 || h <= 0 || n_devices <= 0 || i < 0 || i >= n_devices) return fail(MI_ERR_INVALID_ARG, "bad argument");
    strip_rows(h, n_devices, i, y0, y1);
    return MI_OK;
}

int mi_strip_stage_u8x4_dev(void* const* d_in, void* const* d_out, int w, int h, const int* devices, int n_devices,
                            int pad_rows, int filter, int param_i, float param_f)
{
    if (!d_in || !d_out || !devices || n_devices <= 0 || n_devices > MAX_DEV) return fail(MI_ERR_INVALID_ARG, "need 1..64 strips");
    if (w <= 0 || h <= 0 || pad_rows < 0) return fail(MI_ERR_INVALID_ARG, "bad size %dx%d, pad %d", w, h, pad_rows);
    const int count = mi_device_count();
    for (int i = 0; i < n_devices; i++) {
        if (devices[i] < 0 || devices[i] >= count || devices[i] >= MAX_DEV) return fail(MI_ERR_INVALID_ARG, "device %d of %d visible", devices[i], count);
        if (!d_in[i] || !d_out[i]) return fail(MI_ERR_INVALID_ARG, "null strip buffer %d", i);
        if (d_in[i] == d_out[i]) return fail(MI_ERR_OVERLAP, "a stage's input and output buffers must differ (strip %d)", i);
    }
    const int rc = check_filter(filter, w, h, param_i, param_f);
    if (rc) return rc;
    return strip_stage_dev(d_in, d_out, w, h, devices, n_devices, pad_rows, filter, param_i, param_f);
}

}  // extern "C"
