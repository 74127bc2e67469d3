// host.hip -- the blocking host tier for the u8x4 filters: what the reference's wrappers do
// (kernels/imageTools.cu:233-253 and siblings: device buffers, H2D, kernel, D2H, free), as a
// pipeline.  The image is cut into row chunks; a chunk is staged into pinned memory, copied to
// the device, run through the whole list of filters on the device (each stage's row range shrinks
// by that filter's halo) and copied back, on one of two streams, while the host stages the next
// chunk and a helper thread unloads the finished ones.  Pinned and device buffers are kept
// between calls.
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"

namespace mi {

namespace {

// One host thread moves about 20 GB/s between pageable and pinned memory; the PCIe link takes more than
// twice that in each direction.  A CopyCrew is a few helper threads that each copy a slice of a staging
// memcpy (the caller takes the first slice and returns when every slice is done).  A pipe keeps its two
// crews between calls.  An idle helper polls for about as long as a chunk takes to come round again
// (the chunks of one call follow each other within a fraction of a millisecond) and then parks on a
// condition variable: between calls, and while a device thread waits for the GPU, nothing spins.
class CopyCrew {
public:
    explicit CopyCrew(int helpers)
    {
        for (int i = 0; i < helpers; i++) th_.emplace_back([this, i] { work(i + 1); });
    }
    ~CopyCrew()
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            quit_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
        for (std::thread& t : th_) t.join();
    }
    CopyCrew(const CopyCrew&) = delete;
    CopyCrew& operator=(const CopyCrew&) = delete;
    int helpers() const { return (int)th_.size(); }
    void copy(void* dst, const void* src, size_t n)
    {
        const size_t parts = th_.size() + 1;
        if (parts == 1 || n < ((size_t)1 << 20)) { memcpy(dst, src, n); return; }
        dst_ = (char*)dst; src_ = (const char*)src; n_ = n;
        slice_ = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        pending_.store((int)th_.size(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lock(mu_);      // a helper between its last poll and its wait sees the new job
            gen_.fetch_add(1, std::memory_order_release);
        }
        if (parked_.load(std::memory_order_acquire) != 0) cv_.notify_all();
        memcpy(dst_, src_, std::min(slice_, n_));
        while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }

private:
    void work(int part)
    {
        uint64_t seen = 0;
        for (;;) {
            uint64_t g = seen;
            for (int spin = 0; spin < 2000 && (g = gen_.load(std::memory_order_acquire)) == seen; spin++) {
                if (quit_.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            if (g == seen) {
                std::unique_lock<std::mutex> lock(mu_);
                parked_.fetch_add(1, std::memory_order_acq_rel);
                cv_.wait(lock, [&] { return gen_.load(std::memory_order_acquire) != seen || quit_.load(std::memory_order_acquire); });
                parked_.fetch_sub(1, std::memory_order_acq_rel);
                if (quit_.load(std::memory_order_acquire)) return;
                g = gen_.load(std::memory_order_acquire);
            }
            seen = g;
            const size_t lo = std::min(n_, slice_ * (size_t)part), hi = std::min(n_, lo + slice_);
            if (hi > lo) memcpy(dst_ + lo, src_ + lo, hi - lo);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> pending_{0}, parked_{0};
    std::atomic<bool> quit_{false};
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t n_ = 0, slice_ = 0;
};

// Helper threads per crew: MI_HOST_COPY_THREADS - 1, shared out between the device threads when one call
// drives several devices (run_on_devices sets the share), so that eight GPUs do not bring 56 helpers.
thread_local int t_device_share = 1;
int crew_helpers()
{
    const int threads = std::max(1, std::min(tuning().host_copy_threads, 8));
    return std::max(0, threads / std::max(1, t_device_share) - 1);
}

struct Slot {
    void* pin_in = nullptr;
    void* pin_out = nullptr;
    void* dev[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
};

// One pipe (and one spectral scratch, below) per device: host-tier calls from threads that drive
// different GPUs neither serialise on each other nor free each other's buffers.
constexpr int MAX_DEV = 64;

struct HostPipe {
    std::mutex mu;
    size_t cap_in = 0, cap_out = 0, cap_dev = 0;
    static constexpr int NSLOT = 3;
    Slot slot[NSLOT];
    std::unique_ptr<CopyCrew> crew_in, crew_out;       // kept between calls

    // the pipe's two crews with `helpers` helper threads each (called with mu held)
    void crews(int helpers)
    {
        if (!crew_in || crew_in->helpers() != helpers) crew_in.reset(new CopyCrew(helpers));
        if (!crew_out || crew_out->helpers() != helpers) crew_out.reset(new CopyCrew(helpers));
    }

    void release()
    {
        crew_in.reset();
        crew_out.reset();
        for (Slot& s : slot) {
            if (s.pin_in) (void)hipHostFree(s.pin_in);
            if (s.pin_out) (void)hipHostFree(s.pin_out);
            for (void*& d : s.dev) { if (d) (void)hipFree(d); d = nullptr; }
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.stream) (void)hipStreamDestroy(s.stream);
            s = Slot();
        }
        cap_in = cap_out = cap_dev = 0;
    }

    // called with this pipe's device current
    hipError_t reserve(size_t in_bytes, size_t out_bytes, size_t dev_bytes)
    {
        hipError_t e = hipSuccess;
        for (Slot& s : slot) {
            if (!s.stream && (e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking)) != hipSuccess) return e;
            if (!s.done && (e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess) return e;
        }
        if (in_bytes > cap_in) {
            for (Slot& s : slot) {
                if (s.pin_in) (void)hipHostFree(s.pin_in);
                s.pin_in = nullptr;
                if ((e = hipHostMalloc(&s.pin_in, in_bytes, hipHostMallocDefault)) != hipSuccess) { cap_in = 0; return e; }
            }
            cap_in = in_bytes;
        }
        if (out_bytes > cap_out) {
            for (Slot& s : slot) {
                if (s.pin_out) (void)hipHostFree(s.pin_out);
                s.pin_out = nullptr;
                if ((e = hipHostMalloc(&s.pin_out, out_bytes, hipHostMallocDefault)) != hipSuccess) { cap_out = 0; return e; }
            }
            cap_out = out_bytes;
        }
        if (dev_bytes > cap_dev) {
            for (Slot& s : slot)
                for (void*& d : s.dev) {
                    if (d) (void)hipFree(d);
                    d = nullptr;
                    if ((e = hipMalloc(&d, dev_bytes)) != hipSuccess) { cap_dev = 0; return e; }
                }
            cap_dev = dev_bytes;
        }
        return hipSuccess;
    }
};

HostPipe g_pipes[MAX_DEV];

// device scratch of the host-tier spectral calls (grow-only, one per device)
struct Scratch {
    std::mutex mu;
    void* ptr = nullptr;
    size_t cap = 0;
    HostScratch::TableKey tables{nullptr, 0, 0};
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        tables = HostScratch::TableKey{nullptr, 0, 0};
    }
};

Scratch g_scratches[MAX_DEV];

struct Rows { int lo, hi; };

int current_device(int* dev)
{
    const hipError_t e = hipGetDevice(dev);
    if (e != hipSuccess) return hip_status(e, "hipGetDevice");
    if (*dev < 0 || *dev >= MAX_DEV) return fail(MI_ERR_UNSUPPORTED, "device ordinal %d out of range", *dev);
    return MI_OK;
}

}  // namespace

// Frees every device's cached buffers (the caller's current device is restored).
int host_release()
{
    int keep = -1;
    const bool have = hipGetDevice(&keep) == hipSuccess;
    for (int d = 0; d < MAX_DEV; d++) {
        // lock order everywhere: a device's scratch, then its pipe (the spectral round trips hold the
        // scratch for the whole call and take the pipe for their staged copies)
        std::lock_guard<std::mutex> lock(g_scratches[d].mu);
        std::lock_guard<std::mutex> lock2(g_pipes[d].mu);
        const bool used = g_pipes[d].slot[0].stream || g_pipes[d].cap_dev || g_scratches[d].ptr;
        if (!used) continue;
        (void)hipSetDevice(d);
        g_pipes[d].release();
        g_scratches[d].release();
    }
    if (have) (void)hipSetDevice(keep);
    return MI_OK;
}

int host_scratch(size_t bytes, HostScratch* out)
{
    out->ptr = nullptr;
    int dev = 0;
    const int rc = current_device(&dev);
    if (rc) return rc;
    Scratch& sc = g_scratches[dev];
    out->lock = std::unique_lock<std::mutex>(sc.mu);
    if (bytes > sc.cap) {
        sc.release();
        const hipError_t e = hipMalloc(&sc.ptr, bytes);
        if (e != hipSuccess) return hip_status(e, "host-tier scratch");
        sc.cap = bytes;
    }
    out->ptr = sc.ptr;
    out->tables = &sc.tables;
    return MI_OK;
}

// Whole-buffer copies between pageable host memory and the device for the host-tier calls that need
// the entire image on the device before they can start (the spectral round trips, the load repack):
// staged through the pipe's pinned buffers in 4-MiB chunks -- CopyCrew fills or drains one chunk while
// the DMA engine moves its neighbours -- instead of one blocking hipMemcpy from pageable memory
// (measured on a 4096 x 4096 image: 22 GB/s per direction that way).  Small copies stay a hipMemcpy.
// The device's default stream must be idle for `dev` (callers use blocking calls around these).
namespace {
constexpr size_t STAGE_CHUNK = (size_t)4 << 20;
constexpr size_t STAGE_MIN = (size_t)12 << 20;
}  // namespace

int host_upload(void* dev, const void* host, size_t bytes)
{
    if (bytes < STAGE_MIN) return hip_status(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice), "host-to-device copy");
    int device = 0;
    int rc = current_device(&device);
    if (rc) return rc;
    HostPipe& pipe = g_pipes[device];
    std::lock_guard<std::mutex> lock(pipe.mu);
    hipError_t e = pipe.reserve(STAGE_CHUNK, 0, 0);
    if (e != hipSuccess) return hip_status(e, "host-tier staging buffers");
    pipe.crews(crew_helpers());
    CopyCrew& crew = *pipe.crew_in;
    int c = 0;
    for (size_t off = 0; off < bytes && e == hipSuccess; off += STAGE_CHUNK, c++) {
        Slot& s = pipe.slot[c % HostPipe::NSLOT];
        const size_t n = std::min(STAGE_CHUNK, bytes - off);
        if (c >= HostPipe::NSLOT) e = hipEventSynchronize(s.done);      // the DMA that last read this pinned buffer
        if (e != hipSuccess) break;
        crew.copy(s.pin_in, (const char*)host + off, n);
        e = hipMemcpyAsync((char*)dev + off, s.pin_in, n, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) e = hipEventRecord(s.done, s.stream);
    }
    for (Slot& s : pipe.slot) {
        const hipError_t e2 = hipStreamSynchronize(s.stream);
        if (e == hipSuccess) e = e2;
    }
    return hip_status(e, "staged host-to-device copy");
}

int host_download(void* host, const void* dev, size_t bytes)
{
    if (bytes < STAGE_MIN) return hip_status(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost), "device-to-host copy");
    int device = 0;
    int rc = current_device(&device);
    if (rc) return rc;
    HostPipe& pipe = g_pipes[device];
    std::lock_guard<std::mutex> lock(pipe.mu);
    hipError_t e = pipe.reserve(0, STAGE_CHUNK, 0);
    if (e != hipSuccess) return hip_status(e, "host-tier staging buffers");
    pipe.crews(crew_helpers());
    CopyCrew& crew = *pipe.crew_out;
    const int n_chunks = (int)((bytes + STAGE_CHUNK - 1) / STAGE_CHUNK);
    auto issue = [&](int c) {
        Slot& s = pipe.slot[c % HostPipe::NSLOT];
        const size_t off = (size_t)c * STAGE_CHUNK, n = std::min(STAGE_CHUNK, bytes - off);
        hipError_t err = hipMemcpyAsync(s.pin_out, (const char*)dev + off, n, hipMemcpyDeviceToHost, s.stream);
        if (err == hipSuccess) err = hipEventRecord(s.done, s.stream);
        return err;
    };
    for (int c = 0; c < std::min(n_chunks, (int)HostPipe::NSLOT) && e == hipSuccess; c++) e = issue(c);
    for (int c = 0; c < n_chunks && e == hipSuccess; c++) {
        Slot& s = pipe.slot[c % HostPipe::NSLOT];
        const size_t off = (size_t)c * STAGE_CHUNK, n = std::min(STAGE_CHUNK, bytes - off);
        e = hipEventSynchronize(s.done);
        if (e != hipSuccess) break;
        crew.copy((char*)host + off, s.pin_out, n);
        if (c + HostPipe::NSLOT < n_chunks) e = issue(c + HostPipe::NSLOT);
    }
    for (Slot& s : pipe.slot) {
        const hipError_t e2 = hipStreamSynchronize(s.stream);
        if (e == hipSuccess) e = e2;
    }
    return hip_status(e, "staged device-to-host copy");
}

// imageLoadWrapper's shape (imageTools.cu:158-183: H2D, the repack kernel -- a copy --, D2H) through the
// pipe's cached buffers, a chunk at a time: chunk c is staged, copied up, repacked on the device and copied
// down on slot c % 3's stream while the host stages the next one; its pinned output is unloaded when the slot
// comes round again.  Nothing is allocated per call (the reference allocates two managed buffers per call).
int host_repack(const void* src, void* dst, size_t bytes)
{
    int device = 0;
    int rc = current_device(&device);
    if (rc) return rc;
    HostPipe& pipe = g_pipes[device];
    std::lock_guard<std::mutex> lock(pipe.mu);
    const size_t chunk = std::min(bytes, STAGE_CHUNK);
    hipError_t e = pipe.reserve(chunk, chunk, chunk);
    if (e != hipSuccess) return hip_status(e, "host-tier staging buffers");
    pipe.crews(crew_helpers());
    constexpr int NSLOT = HostPipe::NSLOT;
    const int n_chunks = (int)((bytes + STAGE_CHUNK - 1) / STAGE_CHUNK);
    auto span = [&](int c, size_t* off, size_t* n) { *off = (size_t)c * STAGE_CHUNK; *n = std::min(STAGE_CHUNK, bytes - *off); };
    auto unload = [&](int c) {
        Slot& s = pipe.slot[c % NSLOT];
        size_t off, n;
        span(c, &off, &n);
        const hipError_t err = hipEventSynchronize(s.done);
        if (err == hipSuccess) pipe.crew_out->copy((char*)dst + off, s.pin_out, n);
        return err;
    };
    for (int c = 0; c < n_chunks && e == hipSuccess && rc == MI_OK; c++) {
        Slot& s = pipe.slot[c % NSLOT];
        if (c >= NSLOT) e = unload(c - NSLOT);
        if (e != hipSuccess) break;
        size_t off, n;
        span(c, &off, &n);
        pipe.crew_in->copy(s.pin_in, (const char*)src + off, n);
        e = hipMemcpyAsync(s.dev[0], s.pin_in, n, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) rc = launch_copy(s.dev[0], s.dev[1], n, s.stream);
        if (e == hipSuccess && rc == MI_OK) e = hipMemcpyAsync(s.pin_out, s.dev[1], n, hipMemcpyDeviceToHost, s.stream);
        if (e == hipSuccess && rc == MI_OK) e = hipEventRecord(s.done, s.stream);
    }
    for (int c = std::max(0, n_chunks - NSLOT); c < n_chunks && e == hipSuccess && rc == MI_OK; c++) e = unload(c);
    if (rc != MI_OK || e != hipSuccess)
        for (Slot& s : pipe.slot) if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (rc) return rc;
    return hip_status(e, "host-tier repack pipeline");
}

// ops are validated by the caller (api.hip).
int host_run_ops(const uint8_t* in, uint8_t* out, int w, int h, const mi_op* ops, int n_ops, int chunk_rows_hint)
{
    return host_run_ops_rows(in, out, w, h, ops, n_ops, chunk_rows_hint, 0, h);
}

// Output rows [y0, y1) of the filtered image only (`in` and `out` are whole images): the piece of the work one
// device does when an image is split into row strips; the halo rows come straight from the host image.
int host_run_ops_rows(const uint8_t* in, uint8_t* out, int w, int h, const mi_op* ops, int n_ops, int chunk_rows_hint,
                      int y0, int y1)
{
    const size_t pitch = (size_t)w * 4;
    int halo_sum = 0;
    for (int k = 0; k < n_ops; k++) halo_sum += filter_halo_rows(ops[k].filter, ops[k].ksize, w);
    // chunk: about 8 MiB of output rows, never fewer rows than the accumulated halo (so that a
    // chunk's input never reaches into rows an earlier chunk may already have overwritten when
    // `out` aliases `in`, see below)
    int chunk = chunk_rows_hint > 0 ? chunk_rows_hint : (int)std::max<size_t>(1, ((size_t)8 << 20) / pitch);
    chunk = std::min(y1 - y0, std::max(chunk, std::max(halo_sum, 1)));
    const int n_chunks = (y1 - y0 + chunk - 1) / chunk;
    const size_t in_rows_max = (size_t)std::min(h, chunk + 2 * halo_sum);

    int device = 0;
    int rc = current_device(&device);
    if (rc) return rc;
    HostPipe& pipe = g_pipes[device];
    std::lock_guard<std::mutex> lock(pipe.mu);
    hipError_t e = pipe.reserve(in_rows_max * pitch, (size_t)chunk * pitch, in_rows_max * pitch);
    if (e != hipSuccess) return hip_status(e, "host-tier staging buffers");

    constexpr int NSLOT = HostPipe::NSLOT;
    std::vector<Rows> need(n_ops + 1);
    std::vector<Rows> out_rows(n_chunks);

    // The unloading side: chunk c leaves its slot's pinned buffer for `out` once the device is done
    // with it AND the main thread has staged chunk c+1 -- when `out` aliases `in`, chunk c+1's input
    // reaches into the rows chunk c overwrites (never further: a chunk is at least halo_sum rows).
    // (the pipe's crews; a call of one or two chunks copies on its own threads)
    pipe.crews(crew_helpers());
    CopyCrew solo_in(0), solo_out(0);
    CopyCrew& crew_in = n_chunks >= 3 ? *pipe.crew_in : solo_in;
    CopyCrew& crew_out = n_chunks >= 3 ? *pipe.crew_out : solo_out;
    std::atomic<int> submitted{0}, unloaded{0};
    std::atomic<bool> abort_flag{false};
    hipError_t unload_err = hipSuccess;
    auto unload_all = [&]() {
        for (int c = 0; c < n_chunks; c++) {
            while (submitted.load(std::memory_order_acquire) < std::min(c + 2, n_chunks)) {
                if (abort_flag.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            Slot& s = pipe.slot[c % NSLOT];
            const hipError_t err = hipEventSynchronize(s.done);
            if (err != hipSuccess) { unload_err = err; abort_flag.store(true, std::memory_order_release); return; }
            const Rows r = out_rows[c];
            crew_out.copy(out + (size_t)r.lo * pitch, s.pin_out, (size_t)(r.hi - r.lo) * pitch);
            unloaded.store(c + 1, std::memory_order_release);
        }
    };
    // one chunk: nothing to overlap, stay on this thread
    std::thread helper;
    if (n_chunks > 1) helper = std::thread([&] { (void)hipSetDevice(device); unload_all(); });

    for (int c = 0; c < n_chunks && rc == MI_OK && e == hipSuccess; c++) {
        Slot& s = pipe.slot[c % NSLOT];
        // the slot's previous occupant (chunk c - NSLOT) must have left its pinned output buffer
        while (unloaded.load(std::memory_order_acquire) < c - NSLOT + 1 && !abort_flag.load(std::memory_order_acquire))
            std::this_thread::yield();
        if (abort_flag.load(std::memory_order_acquire)) break;
        // rows each stage must produce, back to front
        need[n_ops] = Rows{y0 + c * chunk, std::min(y1, y0 + (c + 1) * chunk)};
        for (int k = n_ops - 1; k >= 0; k--) {
            const int halo = filter_halo_rows(ops[k].filter, ops[k].ksize, w);
            need[k] = Rows{std::max(0, need[k + 1].lo - halo), std::min(h, need[k + 1].hi + halo)};
        }
        out_rows[c] = need[n_ops];
        crew_in.copy(s.pin_in, in + (size_t)need[0].lo * pitch, (size_t)(need[0].hi - need[0].lo) * pitch);
        e = hipMemcpyAsync(s.dev[0], s.pin_in, (size_t)(need[0].hi - need[0].lo) * pitch, hipMemcpyHostToDevice, s.stream);
        int cur = 0;
        for (int k = 0; k < n_ops && rc == MI_OK && e == hipSuccess; k++) {
            FilterArgs a;
            a.filter = ops[k].filter;
            if (a.filter == MI_FILTER_GAUSSIAN && (ops[k].ksize == 1 || ops[k].ksize > 31)) a.filter = MI_FILTER_GAUSSIAN_DIRECT;
            a.in = s.dev[cur]; a.out = s.dev[cur ^ 1]; a.w = w; a.h_total = h; a.batch = 1;
            a.in_row0 = need[k].lo; a.in_rows = need[k].hi - need[k].lo;
            a.y_begin = need[k + 1].lo; a.y_end = need[k + 1].hi;
            a.ksize = ops[k].ksize; a.sigma = ops[k].sigma; a.stream = s.stream;
            rc = launch_filter(a);
            cur ^= 1;
        }
        if (rc == MI_OK && e == hipSuccess)
            e = hipMemcpyAsync(s.pin_out, s.dev[cur], (size_t)(need[n_ops].hi - need[n_ops].lo) * pitch, hipMemcpyDeviceToHost, s.stream);
        if (rc == MI_OK && e == hipSuccess) e = hipEventRecord(s.done, s.stream);
        if (rc == MI_OK && e == hipSuccess) submitted.store(c + 1, std::memory_order_release);
    }
    if (rc != MI_OK || e != hipSuccess) abort_flag.store(true, std::memory_order_release);
    if (helper.joinable()) helper.join();
    else if (rc == MI_OK && e == hipSuccess) unload_all();
    if (rc == MI_OK && e == hipSuccess && unload_err != hipSuccess) e = unload_err;
    if (rc != MI_OK || e != hipSuccess) {
        // leave nothing in flight that still points at the staging buffers
        for (Slot& s : pipe.slot) if (s.stream) (void)hipStreamSynchronize(s.stream);
    }
    if (rc) return rc;
    return hip_status(e, "host-tier filter pipeline");
}

// ---------------------------------------------------------------- several devices, one process
// One host thread per entry of `devices` (an ordinal may repeat: its threads then take turns on that
// device's pipe).  work(d, index) runs with device devices[index] current; the first failure wins
// and its message is handed to the calling thread.
template <class Work>
static int run_on_devices(const int* devices, int n_devices, Work work)
{
    int keep = 0;
    const bool have = hipGetDevice(&keep) == hipSuccess;
    std::vector<int> rcs(n_devices, MI_OK);
    std::vector<std::string> msgs(n_devices);
    std::vector<std::thread> threads;
    for (int i = 0; i < n_devices; i++)
        threads.emplace_back([&, i] {
            t_device_share = n_devices;
            const hipError_t e = hipSetDevice(devices[i]);
            rcs[i] = e == hipSuccess ? work(i) : hip_status(e, "hipSetDevice");
            if (rcs[i] != MI_OK) msgs[i] = mi_last_error();      // the error text is per thread
        });
    for (std::thread& t : threads) t.join();
    if (have) (void)hipSetDevice(keep);
    for (int i = 0; i < n_devices; i++)
        if (rcs[i] != MI_OK) return fail(rcs[i], "device %d: %s", devices[i], msgs[i].c_str());
    return MI_OK;
}

int host_batch_run(const uint8_t* in, uint8_t* out, int w, int h, int n_frames, const mi_op* ops, int n_ops,
                   const int* devices, int n_devices)
{
    const size_t frame = (size_t)w * h * 4;
    const int per = (n_frames + n_devices - 1) / n_devices;       // contiguous ceil(n/G) frames per device
    return run_on_devices(devices, n_devices, [&](int i) {
        const int f0 = std::min(n_frames, i * per), f1 = std::min(n_frames, f0 + per);
        for (int f = f0; f < f1; f++) {
            const int rc = host_run_ops(in + (size_t)f * frame, out + (size_t)f * frame, w, h, ops, n_ops, 0);
            if (rc) return rc;
        }
        return (int)MI_OK;
    });
}

int host_strip_run(const uint8_t* in, uint8_t* out, int w, int h, const mi_op* ops, int n_ops, const int* devices, int n_devices)
{
    return run_on_devices(devices, n_devices, [&](int i) {
        const int y0 = (int)((int64_t)h * i / n_devices), y1 = (int)((int64_t)h * (i + 1) / n_devices);
        if (y0 >= y1) return (int)MI_OK;
        return host_run_ops_rows(in, out, w, h, ops, n_ops, 0, y0, y1);
    });
}

}  // namespace mi
