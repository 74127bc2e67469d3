#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric: megapixels/s of the filter chain
(gray -> Gaussian 5x5 -> median 3x3 -> Sobel) on 3840x2160 RGBA u8 frames, device-resident.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--unfused] [--legs c3,c4,c5]

A step is one pass of the chain over one batch of B synthetic frames per GPU (inputs already in
HBM): one launch of the fused chain kernel (mi_chain_gray_gauss5_median3_sobel_u8x4_dev, output
bit-identical to the four single-filter kernels), or with --unfused the four kernels in turn.
N>1 is launched by torch.distributed.run, one process per GPU; frames are independent, so ranks
share nothing on the data path (batch sharding, weak scaling) and RCCL is used only for the
barrier and the max-over-ranks time.  Rank 0 prints one JSON line.

Extra objects in the line:
  roofline      the kernel of the timed step (the slowest of the four with --unfused):
                algorithmic bytes (8 B/pixel: 4 read + 4 written) per launch / its
                average launch duration, measured with HIP events on the launch stream inside
                the timed region, against 8 TB/s.  `traffic` is NOT measured by this run: it is
                the rocprofv3 --pmc figure for the same kernel and workload kept in profiles/
                (`traffic_source` names the file and its date).
  unfused       the same chain as four kernels (32 B/pixel through HBM), timed right after the
                K steps with the same events and the same spin-up: per-filter ms, MP/s, GB/s.
  c3, c4, c5    BASELINE.json's other configurations at this N (--legs, default all; each has its own
                `roofline`): c3 = 4096 1920x1080 frames batch-sharded over the N GPUs (strong scaling);
                c4 = one 32768x32768 image in N row strips, Gaussian 15x15 and median 5x5, with the
                RCCL neighbour halo exchange inside the timed region; c5 = 2-D FFT + inverse FFT of
                16 4096x4096 planes per GPU (weak scaling).
  cpu_baseline  the CPU oracle (a scalar C port of the reference's arithmetic) timed on this
                host on a bounded sample of the same workload, rank 0 at N=1 only.
  c1            BASELINE.json config 1 (rank 0 at N=1): the reference's sample image deer.jpeg decoded with
                its stb_image, oracle grayscale + Gaussian 5x5 (sigma 1), PNG out -- ms per stage on one
                host thread, and the same image through the library's host tier beside it.
  checksum      CRC-32 of what the timed launches wrote (whole output for the headline; first, middle
                and last frame or row band plus a device-side byte sum for the big legs), so the line can
                be tied to the outputs the tests pin: tests/test_gpu_bench_checksums.py recomputes them
                and holds them against tests/golden/bench_checksums.json and the oracle.
  process_group backend and world size as torch.distributed reports them.
"""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 3840, 2160
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PER_PIXEL = 8.0           # algorithmic: 4 read + 4 written per u8x4 -> u8x4 filter
FFT_BYTES_PER_PIXEL = 64.0      # forward + inverse 2-D C2C fp32: 2 directions x 2 line passes x (8 R + 8 W)
STAGES = ["grayscale", "gaussian5", "median3", "sobel"]
KERNELS = {"grayscale": "gray_kernel", "gaussian5": "gauss_sep_kernel<2>", "median3": "median3_kernel",
           "sobel": "sobel_kernel", "fused": "chain_kernel"}
# HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
# FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) at the default workload
# (batch 16): see profiles/.  None until measured for a kernel.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_bytes_per_launch.json")
# Share of a launch during which the vector ALUs were issuing (rocprofv3 --pmc SQ_ACTIVE_INST_VALU, tools/
# valu_roofline.sh): a kernel at 0.9 is bound by its instruction count, whatever its HBM fraction says.
VALU_FILE = os.path.join(ROOT, "profiles", "valu_busy.json")
VALU_BOUND = 0.85               # at or above this share of vector-issue time a kernel is labelled "valu"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    """All hardware threads this process may use ('all hardware threads'): the affinity mask,
    capped by the box's CPU share where the harness exports one (OMP_NUM_THREADS: os.cpu_count() shows the
    whole machine there, of which a one-GPU box owns a fraction)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


def cpu_baseline(frames=1, threads=1):
    """Oracle chain on `frames` synthetic 3840x2160 frames (checker code used as the baseline)."""
    from oracle import golden
    imgs = [golden.synth(2, W, H, first_idx=f * W * H) for f in range(frames)]
    t0 = time.perf_counter()
    for img in imgs:
        golden.chain(img, threads=threads)
    dt = time.perf_counter() - t0
    return frames * W * H / 1e6 / dt, dt


class Timer:
    """Wall time of `steps` calls of fn between two barriers, MAX over ranks; also the mean duration of
    each call from events on the launch stream inside the timed region."""

    def __init__(self, torch, dist, dev, backend):
        self.torch, self.dist, self.dev, self.backend = torch, dist, dev, backend

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def run(self, fn, steps, warmup):
        torch = self.torch
        for _ in range(warmup):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        stream = torch.cuda.current_stream()
        self.barrier()
        t0 = time.perf_counter()
        for a, b in ev:
            a.record(stream)
            fn()
            b.record(stream)
        self.barrier()
        elapsed = self.max_over_ranks(time.perf_counter() - t0)
        call_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
        return elapsed, call_ms


_valu = None


def valu_info(kernel):
    global _valu
    if _valu is None:
        _valu = json.load(open(VALU_FILE)) if os.path.exists(VALU_FILE) else {"kernels": {}, "source": None}
    k = _valu["kernels"].get(kernel)
    return (k, _valu.get("source")) if k else (None, None)


def roofline(kernel, alg_bytes, ms, traffic=None, **extra):
    """`achieved`/`peak`/`frac` are always the HBM figures (algorithmic bytes over the launch time against
    8 TB/s).  `bound` says what actually limits the kernel: "valu" when the profile in profiles/ shows its
    vector ALUs issuing for >= 85 % of the launch (`valu_frac`, not measured by this run), else "hbm"."""
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    info, src = valu_info(kernel)
    bound = "valu" if info and info["valu_busy"] >= VALU_BOUND else "hbm"
    r = {"kernel": kernel, "bound": bound, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": ms}
    if info:
        r["valu_frac"] = info["valu_busy"]
        r["valu_source"] = f"not measured in this run: {src}"
    r.update(extra)
    return r


def crc32_tensor(t):
    """CRC-32 (zlib) of a device tensor's bytes."""
    return zlib.crc32(t.contiguous().view(-1).cpu().numpy().tobytes()) & 0xffffffff


def byte_sum(torch, t, chunk=1 << 30):
    """Sum of all bytes of a uint8 device tensor, accumulated on the device in int64."""
    flat = t.view(-1)
    total = 0
    for p in range(0, flat.numel(), chunk):
        total += int(flat[p:p + chunk].sum(dtype=torch.int64).item())
    return total


def leg_c1(ops=None):
    """C1: the reference's sample image deer.jpeg (2592x2000) decoded as the reference's loadImage
    decodes it (stb_image v2.30, main.cpp:12-26), oracle grayscale + Gaussian 5x5 sigma 1 on one host
    thread, PNG out (imageWriteWrapper writes a PNG whatever the name, imageTools.cu:152).  ms per stage,
    MP/s, CRC-32 of the output pixels; beside it the same two filters through the library's host tier
    (host pointers, PCIe both ways) with its output checked against the oracle's to +-1 (separable Gaussian)."""
    import tempfile
    import numpy as np
    from oracle import golden
    path = os.path.join(ROOT, "tests", "golden", "deer.jpeg")
    t0 = time.perf_counter()
    img = golden.stb_decode(path)
    t1 = time.perf_counter()
    h, w = img.shape[:2]
    gray = golden.grayscale(img)
    t2 = time.perf_counter()
    blur = golden.gaussian(gray, 5, 1.0)
    t3 = time.perf_counter()
    out = {"workload": f"tests/golden/deer.jpeg {w}x{h}: stb_image decode -> grayscale -> Gaussian 5x5 sigma 1.0 -> PNG",
           "cpu": {"kind": "port", "cores": 1, "decode_ms": (t1 - t0) * 1e3, "grayscale_ms": (t2 - t1) * 1e3,
                   "gaussian5_ms": (t3 - t2) * 1e3, "filters_MP/s": w * h / 1e6 / (t3 - t1),
                   "crc32_output": zlib.crc32(blur.tobytes()) & 0xffffffff}}
    if ops is not None:
        with tempfile.TemporaryDirectory() as tmp:
            png = os.path.join(tmp, "deer_gray_blur.png")
            t4 = time.perf_counter()
            ops.imageWriteWrapper(png, blur, w, h)
            out["cpu"]["png_write_ms"] = (time.perf_counter() - t4) * 1e3
            out["cpu"]["png_bytes"] = os.path.getsize(png)
        spec = ops.parse_ops("gray,gauss:5:1.0")
        got = np.empty_like(img)
        ops.run_ops(got, img, w, h, spec)                # first call: staging buffers are allocated
        t5 = time.perf_counter()
        for _ in range(5):
            ops.run_ops(got, img, w, h, spec)
        dt = (time.perf_counter() - t5) / 5
        diff = int(np.abs(got.astype(np.int16) - blur.astype(np.int16)).max())
        exact = np.empty_like(img)
        ops.run_ops(exact, img, w, h, ops.parse_ops("gray,gauss_direct:5:1.0"))
        out["gpu_host_tier"] = {"filters_ms": dt * 1e3, "filters_MP/s": w * h / 1e6 / dt,
                                "note": "mi_run_ops_u8x4 on pageable host buffers, PCIe both ways included",
                                "max_abs_diff_vs_cpu": diff, "crc32_output": zlib.crc32(got.tobytes()) & 0xffffffff,
                                "direct_gaussian_bit_exact_vs_cpu": bool(np.array_equal(exact, blur))}
    return out


def leg_c3(torch, ops, timer, rank, world, dev, steps):
    """C3: 4096 synthetic 1920x1080 frames, contiguous ceil(4096/N) frames per GPU, no communication."""
    from cudaimageprocessing_amd import shard
    w, h, total = 1920, 1080, 4096
    f0, f1 = shard.batch_range(total, world, rank)
    n = f1 - f0
    frames = ops.synth(1000, w, h, batch=n, first_idx=f0 * w * h, device=dev)   # frame f = pixels [f*w*h, (f+1)*w*h) of stream 1000
    a, b = torch.empty_like(frames), torch.empty_like(frames)
    elapsed, ms = timer.run(lambda: ops.fused_chain(frames, 1.0, a), steps, 2)

    def four():
        ops.grayscale(frames, a)
        ops.gaussian_blur(a, 5, 1.0, b)
        ops.median_blur(b, 3, a)
        ops.sobel(a, b)
    elapsed4, ms4 = timer.run(four, steps, 1)
    px = n * w * h
    ops.fused_chain(frames, 1.0, a)
    picks = sorted({0, n // 2, n - 1})
    checksum = {"what": "fused-chain output of this rank's frames: CRC-32 of frames (global index) and the sum of all bytes",
                "crc32_frames": {str(f0 + i): crc32_tensor(a[i]) for i in picks}, "byte_sum": byte_sum(torch, a)}
    return {"workload": f"{total} frames {w}x{h} RGBA u8, frames [{f0},{f1}) on rank 0, fused chain, one launch per step",
            "scaling": "strong", "frames_per_gpu": n, "steps": steps, "checksum": checksum,
            "value": total * w * h / 1e6 * steps / elapsed, "unit": "MP/s", "ms_per_step": elapsed / steps * 1e3,
            "roofline": roofline("chain_kernel", px * BYTES_PER_PIXEL, ms),
            "unfused": {"value": total * w * h / 1e6 * steps / elapsed4, "unit": "MP/s", "ms_per_step": elapsed4 / steps * 1e3,
                        "GB/s_rank0": px * 4 * BYTES_PER_PIXEL / (ms4 * 1e-3) / 1e9}}


def leg_c4(torch, ops, timer, rank, world, dev, steps):
    """C4: one 32768x32768 image in N row strips; Gaussian 15x15 (sigma 2.5) and median 5x5, each a stage
    of shard.StripPlan: neighbour halo exchange (RCCL send/recv), interior rows meanwhile, edge bands."""
    from cudaimageprocessing_amd import _lib, shard
    w = h = 32768
    filters = {"gaussian15": (_lib.FILTER_GAUSSIAN, 15, 2.5, "gauss_mfma_kernel"),
               "median5": (_lib.FILTER_MEDIAN, 5, 0.0, "median5_kernel")}
    pad = max(shard._halo_rows(f, k) for f, k, _, _ in filters.values())
    plan = shard.StripPlan(w, h, rank, world, pad, dev)
    strip = plan.strip(0)
    ops_chunk = 1 << 28       # pixels per generator launch
    flat = strip.view(-1, 4)
    for p0 in range(0, plan.n * w, ops_chunk):
        cnt = min(ops_chunk, plan.n * w - p0)
        _lib.check(_lib.lib().mi_synth_u8x4_dev(flat[p0:].data_ptr(), 4, plan.y0 * w + p0, cnt, None))
    torch.cuda.synchronize()
    out = {"workload": f"{w}x{h} RGBA u8 in {world} row strip(s) of {plan.n} rows, halo exchange inside the timed region",
           "scaling": "strong", "steps": steps, "halo_rows": pad, "halo_bytes_per_direction": pad * w * 4, "per_filter": {}}
    for name, (fid, k, sigma, kernel) in filters.items():
        def stage():
            plan.cur = 0                      # the input stays in buffer 0; a stage writes buffer 1
            plan.run_stage(fid, k, sigma)
        elapsed, ms = timer.run(stage, steps, 3)
        halo = shard._halo_rows(fid, k)
        # the same rows as ONE launch with the halos already in place (no exchange): the compute alone
        a = halo if rank > 0 else 0
        b = halo if rank < world - 1 else 0
        src, dst = plan.bufs[0], plan.bufs[1]

        def compute_only():
            ops.filter_strip(fid, src[pad - a:pad + plan.n + b], dst[pad:pad + plan.n], w, h, plan.y0 - a, plan.y0, plan.y1, k, sigma)
        _, ms_compute = timer.run(compute_only, steps, 3)

        def exchange_only():
            sends, recvs = [], []
            if a:
                sends.append((src[pad:pad + halo], rank - 1)); recvs.append((src[pad - halo:pad], rank - 1))
            if b:
                sends.append((src[pad + plan.n - halo:pad + plan.n], rank + 1)); recvs.append((src[pad + plan.n:pad + plan.n + halo], rank + 1))
            shard._p2p(sends, recvs)()
        ms_halo = 0.0
        if world > 1:
            _, ms_halo = timer.run(exchange_only, steps, 3)
        stage()
        res = plan.strip(1)                       # this rank's rows of the stage's output
        bands = sorted({0, (plan.n // 2) // 64 * 64, plan.n - 64})
        checksum = {"what": "this rank's output rows: CRC-32 of 64-row bands (first global row) and the sum of all bytes",
                    "crc32_bands": {str(plan.y0 + y): crc32_tensor(res[y:y + 64]) for y in bands}, "byte_sum": byte_sum(torch, res)}
        # one GPU: the strip is one launch, the kernel's own time.  Several: the stage as timed -- halo exchange,
        # interior launch, edge-band launches -- is what the configuration asks for.
        roof_ms = ms if world > 1 else ms_compute
        out["per_filter"][name] = {
            "value": w * h / 1e6 * steps / elapsed, "unit": "MP/s", "ms_per_step": elapsed / steps * 1e3,
            "stage_ms_rank0": ms, "compute_only_ms_rank0": ms_compute, "halo_exchange_only_ms_rank0": ms_halo,
            "checksum": checksum,
            "roofline": roofline(kernel, plan.n * w * BYTES_PER_PIXEL, roof_ms,
                                 note=("the whole stage on rank 0: halo exchange + interior launch + edge-band launches" if world > 1 else
                                       "this rank's strip as one launch; stage_ms adds the exchange and the edge-band launches"))}
    return out


def leg_c5(torch, ops, timer, rank, world, dev, steps):
    """C5: 2-D fp32 FFT + inverse FFT of 16 4096x4096 planes per GPU (batch-sharded: replicas only)."""
    n, batch = 4096, 16
    gray = ops.synth(5, n, n, batch=batch, first_idx=rank * batch * n * n, device=dev)[..., 0].to(torch.float32)
    x = torch.complex(gray, torch.zeros_like(gray))
    del gray
    f, y = torch.empty_like(x), torch.empty_like(x)

    def roundtrip():
        ops.fft2d(x, out=f)
        ops.fft2d(f, inverse=True, out=y)
    elapsed, ms = timer.run(roundtrip, steps, 3)
    px = batch * n * n
    # floating point: no byte checksum -- the round trip's error, Parseval and the DC coefficient instead
    err = max(float(torch.linalg.vector_norm((y[b] / (n * n) - x[b]).to(torch.complex128)) /
                    torch.linalg.vector_norm(x[b].to(torch.complex128))) for b in (0, batch - 1))
    e_x = float((x[0].real.double() ** 2).sum())
    e_f = float((f[0].real.double() ** 2 + f[0].imag.double() ** 2).sum())
    checksum = {"what": "round trip of planes 0 and last: max rel-L2 of ifft(fft(x))/N - x; Parseval and DC of plane 0",
                "roundtrip_rel_l2": err, "parseval_rel": abs(e_f - e_x * n * n) / (e_x * n * n),
                "dc_rel": abs(complex(f[0, 0, 0].item()) - float(x[0].real.double().sum())) / float(x[0].real.double().sum())}
    if err > 1e-5:
        raise SystemExit(f"C5 round trip error {err} above 1e-5")
    two_pass = ops.get_tuning("MI_FFT_TWOPASS") != 0
    kernel = ("fft2p_cols_kernel + fft2p_rows_kernel (2 passes per direction)" if two_pass
              else "fft_lines_kernel (3 line passes per direction)")
    return {"workload": f"fft2d + inverse fft2d, {batch} planes {n}x{n} complex fp32 per GPU, out of place, natural order in and out",
            "scaling": "weak", "steps": steps, "value": world * px / 1e6 * steps / elapsed, "unit": "MP/s",
            "ms_per_step": elapsed / steps * 1e3, "dtype": "f32", "checksum": checksum,
            "roofline": roofline(kernel, px * FFT_BYTES_PER_PIXEL, ms,
                                 note="64 B/px algorithmic: 2 directions x 2 passes x (8 B read + 8 B written); avg_launch_ms is the "
                                      "round trip (4 launches per plane group)")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=16, help="3840x2160 frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="time the four single-filter kernels as the step")
    ap.add_argument("--legs", default="c3,c4,c5", help="extra configurations to run after the headline one ('' = none)")
    ap.add_argument("--leg-steps", type=int, default=30)
    ap.add_argument("--event-every", type=int, default=int(os.environ.get("MI_BENCH_EVENT_EVERY", "0")),
                    help="0 (default): ONE pair of HIP events around the K timed launches of the fused step, average = "
                         "elapsed / K (launch gaps included); n > 0: a pair around every n-th step's launch (a pair costs "
                         "2-4 us, which lengthens both the step and the bracketed launch: 0.2145 / 0.2092 ms per step / per "
                         "launch with n = 1 against 0.2071 / 0.2070 with one pair).  The four-kernel step always brackets "
                         "every kernel.")
    args = ap.parse_args()

    import torch
    from cudaimageprocessing_amd import ops

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # MI_BENCH_BACKEND=gloo is a rehearsal mode for a box with fewer GPUs than ranks (ranks share devices,
    # the collectives go over the CPU); the real run is one rank per GPU over RCCL
    backend = os.environ.get("MI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # under torch.distributed.run the process group is made for one rank too, so that the RCCL set-up, the barrier
    # and the MAX all-reduce around the timed region are the code a one-GPU box can already run
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    timer = Timer(torch, dist, dev, backend)

    B = args.batch
    # frames [rank*B, (rank+1)*B) of the synthetic stream with seed 2
    frames = ops.synth(2, W, H, batch=B, first_idx=rank * B * W * H, device=dev)
    bufs = [torch.empty_like(frames) for _ in range(4)]

    def unfused_step(events=None):
        stream = torch.cuda.current_stream()
        if events is not None:
            events[0].record(stream)
        ops.grayscale(frames, bufs[0])
        if events is not None:
            events[1].record(stream)
        ops.gaussian_blur(bufs[0], 5, 1.0, bufs[1])
        if events is not None:
            events[2].record(stream)
        ops.median_blur(bufs[1], 3, bufs[2])
        if events is not None:
            events[3].record(stream)
        ops.sobel(bufs[2], bufs[3])
        if events is not None:
            events[4].record(stream)

    def fused_step(events=None):
        stream = torch.cuda.current_stream()
        if events is not None:
            events[0].record(stream)
        ops.fused_chain(frames, 1.0, bufs[3])
        if events is not None:
            events[1].record(stream)

    step = unfused_step if args.unfused else fused_step
    n_ev = 5 if args.unfused else 2

    def spin_up(fn, seconds=0.1):
        # Set-up, before the W warm-up steps: keep the device busy for ~0.1 s so that its clocks are
        # up.  A launch lasts 0.25 ms; the first ~50 ms of work after idle run about 20 % slower
        # (measured: 433 GP/s over steps 4..23, 533 GP/s in steady state), which would make a short
        # run a measurement of the power-state ramp rather than of the kernel.
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()

    spin_up(step)
    for _ in range(args.warmup):
        step()
    every = 1 if args.unfused else max(0, args.event_every)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(n_ev)] if every and k % every == 0 else None
          for k in range(args.steps)]
    region = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    launch_stream = torch.cuda.current_stream()
    timer.barrier()
    t0 = time.perf_counter()
    region[0].record(launch_stream)
    for k in range(args.steps):
        step(ev[k])
    region[1].record(launch_stream)
    timer.barrier()
    elapsed = timer.max_over_ranks(time.perf_counter() - t0)

    # per-kernel launch durations (ms) from the events recorded inside the timed region
    timed_ev = [e for e in ev if e is not None]
    if timed_ev:
        per_kernel = [sum(e[i].elapsed_time(e[i + 1]) for e in timed_ev) / len(timed_ev) for i in range(n_ev - 1)]
        launch_timing = f"HIP event pair around {'every' if every == 1 else 'every %d-th' % every} launch inside the timed region"
    else:
        per_kernel = [region[0].elapsed_time(region[1]) / args.steps]
        launch_timing = f"one HIP event pair around the {args.steps} launches of the timed region, elapsed / {args.steps} (launch gaps included)"

    # the four-kernel chain, timed after the K steps (same buffers, same events, same spin-up: the
    # kernels differ from the timed ones, so their first launches would otherwise be cold)
    if args.unfused:
        per_stage, unfused_elapsed = per_kernel, elapsed
    else:
        spin_up(unfused_step)
        for _ in range(max(5, args.warmup // 4)):
            unfused_step()
        ev4 = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            unfused_step(ev4[k])
        torch.cuda.synchronize()
        unfused_elapsed = time.perf_counter() - t1
        per_stage = [sum(ev4[k][i].elapsed_time(ev4[k][i + 1]) for k in range(args.steps)) / args.steps
                     for i in range(4)]

    # what the timed launches wrote (bufs[3], both for the fused and the unfused step)
    step()
    torch.cuda.synchronize()
    headline_checksum = {"what": "output of the timed step on rank 0 (all frames) and its first frame: CRC-32 (zlib)",
                         "crc32": crc32_tensor(bufs[3]), "crc32_frame0": crc32_tensor(bufs[3][0]),
                         "byte_sum": byte_sum(torch, bufs[3])}
    legs = {}
    del bufs[:3]
    for leg in [s for s in args.legs.split(",") if s]:
        fn = {"c3": leg_c3, "c4": leg_c4, "c5": leg_c5}.get(leg)
        if fn is None:
            raise SystemExit(f"unknown leg {leg!r} (known: c3, c4, c5)")
        torch.cuda.empty_cache()
        try:
            legs[leg] = fn(torch, ops, timer, rank, world, dev, args.leg_steps)
        except torch.OutOfMemoryError as e:          # a smaller or shared GPU: say so, keep the headline
            legs[leg] = {"skipped": f"out of device memory: {str(e)[:120]}"}
            if dist is not None:
                raise                                 # ranks must not diverge
        torch.cuda.empty_cache()

    if rank == 0:
        px_per_launch = B * W * H
        mp_total = world * B * W * H / 1e6 * args.steps
        alg_bytes = px_per_launch * BYTES_PER_PIXEL
        traffic, traffic_source = {}, None
        if os.path.exists(TRAFFIC_FILE) and B == 16:
            traffic = json.load(open(TRAFFIC_FILE))
            traffic_source = ("not measured in this run: profiles/traffic_bytes_per_launch.json (rocprofv3 --pmc FETCH_SIZE x2 + "
                              "WRITE_SIZE per launch, same workload), file dated "
                              + time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(TRAFFIC_FILE))))

        def gbs(ms):
            return alg_bytes / (ms * 1e-3) / 1e9

        per_filter = {STAGES[i]: {"kernel": KERNELS[STAGES[i]], "ms": per_stage[i],
                                  "MP/s": px_per_launch / 1e6 / (per_stage[i] * 1e-3), "GB/s": gbs(per_stage[i]),
                                  "roofline_frac": gbs(per_stage[i]) / HBM_PEAK_GBS,
                                  "traffic": traffic.get(KERNELS[STAGES[i]])}
                      for i in range(4)}
        if args.unfused:
            worst = max(range(4), key=lambda i: per_stage[i])
            kname, kms = KERNELS[STAGES[worst]], per_stage[worst]
            desc = "4 kernels per step (gray, gauss_sep<2>, median3, sobel)"
        else:
            kname, kms = KERNELS["fused"], per_kernel[0]
            desc = "1 fused kernel per step (output bit-identical to the 4 single-filter kernels)"
        out = {
            "metric": "megapixels/sec, full filter chain (gray, Gaussian 5x5, median 3x3, Sobel)",
            "value": mp_total / elapsed,
            "unit": "MP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 (fp32 accumulate in grayscale/Gaussian)",
            "data": "synthetic",
            "config": {"workload": f"C2 chain gray->gauss5(sigma 1.0)->median3->sobel, {W}x{H} RGBA u8, "
                                   f"batch {B} frames per GPU, device-resident, {desc}",
                       "frames_per_gpu": B, "width": W, "height": H, "sharding": f"batch x{world}"},
            "roofline": roofline(kname, alg_bytes, kms, traffic.get(kname), traffic_source=traffic_source,
                                 launch_timing=launch_timing),
            "checksum": headline_checksum,
            "process_group": {"backend": (dist.get_backend() if dist is not None else None),
                              "world_size": (dist.get_world_size() if dist is not None else 1)},
            "unfused": {"value": B * W * H / 1e6 * args.steps / unfused_elapsed, "unit": "MP/s (rank 0)",
                        "ms_per_step": unfused_elapsed / args.steps * 1e3, "per_filter": per_filter},
        }
        out.update(legs)
        if world == 1 and not args.no_cpu_baseline:
            v, dt = cpu_baseline(frames=8, threads=1)
            out["cpu_baseline"] = {"value": v, "unit": "MP/s", "cores": 1, "kind": "port",
                                   "sample": f"the same 4-stage chain on 8 synthetic {W}x{H} frames, "
                                             f"oracle/golden.c single thread, {dt:.1f} s"}
            nthr = host_threads()
            nfr = max(16, min(256, nthr))
            v2, dt2 = cpu_baseline(frames=nfr, threads=nthr)
            out["cpu_baseline_threads"] = {"value": v2, "unit": "MP/s", "cores": nthr, "kind": "port",
                                           "sample": f"{nfr} frames, row bands on {nthr} threads, {dt2:.1f} s",
                                           "cpu_model": cpu_model(), "hardware_threads": os.cpu_count()}
            out["c1"] = leg_c1(ops)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
