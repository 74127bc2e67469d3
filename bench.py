#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric: megapixels/s of the filter chain
(gray -> Gaussian 5x5 -> median 3x3 -> Sobel) on 3840x2160 RGBA u8 frames, device-resident.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--unfused]

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
                the timed region, against 8 TB/s.
  unfused       the same chain as four kernels (32 B/pixel through HBM), timed right after the
                K steps with the same events: per-filter ms, MP/s, GB/s and roofline fraction.
  cpu_baseline  the CPU oracle (a scalar C port of the reference's arithmetic) timed on this
                host on a bounded sample of the same workload, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 3840, 2160
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PER_PIXEL = 8.0           # algorithmic: 4 read + 4 written per u8x4 -> u8x4 filter
STAGES = ["grayscale", "gaussian5", "median3", "sobel"]
KERNELS = {"grayscale": "gray_kernel", "gaussian5": "gauss_sep_kernel<2>", "median3": "median3_kernel",
           "sobel": "sobel_kernel", "fused": "chain_kernel"}
# HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
# FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) at the default workload
# (batch 16): see profiles/.  None until measured for a kernel.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_bytes_per_launch.json")


def cpu_baseline(frames=1, threads=1):
    """Oracle chain on `frames` synthetic 3840x2160 frames (checker code used as the baseline)."""
    from oracle import golden
    imgs = [golden.synth(2, W, H, first_idx=f * W * H) for f in range(frames)]
    t0 = time.perf_counter()
    for img in imgs:
        golden.chain(img, threads=threads)
    dt = time.perf_counter() - t0
    return frames * W * H / 1e6 / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=16, help="3840x2160 frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="time the four single-filter kernels as the step")
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
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B = args.batch
    # frames [rank*B, (rank+1)*B) of the synthetic stream with seed 2
    frames = ops.synth(2, W, H, batch=B, first_idx=rank * B * W * H, device=dev)
    bufs = [torch.empty_like(frames) for _ in range(4)]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

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

    # Set-up, before the W warm-up steps: keep the device busy for ~0.1 s so that its clocks are
    # up.  A launch lasts 0.25 ms; the first ~50 ms of work after idle run about 20 % slower
    # (measured: 433 GP/s over steps 4..23, 533 GP/s in steady state), which would make a short
    # run a measurement of the power-state ramp rather than of the kernel.
    for _ in range(400):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(n_ev)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel launch durations (ms) from the events recorded inside the timed region
    per_kernel = [sum(ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(args.steps)) / args.steps
                  for i in range(n_ev - 1)]

    # the four-kernel chain, timed after the K steps (same buffers, same events)
    if args.unfused:
        per_stage, unfused_elapsed = per_kernel, elapsed
    else:
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

    if rank == 0:
        px_per_launch = B * W * H
        mp_total = world * B * W * H / 1e6 * args.steps
        alg_bytes = px_per_launch * BYTES_PER_PIXEL
        traffic = {}
        if os.path.exists(TRAFFIC_FILE) and B == 16:
            traffic = json.load(open(TRAFFIC_FILE))

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
            "roofline": {"kernel": kname, "bound": "hbm", "achieved": gbs(kms),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs(kms) / HBM_PEAK_GBS,
                         "traffic": traffic.get(kname),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kms},
            "unfused": {"value": B * W * H / 1e6 * args.steps / unfused_elapsed, "unit": "MP/s (rank 0)",
                        "ms_per_step": unfused_elapsed / args.steps * 1e3, "per_filter": per_filter},
        }
        if world == 1 and not args.no_cpu_baseline:
            v, dt = cpu_baseline(frames=8, threads=1)
            out["cpu_baseline"] = {"value": v, "unit": "MP/s", "cores": 1, "kind": "port",
                                   "sample": f"the same 4-stage chain on 8 synthetic {W}x{H} frames, "
                                             f"oracle/golden.c single thread, {dt:.1f} s"}
            nthr = min(16, os.cpu_count() or 1)
            v2, dt2 = cpu_baseline(frames=16, threads=nthr)
            out["cpu_baseline_threads"] = {"value": v2, "unit": "MP/s", "cores": nthr, "kind": "port",
                                           "sample": f"16 frames, row bands on {nthr} threads, {dt2:.1f} s"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
