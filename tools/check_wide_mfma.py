#!/usr/bin/env python3
"""gauss_mfma_wide_kernel (sizes 19..31): the mean against the oracle (bit-exact) and against the prefix-sum march
(MI_MEAN_MFMA_MAX=17), the Gaussian within one level of the oracle; ms per launch on 16 x 3840x2160 by band height.
Usage: python tools/check_wide_mfma.py [--time-only]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cudaimageprocessing_amd import ops, synth as S
from oracle import golden

dev = torch.device("cuda:0")


def timed(fn, n=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ok = True
if "--time-only" not in sys.argv:
    for (w, h) in ((516, 75), (128, 64), (64, 48), (260, 130), (1920, 270), (8, 140), (20, 133), (132, 17)):
        for k in (19, 21, 23, 25, 27, 29, 31):
            if k * k > w * h:
                continue
            for kind in ("random", "constant255", "impulses"):
                if kind == "random":
                    img = S.synth_image(k + w, w, h)
                elif kind == "constant255":
                    img = np.full((h, w, 4), 255, np.uint8)
                else:
                    img = np.zeros((h, w, 4), np.uint8)
                    img[::37, ::29] = 255
                want = golden.meanblur(img, k, threads=8)
                d = torch.from_numpy(img).to(dev)
                got = ops.mean_blur(d, k).cpu().numpy()
                if not np.array_equal(got, want):
                    ok = False
                    bad = np.argwhere(got != want)
                    print(f"mean {w}x{h} k={k} {kind}: {len(bad)} bytes differ, first {bad[:3].tolist()} got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}", flush=True)
                with ops.tuning(MI_MEAN_MFMA_MAX=17):
                    old = ops.mean_blur(d, k).cpu().numpy()
                if not np.array_equal(old, want):
                    ok = False
                    print(f"mean {w}x{h} k={k} {kind}: the prefix-sum march differs from the oracle", flush=True)
    print("mean parity", "ok" if ok else "FAILED", flush=True)
    gok = True
    for (w, h) in ((516, 75), (128, 64), (260, 130), (1920, 270), (8, 140), (20, 133)):
        for k, sigma in ((17, 3.0), (19, 3.0), (21, 0.6), (23, 4.0), (25, 12.0), (27, 5.0), (29, 2.0), (31, 5.0), (31, 40.0)):
            if k * k > w * h:
                continue
            img = S.synth_image(k + w, w, h)
            want = golden.gaussian(img, k, sigma, threads=8).astype(np.int16)
            d = torch.from_numpy(img).to(dev)
            got = ops.gaussian_blur(d, k, sigma).cpu().numpy().astype(np.int16)
            e = np.abs(got - want)
            line = f"gauss {w}x{h} k={k} sigma={sigma}: max {e.max()} differ {100 * (e != 0).mean():.2f} %"
            if e.max() > 1:
                gok = False
                line += f"   BAD at {np.argwhere(e > 1)[:4].tolist()}"
            print(line, flush=True)
    for v in (0, 1, 77, 128, 254, 255):
        const = torch.full((100, 516, 4), v, dtype=torch.uint8, device=dev)
        for k, sigma in ((17, 2.0), (31, 2.0), (23, 0.7), (31, 20.0)):
            r = k // 2
            o = ops.gaussian_blur(const, k, sigma)
            inner = o[r + 1:100 - r - 1, r:516 - r, :3]
            if int(inner.min()) != v or int(inner.max()) != v:
                gok = False
                print(f"constant {v} k={k} sigma={sigma}: {int(inner.min())}..{int(inner.max())}")
    print("gauss parity", "ok" if gok else "FAILED", flush=True)
    ok = ok and gok

W, H, B = 3840, 2160, 16
frames = ops.synth(2, W, H, batch=B)
out = torch.empty_like(frames)
for _ in range(300):
    ops.fused_chain(frames, 1.0, out)
for k in (19, 23, 27, 31):
    cells = []
    for rpb in (0, 64, 128, 256, 512):
        with ops.tuning(MI_ROWS_PER_BAND=rpb):
            t = min(timed(lambda: ops.mean_blur(frames, k, out)) for _ in range(2))
        cells.append(f"rpb {rpb}: {t:.4f} ms ({B * W * H * 8 / (t * 1e-3) / 8e12:.3f})")
    with ops.tuning(MI_MFMA_SERPENTINE=0):
        t = min(timed(lambda: ops.mean_blur(frames, k, out)) for _ in range(2))
    cells.append(f"all bands downwards: {t:.4f} ms")
    with ops.tuning(MI_MEAN_MFMA_MAX=17):
        t = min(timed(lambda: ops.mean_blur(frames, k, out)) for _ in range(2))
    cells.append(f"prefix march: {t:.4f} ms ({B * W * H * 8 / (t * 1e-3) / 8e12:.3f})")
    print(f"16 x 4K mean {k}: " + "  ".join(cells), flush=True)
for k, sigma in ((17, 3.0), (23, 4.0), (31, 5.0)):
    cells = []
    for rpb in (0, 128, 256):
        with ops.tuning(MI_ROWS_PER_BAND=rpb):
            t = min(timed(lambda: ops.gaussian_blur(frames, k, sigma, out)) for _ in range(2))
        cells.append(f"rpb {rpb}: {t:.4f} ms ({B * W * H * 8 / (t * 1e-3) / 8e12:.3f})")
    with ops.tuning(MI_MFMA_SERPENTINE=0):
        t = min(timed(lambda: ops.gaussian_blur(frames, k, sigma, out)) for _ in range(2))
    cells.append(f"all bands downwards: {t:.4f} ms")
    with ops.tuning(MI_GAUSS_MFMA_MAX=15):
        t = timed(lambda: ops.gaussian_blur(frames, k, sigma, out), n=5)
    cells.append(f"direct: {t:.3f} ms")
    print(f"16 x 4K gauss {k}: " + "  ".join(cells), flush=True)
sys.exit(0 if ok else 1)
