#!/usr/bin/env python3
"""gauss_mfma_kernel (MI_GAUSS_MFMA=1) against the oracle and against gauss_wide_kernel (MI_GAUSS_MFMA=0):
max |delta|, share of differing bytes, constants, strips, and ms per launch on 16 x 3840x2160 / 32768^2.
Usage: python tools/check_gauss_mfma.py [--big]"""
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
for (w, h) in ((516, 75), (128, 64), (64, 48), (260, 130), (1920, 270), (4, 40), (8, 33)):
    for k, sigma in ((7, 1.5), (9, 2.0), (11, 2.0), (13, 3.0), (15, 2.5), (15, 10.0), (7, 0.05)):
        if k * k > w * h:
            continue
        img = S.synth_image(k + w, w, h)
        want = golden.gaussian(img, k, sigma).astype(np.int16)
        d = torch.from_numpy(img).to(dev)
        got = ops.gaussian_blur(d, k, sigma).cpu().numpy().astype(np.int16)
        with ops.tuning(MI_GAUSS_MFMA=0):
            wide = ops.gaussian_blur(d, k, sigma).cpu().numpy().astype(np.int16)
        e = np.abs(got - want)
        e2 = np.abs(wide - want)
        line = f"{w}x{h} k={k} sigma={sigma}: mfma max {e.max()} differ {100 * (e != 0).mean():.2f} %   wide max {e2.max()} differ {100 * (e2 != 0).mean():.2f} %"
        if e.max() > 1:
            ok = False
            bad = np.argwhere(e > 1)
            line += f"   BAD at {bad[:4].tolist()} of {len(bad)}"
        print(line, flush=True)

for v in (0, 1, 77, 128, 254, 255):
    const = torch.full((75, 516, 4), v, dtype=torch.uint8, device=dev)
    for k, sigma in ((15, 2.0), (7, 2.0), (11, 0.7), (15, 10.0)):
        r = k // 2
        out = ops.gaussian_blur(const, k, sigma)
        inner = out[r + 1:75 - r - 1, r:516 - r, :3]
        if int(inner.min()) != v or int(inner.max()) != v:
            ok = False
            print(f"constant {v} k={k} sigma={sigma}: {int(inner.min())}..{int(inner.max())}")
print("parity", "ok" if ok else "FAILED", flush=True)

W, H, B = 3840, 2160, 16
frames = ops.synth(2, W, H, batch=B)
out = torch.empty_like(frames)
for k, sigma in ((7, 1.5), (9, 2.0), (11, 2.0), (13, 2.2), (15, 2.5)):
    cells = []
    for rpb in (0, 32, 64, 128, 256):
        with ops.tuning(MI_ROWS_PER_BAND=rpb):
            t = min(timed(lambda: ops.gaussian_blur(frames, k, sigma, out)), timed(lambda: ops.gaussian_blur(frames, k, sigma, out)))
        cells.append(f"rpb {rpb}: {t:.4f} ms ({B * W * H * 8 / (t * 1e-3) / 8e12:.3f})")
    with ops.tuning(MI_GAUSS_MFMA=0):
        t = timed(lambda: ops.gaussian_blur(frames, k, sigma, out))
    cells.append(f"wide: {t:.4f} ms ({B * W * H * 8 / (t * 1e-3) / 8e12:.3f})")
    print(f"16 x 4K k={k}: " + "  ".join(cells), flush=True)

if "--big" in sys.argv:
    del frames, out
    big = ops.synth(4, 32768, 32768)
    out = torch.empty_like(big)
    for rpb in (0, 64, 128, 256):
        with ops.tuning(MI_ROWS_PER_BAND=rpb):
            t = timed(lambda: ops.gaussian_blur(big, 15, 2.5, out), n=10)
        print(f"32768^2 k=15 rpb {rpb}: {t:.4f} ms ({32768 * 32768 * 8 / (t * 1e-3) / 8e12:.3f})", flush=True)
    with ops.tuning(MI_GAUSS_MFMA=0):
        t = timed(lambda: ops.gaussian_blur(big, 15, 2.5, out), n=10)
    print(f"32768^2 k=15 wide: {t:.4f} ms ({32768 * 32768 * 8 / (t * 1e-3) / 8e12:.3f})", flush=True)
