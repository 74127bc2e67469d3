#!/usr/bin/env python3
"""Direct (reference-order) Gaussians 3x3 .. 9x9: bit-exact against the oracle on odd and aligned sizes,
then the time per 16 x 4K batch.  Usage: python tools/check_gauss_direct.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cudaimageprocessing_amd import ops, synth as S  # noqa: E402
from oracle import golden  # noqa: E402

for (w, h) in ((640, 96), (517, 73), (259, 40), (64, 20), (1024, 33)):
    img = S.synth_image(w + h, w, h)
    d = torch.from_numpy(img).cuda()
    for k, sg in ((3, 7.0), (5, 1.0), (7, 1.5), (9, 2.0)):
        got = ops.gaussian_blur(d, k, sg, direct=True).cpu().numpy()
        want = golden.gaussian(img, k, sg, threads=8)
        bad = int((got != want).sum())
        print(f"{w}x{h} k={k}: {'bit-exact' if bad == 0 else f'{bad} bytes differ'}", flush=True)
        assert bad == 0

fr = ops.synth(2, 3840, 2160 * 16).reshape(16, 2160, 3840, 4)
out = torch.empty_like(fr)
def timed(fn, warm=30, reps=50):     # (warm-up: the clocks take some 10 ms to settle under an arithmetic-bound kernel)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


px = 16 * 3840 * 2160
for k, sg in ((3, 7.0), (5, 1.0), (7, 1.5), (9, 2.0)):
    ms = timed(lambda: ops.gaussian_blur(fr, k, sg, out, direct=True))
    line = f"direct Gaussian {k}x{k}: {ms:.3f} ms per 16 x 4K = {px * 8 / ms / 1e6:.0f} GB/s ({px * 8 / ms / 1e6 / 80:.1f} % of 8 TB/s)"
    if k in (5, 7):
        with ops.tuning(MI_GAUSS_DIRECT_WIN=0):
            old = timed(lambda: ops.gaussian_blur(fr, k, sg, out, direct=True))
        line += f"; round 2's kernel (MI_GAUSS_DIRECT_WIN=0): {old:.3f} ms"
    print(line, flush=True)
