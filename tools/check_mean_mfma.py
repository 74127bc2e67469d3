#!/usr/bin/env python3
"""Mean blur on the matrix cores (gauss_mfma_kernel<1>, MI_MEAN_MFMA) against the oracle (bit-exact) and against the
vector-ALU kernels (MI_MEAN_MFMA=0), and ms per launch on 16 x 3840x2160.  Usage: python tools/check_mean_mfma.py"""
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
with ops.tuning(MI_MEAN_MFMA=5):
    for (w, h) in ((516, 75), (128, 64), (64, 48), (260, 130), (1920, 270), (8, 40), (20, 33)):
        for k in (5, 7, 9, 11, 13, 15, 17):
            if k * k > w * h:
                continue
            for kind in ("random", "constant255"):
                img = S.synth_image(k + w, w, h) if kind == "random" else np.full((h, w, 4), 255, np.uint8)
                want = golden.meanblur(img, k)
                got = ops.mean_blur(torch.from_numpy(img).to(dev), k).cpu().numpy()
                if not np.array_equal(got, want):
                    ok = False
                    bad = np.argwhere(got != want)
                    print(f"{w}x{h} k={k} {kind}: {len(bad)} bytes differ, first {bad[:3].tolist()} got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}", flush=True)
print("parity", "ok" if ok else "FAILED", flush=True)

W, H, B = 3840, 2160, 16
frames = ops.synth(2, W, H, batch=B)
out = torch.empty_like(frames)
for _ in range(300):
    ops.fused_chain(frames, 1.0, out)
for k in (5, 7, 9, 11, 13, 15, 17):
    cells = []
    for knob in (5, 0):
        with ops.tuning(MI_MEAN_MFMA=knob):
            t = min(timed(lambda: ops.mean_blur(frames, k, out)) for _ in range(3))
        cells.append(f"{'mfma' if knob else 'valu'} {t:.4f} ms ({B * W * H * 8 / (t * 1e-3) / 8e12:.3f})")
    print(f"16 x 4K mean {k}: " + "  ".join(cells), flush=True)
