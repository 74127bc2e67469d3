#!/usr/bin/env python3
"""compressImage (kernels/imageTools.h:17, defined here as: keep the largest DFT coefficients per channel):
PSNR against the original and wall time per call for a range of keep fractions, on the reference's two
sample images (decoded by Pillow).  SURVEY.md section 8(f) row 3 asks for this table.
Usage (GPU box, repo root): python tools/compress_psnr.py > gpurun_out/compress_psnr.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402
from cudaimageprocessing_amd import ops  # noqa: E402

print("| image | keep fraction | PSNR (dB, R,G,B together) | ms per call (host buffers in and out) |")
print("|---|---|---|---|")
for name in ("lord.jpeg", "deer.jpeg"):
    rgba = np.ascontiguousarray(np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", name)).convert("RGBA")))
    h, w = rgba.shape[:2]
    out = np.empty_like(rgba)
    for keep in (1.0, 0.5, 0.2, 0.1, 0.05, 0.02, 0.01, 0.005):
        for _ in range(3):
            ops.compressImage(out, rgba, w, h, keep)
        t0 = time.perf_counter()
        for _ in range(5):
            ops.compressImage(out, rgba, w, h, keep)
        ms = (time.perf_counter() - t0) / 5 * 1e3
        mse = float(np.mean((out[..., :3].astype(np.float64) - rgba[..., :3].astype(np.float64)) ** 2))
        psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        print(f"| {name} ({w}x{h}) | {keep:g} | {psnr:.2f} | {ms:.2f} |", flush=True)
