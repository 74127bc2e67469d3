#!/usr/bin/env python3
"""The reference's demo workload as a profiling target: imageFFTImageGenerate on a 675x900 image
(the size of image/lord.jpeg), host buffers in and out.  Usage: python tools/run_fft_demo.py [reps] [w] [h]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from cudaimageprocessing_amd import ops, synth as S  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = int(sys.argv[2]) if len(sys.argv) > 2 else 675
h = int(sys.argv[3]) if len(sys.argv) > 3 else 900
img = S.synth_image(5, w, h)
out = np.empty_like(img)
ops.imageFFTImageGenerate(out, img, w, h)
t = time.perf_counter()
for _ in range(reps):
    ops.imageFFTImageGenerate(out, img, w, h)
dt = (time.perf_counter() - t) / reps
print(f"imageFFTImageGenerate {w}x{h}: {dt * 1e3:.3f} ms per call, {w * h / dt / 1e6:.0f} MP/s (host buffers, PCIe included)")
