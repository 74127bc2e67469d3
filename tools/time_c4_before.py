#!/usr/bin/env python3
"""32768^2: the kernels that ran Gaussian 31x31 (separable mode), mean 31x31 and mean 15x15 before the matrix-core ones
(MI_GAUSS_MFMA_MAX=15 / MI_MEAN_MFMA=0), beside the defaults.  ms per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cudaimageprocessing_amd import ops

big = ops.synth(4, 32768, 32768)
out = torch.empty_like(big)


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for _ in range(50):
    ops.sobel(big, out)
print(f"gauss 31 default {timed(lambda: ops.gaussian_blur(big, 31, 5.0, out), 10):.3f} ms", flush=True)
with ops.tuning(MI_GAUSS_MFMA_MAX=15):
    print(f"gauss 31 direct  {timed(lambda: ops.gaussian_blur(big, 31, 5.0, out), 2):.3f} ms", flush=True)
print(f"mean 31 default  {timed(lambda: ops.mean_blur(big, 31, out), 10):.3f} ms", flush=True)
print(f"mean 15 default  {timed(lambda: ops.mean_blur(big, 15, out), 10):.3f} ms", flush=True)
with ops.tuning(MI_MEAN_MFMA=0):
    print(f"mean 31 prefix-sum march {timed(lambda: ops.mean_blur(big, 31, out), 10):.3f} ms", flush=True)
    print(f"mean 15 vector ALUs      {timed(lambda: ops.mean_blur(big, 15, out), 10):.3f} ms", flush=True)
