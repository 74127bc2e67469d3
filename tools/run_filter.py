#!/usr/bin/env python3
"""Runs one filter a few times on a batch of 3840x2160 frames (a profiling target).
Usage: python tools/run_filter.py <gray|sobel|mean3|median3|median5|gauss3|gauss5|gauss9|gauss15|chain> [reps] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cudaimageprocessing_amd import ops  # noqa: E402

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
frames = ops.synth(2, 3840, 2160, batch=B)
out = torch.empty_like(frames)
FILTERS = {
    "gray": lambda: ops.grayscale(frames, out),
    "sobel": lambda: ops.sobel(frames, out),
    "mean3": lambda: ops.mean_blur(frames, 3, out),
    "median3": lambda: ops.median_blur(frames, 3, out),
    "median5": lambda: ops.median_blur(frames, 5, out),
    "gauss3": lambda: ops.gaussian_blur(frames, 3, 7.0, out),
    "gauss5": lambda: ops.gaussian_blur(frames, 5, 1.0, out),
    "gauss9": lambda: ops.gaussian_blur(frames, 9, 2.0, out),
    "gauss15": lambda: ops.gaussian_blur(frames, 15, 2.5, out),
    "chain": lambda: ops.fused_chain(frames, 1.0, out),
}
import re  # noqa: E402
m = re.fullmatch(r"(mean|gauss|median)(\d+)", name)
if name not in FILTERS and m:     # any kernel size: mean11, gauss13, ...
    k = int(m.group(2))
    FILTERS[name] = {"mean": lambda: ops.mean_blur(frames, k, out),
                     "gauss": lambda: ops.gaussian_blur(frames, k, 2.0, out),
                     "median": lambda: ops.median_blur(frames, k, out)}[m.group(1)]
for _ in range(reps):
    FILTERS[name]()
torch.cuda.synchronize()
