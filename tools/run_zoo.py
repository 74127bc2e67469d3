#!/usr/bin/env python3
"""Launches every filter kernel of the path a few times on 16 x 3840x2160 (a profiling target for
tools/valu_roofline.sh: one rocprofv3 --pmc pass sees them all)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cudaimageprocessing_amd import ops  # noqa: E402

frames = ops.synth(2, 3840, 2160, batch=16)
out = torch.empty_like(frames)


def prefix31():
    with ops.tuning(MI_MEAN_MFMA_MAX=17):      # mean_prefix_kernel, the default for 19..31 before the wide matrix-core kernel
        ops.mean_blur(frames, 31, out)


ZOO = [lambda: ops.grayscale(frames, out), lambda: ops.sobel(frames, out), lambda: ops.mean_blur(frames, 3, out),
       lambda: ops.median_blur(frames, 3, out), lambda: ops.median_blur(frames, 5, out),
       lambda: ops.gaussian_blur(frames, 3, 7.0, out), lambda: ops.gaussian_blur(frames, 5, 1.0, out),
       lambda: ops.gaussian_blur(frames, 9, 2.0, out), lambda: ops.gaussian_blur(frames, 15, 2.5, out),
       lambda: ops.mean_blur(frames, 7, out), lambda: ops.mean_blur(frames, 15, out), lambda: ops.mean_blur(frames, 17, out), lambda: ops.mean_blur(frames, 31, out), lambda: ops.gaussian_blur(frames, 31, 5.0, out), prefix31,
       lambda: ops.gaussian_blur(frames, 5, 1.0, out, direct=True), lambda: ops.gaussian_blur(frames, 9, 2.0, out, direct=True),
       lambda: ops.fused_chain(frames, 1.0, out)]
for _ in range(300):          # spin-up: the first ~50 ms after idle run at a lower clock
    ops.fused_chain(frames, 1.0, out)
for _ in range(8):
    for fn in ZOO:
        fn()
torch.cuda.synchronize()
