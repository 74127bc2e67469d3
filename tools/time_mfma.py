#!/usr/bin/env python3
"""ms per launch of the matrix-core Gaussian 15x15 and mean 17x17 on 16 x 3840x2160 (A/B target for tools/_var libraries)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cudaimageprocessing_amd import ops
W, H, B = 3840, 2160, 16
frames = ops.synth(2, W, H, batch=B)
out = torch.empty_like(frames)
def timed(fn, n=200):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(400): ops.fused_chain(frames, 1.0, out)
f = lambda: ops.gaussian_blur(frames, 15, 2.5, out)
m = lambda: ops.mean_blur(frames, 17, out)
print("gauss15 " + " ".join(f"{min(timed(f), timed(f)):.4f}" for _ in range(2)) + "   mean17 " + " ".join(f"{min(timed(m), timed(m)):.4f}" for _ in range(2)), flush=True)
