#!/usr/bin/env python3
"""Sweep of the fused chain kernel's tuning knobs (mi_debug_set_tuning; "-" = 0, the default) on the
bench workload: 16 x 3840x2160 (SWEEP_GEOMETRY=W,H,B for another).  Prints ms per launch, two passes per setting."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cudaimageprocessing_amd import ops

W, H, B = (int(v) for v in os.environ.get("SWEEP_GEOMETRY", "3840,2160,16").split(","))
frames = ops.synth(2, W, H, batch=B)
out = torch.empty_like(frames)

def timed(n=300):
    for _ in range(60):
        ops.fused_chain(frames, 1.0, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.fused_chain(frames, 1.0, out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for _ in range(400):
    ops.fused_chain(frames, 1.0, out)
torch.cuda.synchronize()
knobs = {}
for arg in sys.argv[1:]:
    k, v = arg.split("=")
    knobs[k] = v.split(",")
names = list(knobs)
for combo in itertools.product(*[knobs[k] for k in names]):
    for k, v in zip(names, combo):
        ops.set_tuning(k, 0 if v == "-" else int(v))
    t = [timed(), timed()]
    print(" ".join(f"{k}={v}" for k, v in zip(names, combo)), " ms %.4f %.4f" % tuple(t),
          " frac %.3f" % (B * W * H * 8 / (min(t) * 1e-3) / 8e12), flush=True)
