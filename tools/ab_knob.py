#!/usr/bin/env python3
"""A/B of one tuning knob on one filter, alternating on one box: ms per launch on 16 x 3840x2160.
Usage: python tools/ab_knob.py <mean|gauss> <k> <KNOB> <a> <b> [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cudaimageprocessing_amd import ops

kind, k, knob, a, b = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 6
frames = ops.synth(2, 3840, 2160, batch=16)
out = torch.empty_like(frames)
fn = (lambda: ops.mean_blur(frames, k, out)) if kind == "mean" else (lambda: ops.gaussian_blur(frames, k, 0.2 * k, out))
for _ in range(300):
    ops.fused_chain(frames, 1.0, out)


def timed(n=200):
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


res = {a: [], b: []}
for _ in range(rounds):
    for v in (a, b):
        with ops.tuning(**{knob: v}):
            res[v].append(timed())
for v in (a, b):
    print(f"{kind} {k} {knob}={v}: " + " ".join(f"{t:.4f}" for t in res[v]) + f"   median {sorted(res[v])[len(res[v]) // 2]:.4f} ms", flush=True)
