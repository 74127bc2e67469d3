#!/usr/bin/env python3
"""Times every filter kernel on a batch of 3840x2160 frames under different launch knobs
(set through mi_debug_set_tuning), interleaved in one process.
Usage: python tools/tune_filters.py [batch] ; prints GB/s (8 B/pixel algorithmic) per config."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from cudaimageprocessing_amd import ops  # noqa: E402

W, H = 3840, 2160
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ROUNDS = 3

frames = ops.synth(2, W, H, batch=B)
out = torch.empty_like(frames)
px = B * W * H

FILTERS = {
    "gray": lambda: ops.grayscale(frames, out),
    "sobel": lambda: ops.sobel(frames, out),
    "mean3": lambda: ops.mean_blur(frames, 3, out),
    "median3": lambda: ops.median_blur(frames, 3, out),
    "gauss5": lambda: ops.gaussian_blur(frames, 5, 1.0, out),
}
CHAIN = {"chain": lambda: ops.fused_chain(frames, 1.0, out)}
EXTRA = {
    "gauss3": lambda: ops.gaussian_blur(frames, 3, 7.0, out),
    "gauss9": lambda: ops.gaussian_blur(frames, 9, 2.0, out),
    "gauss15": lambda: ops.gaussian_blur(frames, 15, 2.5, out),
    "median5": lambda: ops.median_blur(frames, 5, out),
}


def timed(fn, reps=40):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def sweep(filters, configs):
    res = {(f, i): [] for f in filters for i in range(len(configs))}
    for _ in range(ROUNDS):
        for i, cfg in enumerate(configs):
            for k, v in cfg.items():
                ops.set_tuning(k, v)
            for f, fn in filters.items():
                res[(f, i)].append(timed(fn))
    print("config".ljust(44) + "".join(f"{f:>10s}" for f in filters))
    for i, cfg in enumerate(configs):
        label = " ".join(f"{k[3:]}={v}" for k, v in cfg.items())
        cells = []
        for f in filters:
            ms = sorted(res[(f, i)])[ROUNDS // 2]
            cells.append(f"{px * 8 / (ms * 1e-3) / 1e9:10.0f}")
        print(label.ljust(44) + "".join(cells), flush=True)


if __name__ == "__main__":
    for fn in list(FILTERS.values()) + list(EXTRA.values()):
        fn()
    timed(CHAIN["chain"], 400)     # spin the clocks up: the first ~50 ms of work run 20 % slower
    torch.cuda.synchronize()
    mode = sys.argv[2] if len(sys.argv) > 2 else "main"
    if mode == "chain":
        sweep(CHAIN, [dict(MI_CHAIN_ROWS_PER_BAND=r) for r in (16, 20, 24, 27, 30, 32, 36, 40, 45, 48, 54, 60, 72, 90, 108)])
    elif mode == "main":
        sweep(FILTERS, [dict(MI_ROWS_PER_BAND=r) for r in (4, 6, 8, 12, 16, 24, 32)])
    else:
        sweep(EXTRA, [dict(MI_ROWS_PER_BAND=r) for r in (8, 16, 32, 64)])
