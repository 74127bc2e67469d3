#!/usr/bin/env python3
"""Randomised parity run for the matrix-core kernels: random widths (multiples of 4), heights, kernel sizes, sigmas,
batches and row-strip splits; Gaussian within one level of the oracle, mean bit-exact, strips == single launch.
Usage: python tools/fuzz_mfma.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cudaimageprocessing_amd import _lib, ops
from oracle import golden

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
bad = 0
for it in range(cases):
    mean = bool(rng.integers(0, 2))
    k = int(rng.choice([11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31] if mean else [7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31]))
    w = 4 * int(rng.integers(1, 180))
    h = int(rng.integers(3, 220))
    if k * k > w * h:
        continue
    b = int(rng.integers(1, 4))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        img = rng.integers(0, 256, (b, h, w, 4), dtype=np.uint8)
    elif kind == 1:
        img = np.full((b, h, w, 4), int(rng.integers(0, 256)), np.uint8)
    elif kind == 2:
        img = (rng.integers(0, 2, (b, h, w, 4)) * 255).astype(np.uint8)
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([((xx * 3 + yy * 5 + c * 40) % 256).astype(np.uint8) for c in range(4)], -1)[None].repeat(b, 0)
        img = np.ascontiguousarray(np.stack([np.roll(img[f], f, 0) for f in range(b)]))
    sigma = float(rng.choice([0.3, 0.8, 1.5, 2.5, 4.0, 9.0, 30.0]))
    d = torch.from_numpy(img).to(dev)
    rpb = int(rng.choice([0, 0, 16, 48, 200]))
    with ops.tuning(MI_ROWS_PER_BAND=rpb):
        got = (ops.mean_blur(d, k) if mean else ops.gaussian_blur(d, k, sigma)).cpu().numpy()
    for f in range(b):
        want = golden.meanblur(img[f], k, threads=8) if mean else golden.gaussian(img[f], k, sigma, threads=8)
        diff = np.abs(got[f].astype(np.int16) - want.astype(np.int16))
        if (mean and diff.max() != 0) or diff.max() > 1:
            bad += 1
            print(f"case {it}: {'mean' if mean else 'gauss'} k={k} sigma={sigma} {w}x{h} batch {b} frame {f} kind {kind} rpb {rpb}: max diff {diff.max()} at {np.argwhere(diff == diff.max())[0].tolist()}", flush=True)
    # a random split into row strips of frame 0
    filt = _lib.FILTER_MEAN if mean else _lib.FILTER_GAUSSIAN
    halo = ops.halo_rows(filt, k, w)
    cuts = sorted(set([0, h] + [int(x) for x in rng.integers(1, h, int(rng.integers(0, 4)))]))
    pieces = []
    for y0, y1 in zip(cuts[:-1], cuts[1:]):
        r0, r1 = max(0, y0 - halo), min(h, y1 + halo)
        out = torch.empty((y1 - y0, w, 4), dtype=torch.uint8, device=dev)
        ops.filter_strip(filt, d[0, r0:r1].contiguous(), out, w, h, r0, y0, y1, k, sigma)
        pieces.append(out.cpu().numpy())
    if not np.array_equal(np.concatenate(pieces), got[0]):
        bad += 1
        print(f"case {it}: strips {cuts} differ from the single launch ({'mean' if mean else 'gauss'} k={k} {w}x{h})", flush=True)
    if it % 25 == 0:
        print(f"... {it} cases, {bad} bad", flush=True)
print(f"{cases} cases, {bad} bad", flush=True)
sys.exit(1 if bad else 0)
