#!/usr/bin/env python3
"""PCIe-inclusive rate of the host tier: the C2 chain on host images through mi_run_ops_u8x4
(pageable caller buffers, as the reference's wrappers take them).  Prints MP/s per image size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from cudaimageprocessing_amd import ops  # noqa: E402

SPEC = "gray,gauss:5:1.0,median:3,sobel"
for w, h in ((3840, 2160), (8192, 8192), (16384, 16384)):
    img = ops.synth(2, w, h).cpu().numpy()
    out = np.empty_like(img)
    # the first dozen calls on fresh host buffers run 30 % slower (the OS is still settling their pages:
    # one call takes a hiccup, the following ones are fast), so the steady state is measured after 14 calls
    for _ in range(14):
        ops.run_ops(out, img, w, h, SPEC)
    for chunk in (None, 64, 256, 1024, h):
        reps = 5
        ops.run_ops(out, img, w, h, SPEC, chunk_rows=chunk)
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.run_ops(out, img, w, h, SPEC, chunk_rows=chunk)
        dt = (time.perf_counter() - t0) / reps
        print(f"{w}x{h} chunk_rows={chunk}: {dt * 1e3:.2f} ms  {w * h / 1e6 / dt:.0f} MP/s  "
              f"({2 * w * h * 4 / dt / 1e9:.1f} GB/s over PCIe both ways)", flush=True)
