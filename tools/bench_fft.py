#!/usr/bin/env python3
"""BASELINE.json config 5: 2-D fp32 FFT + inverse FFT round trip on 4096x4096 planes, batched.
Prints ms per pass, MP/s and GB/s at the algorithmic 32 B/pixel per direction.
Usage: python tools/bench_fft.py [n] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cudaimageprocessing_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
gray = ops.synth(5, n, n * batch)[..., 0].to(torch.float32).reshape(batch, n, n)
x = torch.complex(gray, torch.zeros_like(gray))
f = torch.empty_like(x)
y = torch.empty_like(x)
del gray


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


px = batch * n * n
fwd = timed(lambda: ops.fft2d(x, out=f))
inv = timed(lambda: ops.fft2d(f, inverse=True, out=y))
rt = timed(lambda: (ops.fft2d(x, out=f), ops.fft2d(f, inverse=True, out=y)))
err = float(torch.linalg.vector_norm((y[0] / (n * n) - x[0]).to(torch.complex128)) / torch.linalg.vector_norm(x[0].to(torch.complex128)))
print(f"fft2d {n}x{n} batch {batch}: forward {fwd:.3f} ms ({px * 32 / fwd / 1e6:.0f} GB/s), inverse {inv:.3f} ms, "
      f"round trip {rt:.3f} ms = {px / rt / 1e3:.0f} MP/s = {px * 64 / rt / 1e6:.0f} GB/s "
      f"({px * 64 / rt / 1e6 / 8000 * 100:.1f}% of 8 TB/s); round-trip rel-L2 error {err:.2e}")
