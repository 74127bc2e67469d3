#!/usr/bin/env python3
"""rows x cols planes (1024, 2048 or 4096 each): the two-pass plan (fft_twopass.h) against the three-pass plan
and numpy.fft, then the timing of both.  Usage: python tools/check_twopass.py [batch [rows cols]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cudaimageprocessing_amd import ops  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rows = int(sys.argv[2]) if len(sys.argv) > 3 else 4096
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
rng = np.random.default_rng(1)
x0 = (rng.uniform(0, 255, (rows, cols)) + 1j * rng.uniform(-50, 50, (rows, cols))).astype(np.complex64)
xd = torch.from_numpy(x0).cuda()


def rel(a, b):
    a = a.astype(np.complex128); b = b.astype(np.complex128)
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))


t0 = time.time()
ref = np.fft.fft2(x0.astype(np.complex128))
refi = np.fft.ifft2(x0.astype(np.complex128)) * (rows * cols)
print(f"{rows} x {cols}: numpy reference {time.time() - t0:.1f} s", flush=True)
for mode in (0, 1, 2):
    with ops.tuning(MI_FFT_TWOPASS=mode, restore={"MI_FFT_TWOPASS": 1}):
        f = ops.fft2d(xd).cpu().numpy()
        i = ops.fft2d(xd, inverse=True).cpu().numpy()
        y = xd.clone()
        ws = torch.empty(ops.fft2d_workspace_bytes(rows, cols), dtype=torch.uint8, device="cuda")
        ops.fft2d(y, out=y, workspace=ws)
        print(f"MI_FFT_TWOPASS={mode}: forward rel-L2 {rel(f, ref):.2e}, inverse {rel(i, refi):.2e}, in place {rel(y.cpu().numpy(), ref):.2e}", flush=True)

gray = ops.synth(5, cols, rows * batch)[..., 0].to(torch.float32).reshape(batch, rows, cols)
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


px = batch * rows * cols
for mode in (0, 1, 0, 1):
    with ops.tuning(MI_FFT_TWOPASS=mode, restore={"MI_FFT_TWOPASS": 1}):
        fwd = timed(lambda: ops.fft2d(x, out=f))
        inv = timed(lambda: ops.fft2d(f, inverse=True, out=y))
        rt = timed(lambda: (ops.fft2d(x, out=f), ops.fft2d(f, inverse=True, out=y)))
        print(f"MI_FFT_TWOPASS={mode} batch {batch}: forward {fwd:.3f} ms ({px * 32 / fwd / 1e6:.0f} GB/s), inverse {inv:.3f} ms, "
              f"round trip {rt:.3f} ms = {px * 64 / rt / 1e6:.0f} GB/s ({px * 64 / rt / 1e6 / 8000 * 100:.1f}% of 8 TB/s)", flush=True)
