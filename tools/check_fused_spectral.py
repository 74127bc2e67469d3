#!/usr/bin/env python3
"""imageFFTImageGenerate on 4096 x 4096 device images: the fused four-pass form (MI_FFT_FUSED_SPECTRAL=1) against the
step-by-step form, bytes and time.  Usage: python tools/check_fused_spectral.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cudaimageprocessing_amd import ops, synth as S  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 4096
import numpy as np  # noqa: E402
img = np.stack([S.pattern_image("natural", n, n, 3 + i) for i in range(batch)])
d = torch.from_numpy(img).cuda()
ws = torch.empty(ops.fft_highpass_workspace_bytes(n, n, batch), dtype=torch.uint8, device="cuda")
outs = {}
for mode in (0, 1):
    with ops.tuning(MI_FFT_FUSED_SPECTRAL=mode):
        out = torch.empty_like(d)
        ops.fft_highpass(d, 10.0, out, ws)
        torch.cuda.synchronize()
        outs[mode] = out.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ops.fft_highpass(d, 10.0, out, ws)
        a.record()
        for _ in range(10):
            ops.fft_highpass(d, 10.0, out, ws)
        b.record()
        b.synchronize()
        print(f"MI_FFT_FUSED_SPECTRAL={mode}: {a.elapsed_time(b) / 10 / batch:.3f} ms per 4096x4096 image (batch {batch})", flush=True)
diff = (outs[0].to(torch.int16) - outs[1].to(torch.int16)).abs()
print("max |fused - steps| =", int(diff.max()), " differing bytes:", int((diff != 0).sum()), "of", diff.numel(),
      " alpha all 255:", bool((outs[1][..., 3] == 255).all()))
