#!/usr/bin/env python3
"""Every device kernel on the path timed at BASELINE.json's sizes on one GPU,
against the 8 TB/s HBM roofline at its ALGORITHMIC bytes per pixel.
Prints a markdown table (committed as profiles/rNN_all_kernels.md).
  python tools/bench_all.py [--reps 50]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cudaimageprocessing_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=50)
args = ap.parse_args()
PEAK = 8000.0


def timed(fn, reps):
    """ms per call: three untimed calls, then the faster of two batches of `reps` (the clock of the part
    settles within the first batch after a change of kernel)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b) / reps
        best = ms if best is None else min(best, ms)
    return best


rows = []


def report(name, ref, px, bytes_per_px, fn, reps=None, note=""):
    ms = timed(fn, reps or args.reps)
    gbs = px * bytes_per_px / ms / 1e6
    rows.append(f"| {name} | {ref} | {px / 1e6:.0f} | {bytes_per_px:g} | {ms:.3f} | {px / ms / 1e6:.1f} | {gbs:.0f} | {100 * gbs / PEAK:.0f} % | {note} |")
    print(rows[-1], flush=True)


print("| kernel (workload) | reference | MP per launch | algorithmic B/px | ms | GP/s | GB/s | of 8 TB/s | note |")
print("|---|---|---|---|---|---|---|---|---|")

# ---- C2 / C3 shape: batch of 16 4K frames (531 MB per buffer: beyond the 256 MiB Infinity Cache)
W, H, B = 3840, 2160, 16
fr = ops.synth(2, W, H, batch=B)
out = torch.empty_like(fr)
px = B * W * H
timed(lambda: ops.fused_chain(fr, 1.0, out), 400)       # clocks up
C = "16 x 3840x2160"
report(f"grayscale ({C})", "imageTools.cu:217-231", px, 8, lambda: ops.grayscale(fr, out))
report(f"Sobel ({C})", ":255-285", px, 8, lambda: ops.sobel(fr, out))
report(f"mean 3x3 ({C})", ":313-340", px, 8, lambda: ops.mean_blur(fr, 3, out))
report(f"mean 7x7 ({C})", ":313-340", px, 8, lambda: ops.mean_blur(fr, 7, out), 20)
report(f"mean 11x11, matrix cores ({C})", ":313-340", px, 8, lambda: ops.mean_blur(fr, 11, out), 20, "sizes 11..17: gauss_mfma_kernel<1>, exact")
report(f"mean 15x15, matrix cores ({C})", ":313-340", px, 8, lambda: ops.mean_blur(fr, 15, out), 20, "sizes 11..17")
report(f"mean 17x17, matrix cores ({C})", ":313-340", px, 8, lambda: ops.mean_blur(fr, 17, out), 20, "sizes 11..17")
report(f"mean 19x19, matrix cores ({C})", ":313-340", px, 8, lambda: ops.mean_blur(fr, 19, out), 20, "sizes 19..31: gauss_mfma_wide_kernel<1>, exact")
report(f"mean 31x31, matrix cores ({C})", ":313-340", px, 8, lambda: ops.mean_blur(fr, 31, out), 20, "sizes 19..31")
ops.set_tuning("MI_MEAN_MFMA_MAX", 17)
report(f"mean 19x19, prefix-sum march (MI_MEAN_MFMA_MAX=17) ({C})", ":313-340", px, 8, lambda: ops.mean_blur(fr, 19, out), 20, "the default before the wide matrix-core kernel: no row window, the leaving row is read again")
report(f"mean 31x31, prefix-sum march (MI_MEAN_MFMA_MAX=17) ({C})", ":313-340", px, 8, lambda: ops.mean_blur(fr, 31, out), 20, "")
ops.set_tuning("MI_MEAN_PREFIX", 0)
report(f"mean 31x31, LDS tiles (round 1's kernel, MI_MEAN_PREFIX=0) ({C})", ":313-340", px, 8, lambda: ops.mean_blur(fr, 31, out), 10, "kept as a cross-check")
ops.set_tuning("MI_MEAN_PREFIX", 1)
ops.set_tuning("MI_MEAN_MFMA_MAX", 31)
report(f"median 3x3 ({C})", "README.md:3", px, 8, lambda: ops.median_blur(fr, 3, out))
report(f"median 5x5 ({C})", "README.md:3", px, 8, lambda: ops.median_blur(fr, 5, out), 20, "VALU-bound")
for k, s in ((3, 7.0), (5, 1.0), (9, 2.0), (15, 2.5), (17, 3.0), (23, 4.0), (31, 5.0)):
    report(f"Gaussian {k}x{k} sigma {s}, separable ({C})", ":287-311", px, 8, lambda: ops.gaussian_blur(fr, k, s, out), 20,
           "matrix cores, within one level" if k >= 7 else "")
report(f"Gaussian 31x31, direct, LDS tiles ({C})", ":287-311", px, 8, lambda: ops.gaussian_blur(fr, 31, 5.0, out, direct=True), 3, "VALU-bound: 2883 fma per pixel; what the separable mode ran before the wide matrix-core kernel")
for k, s in ((3, 7.0), (5, 1.0), (7, 1.5)):
    report(f"Gaussian {k}x{k}, direct = the reference's summation order, bit-exact ({C})", ":287-311", px, 8, lambda: ops.gaussian_blur(fr, k, s, out, direct=True), 10, "VALU-bound: k*k fma per pixel and channel; 5x5 and 7x7: csrc/gauss_direct.hip")
report(f"Gaussian 9x9, direct, LDS tiles ({C})", ":287-311", px, 8, lambda: ops.gaussian_blur(fr, 9, 2.0, out, direct=True), 10, "VALU-bound: 243 fma per pixel")
report(f"fused chain gray>gauss5>median3>Sobel ({C})", "C2 chain", px, 8, lambda: ops.fused_chain(fr, 1.0, out), None, "the four kernels move 32 B/px")
bufs = [torch.empty_like(fr) for _ in range(3)] + [out]
report(f"the same chain as four kernels ({C})", "C2 chain", px, 32, lambda: ops.filter_chain(fr, bufs))
report(f"image load/write repack ({C})", ":107-132", px, 8, lambda: ops.image_load(fr.view(-1), out.view(-1)))
del bufs

# ---- north_star's "8K RGB batches": 4 frames of 7680 x 4320 (the same 531 MB per buffer as 16 4K frames)
del fr, out
torch.cuda.empty_cache()
W8, H8, B8 = 7680, 4320, 4
fr = ops.synth(8, W8, H8, batch=B8)
out = torch.empty_like(fr)
px8 = B8 * W8 * H8
C8 = "4 x 7680x4320"
report(f"grayscale ({C8})", "imageTools.cu:217-231", px8, 8, lambda: ops.grayscale(fr, out))
report(f"Sobel ({C8})", ":255-285", px8, 8, lambda: ops.sobel(fr, out))
report(f"mean 3x3 ({C8})", ":313-340", px8, 8, lambda: ops.mean_blur(fr, 3, out))
report(f"median 3x3 ({C8})", "README.md:3", px8, 8, lambda: ops.median_blur(fr, 3, out))
report(f"median 5x5 ({C8})", "README.md:3", px8, 8, lambda: ops.median_blur(fr, 5, out), 20, "VALU-bound")
report(f"Gaussian 5x5 sigma 1, separable ({C8})", ":287-311", px8, 8, lambda: ops.gaussian_blur(fr, 5, 1.0, out), 20)
report(f"Gaussian 15x15 sigma 2.5, separable ({C8})", ":287-311", px8, 8, lambda: ops.gaussian_blur(fr, 15, 2.5, out), 20)
report(f"fused chain gray>gauss5>median3>Sobel ({C8})", "C2 chain", px8, 8, lambda: ops.fused_chain(fr, 1.0, out))

# ---- C4 shape: one 32768 x 32768 image (4 GiB in, 4 GiB out)
del fr, out
torch.cuda.empty_cache()
W4 = H4 = 32768
big = ops.synth(4, W4, H4)
bout = torch.empty_like(big)
px4 = W4 * H4
report("Gaussian 15x15 sigma 2.5 (32768x32768)", ":287-311", px4, 8, lambda: ops.gaussian_blur(big, 15, 2.5, bout), 5)
report("Gaussian 31x31 sigma 5, separable (32768x32768)", ":287-311", px4, 8, lambda: ops.gaussian_blur(big, 31, 5.0, bout), 5, "matrix cores (radius 9..15 kernel)")
report("mean 31x31 (32768x32768)", ":313-340", px4, 8, lambda: ops.mean_blur(big, 31, bout), 5, "matrix cores (radius 9..15 kernel), exact")
report("mean 15x15 (32768x32768)", ":313-340", px4, 8, lambda: ops.mean_blur(big, 15, bout), 5, "matrix cores, exact")
report("median 5x5 (32768x32768)", "README.md:3", px4, 8, lambda: ops.median_blur(big, 5, bout), 5)
report("Sobel (32768x32768)", ":255-285", px4, 8, lambda: ops.sobel(big, bout), 5)
report("grayscale (32768x32768)", ":217-231", px4, 8, lambda: ops.grayscale(big, bout), 5)
del big, bout
torch.cuda.empty_cache()

# ---- C5 shape: 16 planes of 4096 x 4096 complex fp32
n, nb = 4096, 16
gray = ops.synth(5, n, n * nb)[..., 0].to(torch.float32).reshape(nb, n, n)
x = torch.complex(gray, torch.zeros_like(gray))
f = torch.empty_like(x)
y = torch.empty_like(x)
del gray
pxf = nb * n * n
C5 = "16 x 4096x4096"
report(f"fft2d forward ({C5})", "cufftExecC2C :690-701", pxf, 32, lambda: ops.fft2d(x, out=f), 10)
report(f"fft2d inverse ({C5})", "cufftExecC2C :741-751", pxf, 32, lambda: ops.fft2d(f, inverse=True, out=y), 10)
report(f"fft2d forward + inverse ({C5})", "C5 round trip", pxf, 64, lambda: (ops.fft2d(x, out=f), ops.fft2d(f, inverse=True, out=y)), 10)
# the other sizes of the two-pass plan (same 2 GiB of planes), and the device tier of imageFFTImageGenerate /
# compressImage with the conversions fused into the passes (8 images per launch)
for (r, c, b) in ((2048, 2048, 64), (1024, 1024, 256), (1024, 4096, 64), (4096, 1024, 64)):
    xs, fs_, ys = x.view(b, r, c), f.view(b, r, c), y.view(b, r, c)
    report(f"fft2d forward + inverse ({b} x {c}x{r})", "cufftExecC2C", pxf, 64, lambda: (ops.fft2d(xs, out=fs_), ops.fft2d(fs_, inverse=True, out=ys)), 10)
for nn, bb in ((4096, 8), (2048, 32)):
    imgs = ops.synth(5, nn, nn * bb).reshape(bb, nn, nn, 4)
    wsh = torch.empty(ops.fft_highpass_workspace_bytes(nn, nn, bb), dtype=torch.uint8, device="cuda")
    outs = torch.empty_like(imgs)
    report(f"imageFFTImageGenerate, device tier, fused passes ({bb} x {nn}x{nn})", ":633-791", bb * nn * nn, 160, lambda: ops.fft_highpass(imgs, 10.0, outs, wsh), 10,
           "B/px: 4 + 3 x 8 written, 3 x 16 per cols pass x 2, 3 x 8 + 4 at the end")
    with ops.tuning(MI_FFT_FUSED_SPECTRAL=0):
        report(f"the same, step by step ({bb} x {nn}x{nn})", ":633-791", bb * nn * nn, 160, lambda: ops.fft_highpass(imgs, 10.0, outs, wsh), 10, "same algorithmic bytes, nine passes")
    del imgs, wsh, outs
one = x[0]
report("fftShift (4096x4096)", ":564-583", n * n, 16, lambda: ops.fftshift_(one), 20, "Infinity-Cache resident (working set < 256 MiB): NOT an HBM number")
report("highpassFilter (4096x4096)", ":530-549", n * n, 16, lambda: ops.highpass_(one, 10.0), 20, "Infinity-Cache resident (working set < 256 MiB): NOT an HBM number")
report("NormalizeAndZeroImaginary (4096x4096)", ":551-562", n * n, 16, lambda: ops.normalize_zero_imag_(one), 20, "Infinity-Cache resident (working set < 256 MiB): NOT an HBM number")
img = ops.synth(5, n, n)
report("uchar4ToCufftComplex (4096x4096)", ":506-515", n * n, 28, lambda: ops.u8_to_complex(img), 20, "allocates its 3 planes per call; Infinity-Cache resident (working set < 256 MiB): NOT an HBM number")
pl = ops.u8_to_complex(img)
report("ComplexRGBToUchar (4096x4096)", ":517-527", n * n, 28, lambda: ops.complex_to_u8(pl[0], pl[1], pl[2]), 20, "Infinity-Cache resident (working set < 256 MiB): NOT an HBM number")
fl = x[0].real.contiguous()
report("findMinMax (4096x4096)", ":416-454", n * n, 4, lambda: ops.minmax(fl), 20, "Infinity-Cache resident (working set < 256 MiB): NOT an HBM number")
mm = ops.minmax(fl)
report("floatToUchar4 normalise (4096x4096)", ":458-471", n * n, 8, lambda: ops.normalize_to_u8(fl, mm), 20, "Infinity-Cache resident (working set < 256 MiB): NOT an HBM number")
# non power of two: the reference's demo image size and the sample image size
for (r, c) in ((900, 675), (2000, 2592)):
    z = torch.complex(torch.rand(r, c, device="cuda"), torch.rand(r, c, device="cuda"))
    ws = torch.empty(ops.fft2d_workspace_bytes(r, c), dtype=torch.uint8, device="cuda")
    zo = torch.empty_like(z)
    report(f"fft2d forward, chirp-z ({c}x{r})", "cufftExecC2C", r * c, 32, lambda: ops.fft2d(z, out=zo, workspace=ws), 20, "small: launch-bound")

# ---- the one workload the reference's authors ran: imageFFTImageGenerate on a 675x900 image, host buffers
import time  # noqa: E402
import numpy as np  # noqa: E402
from cudaimageprocessing_amd import synth as S  # noqa: E402
for (w, h) in ((675, 900), (2592, 2000)):
    himg = S.synth_image(5, w, h)
    hout = np.empty_like(himg)
    ops.imageFFTImageGenerate(hout, himg, w, h)
    t0 = time.perf_counter()
    for _ in range(20):
        ops.imageFFTImageGenerate(hout, himg, w, h)
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"| imageFFTImageGenerate, host buffers in and out ({w}x{h}) | main.cpp:43, :633-791 | {w * h / 1e6:.1f} | — | {ms:.3f} | "
          f"{w * h / ms / 1e6:.2f} | — | — | wall clock per call, PCIe copies included |", flush=True)
