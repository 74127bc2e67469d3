"""Demo CLI, the Python counterpart of the reference's demo program (main.cpp:12-49) and of
demo/imgtool.cpp: decode an image on the host (Pillow: JPEG, PNG, GIF ... -- upstream used the
stb_image header), run a list of operations through libmi_imagetools, write a PNG.

  python -m cudaimageprocessing_amd.cli IN OUT.png [--ops gray,gauss:5:1.0,median:3,sobel] [--one-trip]
  python -m cudaimageprocessing_amd.cli --selftest

ops: gray | sobel | gauss:K:SIGMA | gauss_direct:K:SIGMA | mean[:K] | median[:K] | fft_highpass[:CUTOFF]
     | compress[:KEEP]
"""
import argparse
import sys
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(prog="cudaimageprocessing_amd.cli", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("input", nargs="?")
    ap.add_argument("output", nargs="?")
    ap.add_argument("--ops", default="fft_highpass", help="comma-separated operations (default: what upstream's main.cpp runs)")
    ap.add_argument("--one-trip", action="store_true", help="send runs of pixel filters through mi_run_ops_u8x4")
    ap.add_argument("--selftest", action="store_true")
    args = ap.parse_args(argv)

    from . import ops
    if args.selftest:
        ok = ops.test()
        print("self-test", "passed" if ok else "FAILED")
        return 0 if ok else 1
    if not args.input or not args.output:
        ap.error("need an input and an output file")
    from PIL import Image
    img = np.ascontiguousarray(np.asarray(Image.open(args.input).convert("RGBA")))
    h, w = img.shape[:2]
    loaded = np.empty_like(img)
    ops.imageLoadWrapper(img, loaded, w * h)
    print(f"loaded {args.input}: {w} x {h}")
    items = [s.strip() for s in args.ops.split(",") if s.strip()]
    t0 = time.perf_counter()
    i = 0
    while i < len(items):
        name, *par = items[i].split(":")
        if name == "fft_highpass":
            ops.imageFFTImageGenerate(loaded, loaded, w, h, float(par[0]) if par else 10.0)
            i += 1
        elif name == "compress":
            ops.compressImage(loaded, loaded, w, h, float(par[0]) if par else 0.10)
            i += 1
        else:
            j = i + 1
            while args.one_trip and j < len(items) and items[j].split(":")[0] not in ("fft_highpass", "compress"):
                j += 1
            ops.run_ops(loaded, loaded, w, h, ",".join(items[i:j]))
            i = j
    dt = time.perf_counter() - t0
    ops.imageWriteWrapper(args.output, loaded, w, h)
    print(f"{len(items)} operation(s) in {dt * 1e3:.1f} ms ({w * h / 1e6 / dt:.1f} MP/s incl. PCIe); wrote {args.output}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
