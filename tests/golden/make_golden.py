#!/usr/bin/env python3
"""Writes tests/golden/vectors.npz: small inputs and the CPU oracle's outputs for every function
on the path, so that parity can be checked against committed data (the reference holds no
vectors of its own: SURVEY.md 8c).  Inputs: a 64x48 crop of the reference's sample image
image/deer.jpeg (committed beside this script, decoded with Pillow) and a 40x33 seeded synthetic
image whose width is not a multiple of 4.  Regenerate with:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import golden  # noqa: E402
from cudaimageprocessing_amd import synth as S  # noqa: E402


def outputs(img):
    out = {"grayscale": golden.grayscale(img), "sobel": golden.sobel(img), "mean3": golden.meanblur(img, 3),
           "mean7": golden.meanblur(img, 7), "median3": golden.medianblur(img, 3), "median5": golden.medianblur(img, 5),
           "gauss3_s7": golden.gaussian(img, 3, 7.0), "gauss5_s1": golden.gaussian(img, 5, 1.0),
           "gauss15_s2p5": golden.gaussian(img, 15, 2.5)}
    stages = golden.chain(img)
    out["chain_c2"] = stages[-1] if isinstance(stages, (list, tuple)) else stages
    u8, real = golden.fft_highpass_image(img, 10.0)
    out["fft_highpass_u8"] = u8
    out["fft_highpass_real"] = real
    out["compress_keep10"] = golden.compress_image(img, 0.10)
    h, w = img.shape[:2]
    plane = img[..., 0].astype(np.complex128)
    out["fft2d_red_f64"] = golden.fft2d_f64(plane, -1)
    return out


def main():
    from PIL import Image
    deer = np.ascontiguousarray(np.asarray(Image.open(os.path.join(HERE, "deer.jpeg")).convert("RGBA")))
    crop = np.ascontiguousarray(deer[1000:1048, 1300:1364])            # 48 rows x 64 columns
    synth = S.synth_image(7, 40, 33)
    data = {"deer_crop": crop, "synth_40x33": synth}
    for name, img in (("deer_crop", crop), ("synth_40x33", synth)):
        for k, v in outputs(img).items():
            data[f"{name}.{k}"] = v
    for k, s in ((3, 7.0), (5, 1.0), (15, 2.5)):
        data[f"gaussian_kernel_{k}"] = golden.gaussian_kernel(k, s)
    path = os.path.join(HERE, "vectors.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes,", len(data), "arrays")


if __name__ == "__main__":
    main()
