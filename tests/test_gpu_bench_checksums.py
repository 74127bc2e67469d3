"""The checksums bench.py prints beside its timings, recomputed here through the same code at the same
workloads (BASELINE.json configs 2, 3 and 4 at one GPU) and held against the committed constants
(tests/golden/bench_checksums.json, made by tools/make_bench_checksums.py from a bench.py line) AND against
the CPU oracle on the frames / row bands the CRCs cover: a bench line whose checksums equal these constants
timed launches that wrote what the oracle writes."""
import json
import os
import sys
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402
from cudaimageprocessing_amd import _lib, ops  # noqa: E402


@pytest.fixture(scope="module")
def want():
    return json.load(open(os.path.join(HERE, "golden", "bench_checksums.json")))


@pytest.fixture(scope="module")
def timer():
    assert torch.cuda.is_available()
    _lib.lib()
    dev = torch.device("cuda:0")
    return bench.Timer(torch, None, dev, "nccl"), dev


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def oracle_chain_crc(golden, img, img_dev):
    gray = golden.grayscale(img, threads=8)
    assert np.array_equal(ops.grayscale(img_dev).cpu().numpy(), gray)
    blur_dev = ops.gaussian_blur(torch.from_numpy(gray).to(img_dev.device), 5, 1.0).cpu().numpy()
    assert np.abs(blur_dev.astype(np.int16) - golden.gaussian(gray, 5, 1.0, threads=8).astype(np.int16)).max() <= 1
    return crc(golden.sobel(golden.medianblur(blur_dev, 3, threads=8), threads=8))


def test_c2_headline_output(want, timer, golden):
    _, dev = timer
    frames = ops.synth(2, bench.W, bench.H, batch=16, device=dev)
    out = ops.fused_chain(frames, 1.0)
    assert bench.crc32_tensor(out) == want["c2"]["crc32"]
    assert bench.crc32_tensor(out[0]) == want["c2"]["crc32_frame0"]
    assert bench.byte_sum(torch, out) == want["c2"]["byte_sum"]
    # The same constant from the oracle, frame 0 of the synthetic stream.  The chain's Gaussian stage is the
    # separable one (+-1 against the reference's direct sum by contract), so the tie goes through that stage's
    # device output: oracle gray == device gray, device Gaussian within 1 of the oracle's, and from there the
    # oracle's median and Sobel give the constant bit for bit.
    assert oracle_chain_crc(golden, golden.synth(2, bench.W, bench.H), frames[0]) == want["c2"]["crc32_frame0"]


def test_c3_leg_output(want, timer, golden):
    t, dev = timer
    leg = bench.leg_c3(torch, ops, t, 0, 1, dev, 1)
    got = leg["checksum"]
    assert got["crc32_frames"] == want["c3"]["crc32_frames"]
    assert got["byte_sum"] == want["c3"]["byte_sum"]
    w, h = 1920, 1080
    for f, c in want["c3"]["crc32_frames"].items():
        img = golden.synth(1000, w, h, first_idx=int(f) * w * h)
        assert oracle_chain_crc(golden, img, torch.from_numpy(img).to(dev)) == c, f


def test_c4_leg_output(want, timer, golden):
    t, dev = timer
    leg = bench.leg_c4(torch, ops, t, 0, 1, dev, 1)
    w = 32768
    for name, v in leg["per_filter"].items():
        assert v["checksum"]["crc32_bands"] == want["c4"][name]["crc32_bands"], name
        assert v["checksum"]["byte_sum"] == want["c4"][name]["byte_sum"], name
    # the first band (rows 0..63) from the oracle: the top 80 rows of the image are all those rows depend on
    # (7 rows of taps + 1 for the flat-index wrap), and the bottom border of an 80-row image is below them
    top = golden.synth(4, w, 80)
    ref = golden.medianblur(top, 5, threads=8)[:64]
    assert crc(ref) == want["c4"]["median5"]["crc32_bands"]["0"]
    # Gaussian 15x15: separable fixed point here against the oracle's direct fp32 sum is +-1 by contract, so no
    # CRC can tie them; the device output of that band is compared value by value instead
    img = torch.from_numpy(top).to(dev)
    dev_out = ops.gaussian_blur(img, 15, 2.5)[:64].cpu().numpy()
    assert crc(dev_out) == want["c4"]["gaussian15"]["crc32_bands"]["0"]
    gref = golden.gaussian(top, 15, 2.5, threads=8)[:64]
    assert np.abs(dev_out.astype(np.int16) - gref.astype(np.int16)).max() <= 1
