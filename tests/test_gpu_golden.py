"""GPU parity against the committed vectors of tests/golden/vectors.npz (a crop of the reference's
sample image and a seeded synthetic image, with the CPU oracle's outputs; generating script:
tests/golden/make_golden.py).  Needs neither the oracle nor /root/reference at run time."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from cudaimageprocessing_amd import _lib, ops  # noqa: E402

VEC = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    _lib.lib()
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = np.asarray(a, np.complex128), np.asarray(b, np.complex128)
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))


@pytest.mark.parametrize("name", ["deer_crop", "synth_40x33"])
def test_filters_match_committed_vectors(dev, name):
    img = VEC[name]
    d = torch.from_numpy(img).to(dev)
    host = lambda t: t.cpu().numpy()
    assert np.array_equal(host(ops.grayscale(d)), VEC[f"{name}.grayscale"])
    assert np.array_equal(host(ops.sobel(d)), VEC[f"{name}.sobel"])
    assert np.array_equal(host(ops.mean_blur(d, 3)), VEC[f"{name}.mean3"])
    assert np.array_equal(host(ops.mean_blur(d, 7)), VEC[f"{name}.mean7"])
    assert np.array_equal(host(ops.median_blur(d, 3)), VEC[f"{name}.median3"])
    assert np.array_equal(host(ops.median_blur(d, 5)), VEC[f"{name}.median5"])
    for key, k, s in (("gauss3_s7", 3, 7.0), ("gauss5_s1", 5, 1.0), ("gauss15_s2p5", 15, 2.5)):
        want = VEC[f"{name}.{key}"]
        assert np.array_equal(host(ops.gaussian_blur(d, k, s, direct=True)), want)          # reference summation order
        sep = host(ops.gaussian_blur(d, k, s)).astype(np.int16)                             # separable: +-1 LSB
        assert np.abs(sep - want.astype(np.int16)).max() <= 1
    # config 2's chain with the Gaussian in the reference's order: bit-exact end to end
    c = ops.sobel(ops.median_blur(ops.gaussian_blur(ops.grayscale(d), 5, 1.0, direct=True), 3))
    assert np.array_equal(host(c), VEC[f"{name}.chain_c2"])


@pytest.mark.parametrize("name", ["deer_crop", "synth_40x33"])
def test_fft_path_matches_committed_vectors(dev, name):
    img = VEC[name]
    h, w = img.shape[:2]
    red = torch.from_numpy(img[..., 0].astype(np.complex64)).to(dev)
    assert rel_l2(ops.fft2d(red).cpu().numpy(), VEC[f"{name}.fft2d_red_f64"]) <= 1e-5
    out = np.empty_like(img)
    ops.imageFFTImageGenerate(out, img, w, h)
    assert np.abs(out.astype(np.int16) - VEC[f"{name}.fft_highpass_u8"].astype(np.int16)).max() <= 1
    ops.compressImage(out, img, w, h, 0.10)
    diff = np.abs(out.astype(np.int16) - VEC[f"{name}.compress_keep10"].astype(np.int16))
    assert diff.max() <= 3 and diff.mean() < 0.2


def test_gaussian_weights_match_committed_vectors(dev):
    import ctypes
    for k in (3, 5, 15):
        want = VEC[f"gaussian_kernel_{k}"]
        got = np.empty((k, k), np.float32)
        sigma = {3: 7.0, 5: 1.0, 15: 2.5}[k]
        _lib.check(_lib.lib().mi_gaussian_kernel(k, ctypes.c_float(sigma), got.ctypes.data))
        assert np.array_equal(got, want)


def test_fft_highpass_reproduces_the_upstream_sample_output(dev):
    """imageFFTImageGenerate on the reference's own demo input against the reference's own output
    (tests/golden/lord.jpeg -> tests/golden/upstream_output_of_lord.png, see tests/test_oracle.py): the
    675x900 image goes through the chirp-z path and through the reference's cufftPlan2d(width, height)
    geometry.  The input is the stb_image decode upstream used: fp32 transforms here and in cuFFT, so a few
    values that sit on a truncation boundary differ by one level; nothing differs by more."""
    Image = pytest.importorskip("PIL.Image")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    # lord.jpeg decoded by stb_image v2.30 as the reference decodes it (main.cpp:14): `demo/imgtool --decode
    # tests/golden/lord.jpeg x.png`, re-saved as an RGB PNG (a data fixture; Pillow's JPEG decoder differs from
    # stb's on 1.3 % of the bytes, which used to cap this comparison at 98.4 % identical)
    img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(here, "lord_decoded_by_stb.png")).convert("RGBA")))
    want = np.asarray(Image.open(os.path.join(here, "upstream_output_of_lord.png")))
    h, w = img.shape[:2]
    out = np.empty_like(img)
    ops.imageFFTImageGenerate(out, img, w, h)
    assert np.array_equal(out[..., 3], want[..., 3])
    d = np.abs(out[..., :3].astype(np.int16) - want[..., :3].astype(np.int16))
    print(f"identical {100 * (d == 0).mean():.2f} %, within 1: {100 * (d <= 1).mean():.2f} %, max {d.max()}")
    assert (d == 0).mean() >= 0.9999 and d.max() <= 1
    # in place, as the reference's main.cpp calls it (main.cpp:43)
    buf = img.copy()
    ops.imageFFTImageGenerate(buf, buf, w, h)
    assert np.array_equal(buf, out)
