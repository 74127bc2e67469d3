"""GPU parity: the HIP path (through the C ABI in libmi_imagetools.so) against the CPU oracle on
the same seeded inputs.  Bar: bit-exact for grayscale, Sobel, mean, median and the direct
Gaussian; max |delta| <= 1 LSB for the separable Gaussian (the reference truncates float->u8,
kernels/imageTools.cu:306-308, so a different summation order flips LSBs; BASELINE.json)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from cudaimageprocessing_amd import _lib, ops, synth as S  # noqa: E402

GAUSS_TOL = 1   # LSB, separable mode


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.lib()
    return torch.device("cuda:0")


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def host(t):
    return t.cpu().numpy()


SIZES = [(64, 48), (67, 45), (3, 3), (4, 3), (5, 7), (256, 16), (250, 33), (1000, 37), (1021, 40), (516, 130)]
PATTERNS = ["random", "natural", "binary", "checker", "gradient", "impulse", "constant"]


def make(kind, w, h, seed):
    return S.synth_image(seed, w, h) if kind == "random" else S.pattern_image(kind, w, h, seed)


# ------------------------------------------------------------------ generator
def test_device_synth_matches_host(dev, golden):
    d = host(ops.synth(5, 131, 77))
    assert np.array_equal(d, golden.synth(5, 131, 77))
    b = host(ops.synth(1000, 64, 32, batch=3))
    assert np.array_equal(b.reshape(-1, 4), golden.synth(1000, 64, 96).reshape(-1, 4))


# ------------------------------------------------------------------ grayscale
def test_grayscale_all_rgb_triples(dev, golden):
    """Every one of the 2^24 (R,G,B) inputs, as a 4096x4096 image."""
    idx = np.arange(1 << 24, dtype=np.uint32)
    img = np.empty((1 << 24, 4), np.uint8)
    img[:, 0] = idx >> 16
    img[:, 1] = (idx >> 8) & 255
    img[:, 2] = idx & 255
    img[:, 3] = (idx * 7) & 255
    out = host(ops.grayscale(to_dev(img.reshape(4096, 4096, 4), dev))).reshape(-1, 4)
    table = golden.grayscale_table(True)
    assert np.array_equal(out[:, 0], table)
    assert np.array_equal(out[:, 1], table) and np.array_equal(out[:, 2], table)
    assert np.array_equal(out[:, 3], img[:, 3])          # alpha passes through


@pytest.mark.parametrize("w,h", SIZES + [(1, 1), (2, 5), (1023, 3)])
def test_grayscale_sizes(dev, golden, w, h):
    img = S.synth_image(w * 31 + h, w, h)
    img[..., 3] = img[..., 1]
    assert np.array_equal(host(ops.grayscale(to_dev(img, dev))), golden.grayscale(img))


# ------------------------------------------------------------------ Sobel
def test_sobel_magnitude_every_reachable_sum(dev):
    n = 2 * 1020 * 1020 + 1
    got = host(ops.sobel_mag_table(n))
    s = np.arange(n, dtype=np.int64)
    expect = np.minimum(255, np.floor(np.sqrt(s.astype(np.float64)))).astype(np.int64)
    # float64 sqrt of an integer < 2^53 floors correctly except possibly at perfect squares: fix up
    expect = np.where((expect + 1) ** 2 <= s, expect + 1, expect)
    expect = np.where(expect ** 2 > s, expect - 1, expect)
    expect = np.minimum(255, expect)
    assert np.array_equal(got.astype(np.int64), expect)


@pytest.mark.parametrize("variant", [0])
def test_sobel_magnitude_of_every_gradient_pair_in_the_chain_form(dev, variant):
    # the fused chain evaluates the magnitude from (Gx, Gy) as f16 pairs: all 2041 x 2041 pairs
    g = np.arange(-1020, 1021, dtype=np.int64)
    s = g[None, :] ** 2 + g[:, None] ** 2
    expect = np.floor(np.sqrt(s.astype(np.float64))).astype(np.int64)
    expect = np.where((expect + 1) ** 2 <= s, expect + 1, expect)
    expect = np.where(expect ** 2 > s, expect - 1, expect)
    expect = np.minimum(255, expect)
    got = host(ops.sobel_mag_pairs(1020, variant)).astype(np.int64)
    assert np.array_equal(got, expect)


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("kind", ["random", "natural", "binary", "gradient"])
def test_sobel(dev, golden, w, h, kind):
    img = make(kind, w, h, 3)
    assert np.array_equal(host(ops.sobel(to_dev(img, dev))), golden.sobel(img))


# ------------------------------------------------------------------ mean
@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("kind", ["random", "constant", "binary"])
def test_mean3(dev, golden, w, h, kind):
    img = make(kind, w, h, 4)
    assert np.array_equal(host(ops.mean_blur(to_dev(img, dev), 3)), golden.meanblur(img, 3))


@pytest.mark.parametrize("k", [1, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31])
def test_mean_other_sizes(dev, golden, k):
    for (w, h) in ((67, 45), (256, 16), (5, 7), (300, 70), (512, 33)):
        img = S.synth_image(k, w, h)
        assert np.array_equal(host(ops.mean_blur(to_dev(img, dev), k)), golden.meanblur(img, k))
    white = np.full((40, 300, 4), 255, np.uint8)       # the largest sums the packed 16-bit lanes must hold
    assert np.array_equal(host(ops.mean_blur(to_dev(white, dev), k)), golden.meanblur(white, k))


@pytest.mark.parametrize("k", [5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31])
def test_mean_on_the_matrix_cores_is_exact(dev, golden, k):
    """gauss_mfma_kernel<1> and, from 19x19, gauss_mfma_wide_kernel<1> (the default for 11x11 .. 31x31; MI_MEAN_MFMA=5
    sends every size they take through them):
    integer row sums on the matrix cores, two exact f16 parts into the column pass, the division as one fused
    multiply-add -- bit-exact against the oracle on noise and on an all-white image (the largest sums), equal to the
    vector-ALU kernels, and independent of band height and row strips."""
    with ops.tuning(MI_MEAN_MFMA=5):
        for (w, h) in ((516, 75), (128, 64), (64, 48), (260, 131), (8, 40), (20, 33), (8, 140), (132, 17)):
            if k * k > w * h:
                continue
            for img in (S.synth_image(k + w, w, h), np.full((h, w, 4), 255, np.uint8)):
                d = to_dev(img, dev)
                got = ops.mean_blur(d, k)
                assert np.array_equal(host(got), golden.meanblur(img, k, threads=8)), (k, w, h)
                with ops.tuning(MI_MEAN_MFMA=0):
                    assert torch.equal(ops.mean_blur(d, k), got), (k, w, h)
        w, h = 1000, 203
        img = S.synth_image(3 * k, w, h)
        d = to_dev(img, dev)
        ref = ops.mean_blur(d, k)
        assert np.array_equal(host(ref), golden.meanblur(img, k, threads=8))
        for rpb in (1, 40, 512):
            with ops.tuning(MI_ROWS_PER_BAND=rpb):
                assert torch.equal(ops.mean_blur(d, k), ref), (k, rpb)
                for strip in (64, 256):      # every band downwards; 64- and 256-pixel strips: the same bytes
                    with ops.tuning(MI_MFMA_SERPENTINE=0, MI_GM_STRIP=strip):
                        assert torch.equal(ops.mean_blur(d, k), ref), (k, rpb, strip)
        halo = ops.halo_rows(_lib.FILTER_MEAN, k)
        edges = (0, 1, 9, 33, 100, 101, 203)
        pieces = []
        for y0, y1 in zip(edges[:-1], edges[1:]):
            r0, r1 = max(0, y0 - halo), min(h, y1 + halo)
            out = torch.empty((y1 - y0, w, 4), dtype=torch.uint8, device=dev)
            ops.filter_strip(_lib.FILTER_MEAN, d[r0:r1].contiguous(), out, w, h, r0, y0, y1, k, 0.0)
            pieces.append(out)
        assert torch.equal(torch.cat(pieces), ref), k


# ------------------------------------------------------------------ median
@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("kind", PATTERNS)
def test_median3(dev, golden, w, h, kind):
    img = make(kind, w, h, 5)
    assert np.array_equal(host(ops.median_blur(to_dev(img, dev), 3)), golden.medianblur(img, 3))


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48), (5, 7), (3, 3), (250, 33)])
@pytest.mark.parametrize("kind", ["random", "binary", "natural", "checker"])
def test_median5(dev, golden, w, h, kind):
    img = make(kind, w, h, 6)
    assert np.array_equal(host(ops.median_blur(to_dev(img, dev), 5)), golden.medianblur(img, 5))


# ------------------------------------------------------------------ Gaussian
@pytest.mark.parametrize("k,sigma", [(3, 7.0), (5, 1.0), (7, 1.5), (9, 2.0), (11, 2.0), (13, 3.0), (15, 2.5)])
def test_gaussian_separable_within_1_lsb(dev, golden, k, sigma):
    worst, flips, total = 0, 0, 0
    for (w, h) in SIZES:
        for kind in ("random", "natural", "constant"):
            img = make(kind, w, h, k)
            got = host(ops.gaussian_blur(to_dev(img, dev), k, sigma)).astype(np.int16)
            ref = golden.gaussian(img, k, sigma).astype(np.int16)
            d = np.abs(got - ref)
            worst = max(worst, int(d.max()))
            flips += int((d != 0).sum())
            total += d.size
            assert np.array_equal(got[..., 3], ref[..., 3])
            # border is exactly zero in both
            assert (got[0] == 0).all() and (got[:, 0] == 0).all() and (got[-1] == 0).all() and (got[:, -1] == 0).all()
    print(f"gaussian k={k} sigma={sigma}: max|delta|={worst}, {flips}/{total} bytes differ")
    assert worst <= GAUSS_TOL


@pytest.mark.parametrize("k,sigma", [(1, 0.5), (3, 7.0), (5, 1.0), (7, 1.5), (9, 2.0), (11, 2.0), (13, 3.0), (15, 2.5)])
def test_gaussian_direct_is_bit_exact(dev, golden, k, sigma):
    for (w, h) in ((67, 45), (64, 48), (5, 7), (256, 16), (130, 35)):
        img = S.synth_image(100 + k, w, h)
        got = host(ops.gaussian_blur(to_dev(img, dev), k, sigma, direct=True))
        assert np.array_equal(got, golden.gaussian(img, k, sigma))


@pytest.mark.parametrize("k,sigma", [(5, 1.0), (5, 0.3), (5, 40.0), (7, 1.5), (7, 0.4)])
def test_gaussian_direct_5_and_7_packed_pairs_and_round_2_kernel_agree_with_the_oracle(dev, golden, k, sigma):
    """5x5 and 7x7 come from csrc/gauss_direct.hip: 5x5 with its window as float pairs ((red, green) of a pixel,
    blue of pixels two columns apart) feeding v_pk_fma_f32, 7x7 with per-step conversions and vector-register
    weights.  Each half of a packed multiply-add is the scalar operation, so the bytes are the oracle's; round 2's
    kernel (MI_GAUSS_DIRECT_WIN=0) stays as a second implementation of the same sums.  Aligned (buffer addressing)
    and odd widths, bands that end mid-window, a batch, the structured patterns."""
    assert ops.get_tuning("MI_GAUSS_DIRECT_WIN") == 1
    for (w, h, pattern) in ((640, 61, None), (517, 73, None), (259, 40, "checker"), (1024, 33, "natural"), (8, 9, None), (300, 7, "impulse")):
        img = S.synth_image(7 * k + w, w, h) if pattern is None else S.pattern_image(pattern, w, h, k)
        want = golden.gaussian(img, k, sigma, threads=8)
        d = to_dev(img, dev)
        assert np.array_equal(host(ops.gaussian_blur(d, k, sigma, direct=True)), want), (w, h, "new")
        with ops.tuning(MI_GAUSS_DIRECT_WIN=0):
            assert np.array_equal(host(ops.gaussian_blur(d, k, sigma, direct=True)), want), (w, h, "round 2")
        with ops.tuning(MI_ROWS_PER_BAND=k + 1):
            assert np.array_equal(host(ops.gaussian_blur(d, k, sigma, direct=True)), want), (w, h, "short bands")
    frames = np.stack([S.synth_image(300 + f, 516, 50) for f in range(3)])
    got = host(ops.gaussian_blur(to_dev(frames, dev), k, sigma, direct=True))
    for f in range(3):
        assert np.array_equal(got[f], golden.gaussian(frames[f], k, sigma, threads=8))


# ------------------------------------------------------------------ batches, strips, chain
def test_batch_equals_per_frame(dev, golden):
    w, h, b = 130, 37, 5
    frames = np.stack([S.synth_image(1000 + f, w, h) for f in range(b)])
    d = to_dev(frames, dev)
    for name, fn, ref in (
        ("gray", lambda t: ops.grayscale(t), golden.grayscale),
        ("sobel", lambda t: ops.sobel(t), golden.sobel),
        ("mean", lambda t: ops.mean_blur(t, 3), lambda x: golden.meanblur(x, 3)),
        ("median3", lambda t: ops.median_blur(t, 3), lambda x: golden.medianblur(x, 3)),
        ("median5", lambda t: ops.median_blur(t, 5), lambda x: golden.medianblur(x, 5)),
        ("gauss_direct", lambda t: ops.gaussian_blur(t, 5, 1.0, direct=True), lambda x: golden.gaussian(x, 5, 1.0)),
        ("mean15", lambda t: ops.mean_blur(t, 15), lambda x: golden.meanblur(x, 15)),
        ("mean19_tiles", lambda t: ops.mean_blur(t, 19), lambda x: golden.meanblur(x, 19)),
        ("gauss_direct11_tiles", lambda t: ops.gaussian_blur(t, 11, 2.0, direct=True), lambda x: golden.gaussian(x, 11, 2.0)),
    ):
        out = host(fn(d))
        for f in range(b):
            assert np.array_equal(out[f], ref(frames[f])), (name, f)
    g = host(ops.gaussian_blur(d, 5, 1.0)).astype(np.int16)
    for f in range(b):
        assert np.abs(g[f] - golden.gaussian(frames[f], 5, 1.0).astype(np.int16)).max() <= GAUSS_TOL


STRIP_FILTERS = [
    (_lib.FILTER_GRAYSCALE, 0, 0.0), (_lib.FILTER_SOBEL, 3, 0.0), (_lib.FILTER_MEAN, 3, 0.0),
    (_lib.FILTER_MEDIAN, 3, 0.0), (_lib.FILTER_MEDIAN, 5, 0.0), (_lib.FILTER_GAUSSIAN, 5, 1.0),
    (_lib.FILTER_GAUSSIAN, 15, 2.5), (_lib.FILTER_GAUSSIAN_DIRECT, 5, 1.0), (_lib.FILTER_MEAN, 7, 0.0),
    (_lib.FILTER_MEAN, 15, 0.0), (_lib.FILTER_MEAN, 17, 0.0), (_lib.FILTER_MEAN, 31, 0.0),
    (_lib.FILTER_GAUSSIAN_DIRECT, 9, 2.0),
]


def run_full(filt, img_d, k, sigma):
    if filt == _lib.FILTER_GRAYSCALE:
        return ops.grayscale(img_d)
    if filt == _lib.FILTER_SOBEL:
        return ops.sobel(img_d)
    if filt == _lib.FILTER_MEAN:
        return ops.mean_blur(img_d, k)
    if filt == _lib.FILTER_MEDIAN:
        return ops.median_blur(img_d, k)
    return ops.gaussian_blur(img_d, k, sigma, direct=(filt == _lib.FILTER_GAUSSIAN_DIRECT))


@pytest.mark.parametrize("filt,k,sigma", STRIP_FILTERS)
@pytest.mark.parametrize("w,h,parts", [(64, 64, 4), (67, 45, 3), (256, 40, 2)])
def test_strips_stitch_to_full_image(dev, filt, k, sigma, w, h, parts):
    """Row strips with halo (the multi-GPU decomposition) are bit-identical to one full launch."""
    img = S.synth_image(40 + k, w, h)
    img_d = to_dev(img, dev)
    full = host(run_full(filt, img_d, k, sigma))
    halo = ops.halo_rows(filt, k)
    edges = [h * p // parts for p in range(parts + 1)]
    pieces = []
    for p in range(parts):
        y0, y1 = edges[p], edges[p + 1]
        r0, r1 = max(0, y0 - halo), min(h, y1 + halo)
        strip = img_d[r0:r1].contiguous()
        out = torch.empty((y1 - y0, w, 4), dtype=torch.uint8, device=dev)
        ops.filter_strip(filt, strip, out, w, h, r0, y0, y1, k, sigma)
        pieces.append(host(out))
    assert np.array_equal(np.concatenate(pieces), full)


def test_chain_4k_frame_against_oracle(dev, golden):
    """BASELINE.json config 2 at full size: gray -> Gaussian 5x5 -> median 3x3 -> Sobel."""
    w, h = 3840, 2160
    img_d = ops.synth(2, w, h)
    img = host(img_d)
    assert np.array_equal(img, S.synth_image(2, w, h))
    a, b, c, d = [host(t) for t in ops.filter_chain(img_d)]
    ga = golden.grayscale(img, threads=8)
    assert np.array_equal(a, ga)
    # stage by stage on the device's own previous stage, so a +-1 in the Gaussian does not leak
    gb = golden.gaussian(a, 5, 1.0, threads=8)
    assert np.abs(b.astype(np.int16) - gb.astype(np.int16)).max() <= GAUSS_TOL
    assert np.array_equal(c, golden.medianblur(b, 3, threads=8))
    assert np.array_equal(d, golden.sobel(c, threads=8))


@pytest.mark.parametrize("k", [33, 41, 49, 51, 63, 101])
def test_kernel_sizes_above_31_against_oracle(dev, golden, k):
    """The reference takes any odd kernalSize (imageTools.cu:313-340, :478-502; its 1-pixel guard and flat-index
    taps whatever the size).  Above 31 the mean runs on the LDS-tile kernel (to 49) and then one lane per pixel,
    the Gaussian one lane per pixel with its weights in device memory: bit-exact against the oracle, whole frames,
    a batch, and row strips; also through the host tier."""
    for (w, h) in ((150, 97), (64, 40)):
        img = make("random", w, h, k)
        d = to_dev(img, dev)
        assert np.array_equal(host(ops.mean_blur(d, k)), golden.meanblur(img, k)), (w, h)
        if k <= 41:
            assert np.array_equal(host(ops.gaussian_blur(d, k, 0.3 * k)), golden.gaussian(img, k, 0.3 * k)), (w, h)
    w, h = 150, 97
    img = make("natural", w, h, 3)
    d = to_dev(img, dev)
    batch = torch.stack([d, d.flip(0)]).contiguous()
    mb = ops.mean_blur(batch, k)
    assert torch.equal(mb[0], ops.mean_blur(d, k)) and torch.equal(mb[1], ops.mean_blur(batch[1].contiguous(), k))
    # rows [30, 60) as a strip with its halo
    halo = k // 2 + 1
    r0, r1 = max(0, 30 - halo), min(h, 60 + halo)
    out = torch.empty((30, w, 4), dtype=torch.uint8, device=dev)
    ops.filter_strip(_lib.FILTER_MEAN, d[r0:r1].contiguous(), out, w, h, r0, 30, 60, k, 0.0)
    assert torch.equal(out, ops.mean_blur(d, k)[30:60])
    if k == 33:
        got = np.empty_like(img)
        ops.run_ops(got, img, w, h, ops.parse_ops(f"mean:{k},gauss:{k}:9.5"))
        assert np.array_equal(got, golden.gaussian(golden.meanblur(img, k), k, 9.5))


def test_8k_frame_every_filter_against_oracle(dev, golden):
    """north_star's target workload, "8K RGB batches": one 7680 x 4320 frame through every filter of the path
    against the oracle (bit-exact; +-1 for the separable Gaussians), the fused chain against the four kernels,
    and a batch of four such frames against the single-frame launches."""
    w, h = 7680, 4320
    img_d = ops.synth(8, w, h)
    img = host(img_d)
    a, b, c, d = [host(t) for t in ops.filter_chain(img_d)]
    assert np.array_equal(a, golden.grayscale(img, threads=8))
    assert np.abs(b.astype(np.int16) - golden.gaussian(a, 5, 1.0, threads=8).astype(np.int16)).max() <= GAUSS_TOL
    assert np.array_equal(c, golden.medianblur(b, 3, threads=8))
    assert np.array_equal(d, golden.sobel(c, threads=8))
    assert torch.equal(ops.fused_chain(img_d), ops.filter_chain(img_d)[3])
    assert np.array_equal(host(ops.mean_blur(img_d, 3)), golden.meanblur(img, 3, threads=8))
    # the expensive oracle filters on the top 520 rows only: output rows 0..499 of the full frame depend on
    # nothing below row 508 (radius 7 + 1 for the flat-index wrap), and the crop's bottom border lies below them
    top = img[:520]
    assert np.array_equal(host(ops.median_blur(img_d, 5))[:500], golden.medianblur(top, 5, threads=8)[:500])
    assert np.array_equal(host(ops.gaussian_blur(img_d, 5, 1.0, direct=True))[:500], golden.gaussian(top, 5, 1.0, threads=8)[:500])
    g15 = host(ops.gaussian_blur(img_d, 15, 2.5))[:500]
    assert np.abs(g15.astype(np.int16) - golden.gaussian(top, 15, 2.5, threads=8)[:500].astype(np.int16)).max() <= GAUSS_TOL
    batch = torch.stack([img_d, img_d.flip(0), img_d.flip(1), img_d]).contiguous()
    for fn in (ops.sobel, lambda t: ops.median_blur(t, 3), lambda t: ops.gaussian_blur(t, 5, 1.0), lambda t: ops.fused_chain(t)):
        out = fn(batch)
        for i in range(4):
            assert torch.equal(out[i], fn(batch[i].contiguous())), i


@pytest.mark.parametrize("w,h", SIZES + [(3840, 270), (1920, 1080), (249, 64), (247, 9), (496, 12)])
def test_fused_chain_equals_four_kernels(dev, golden, w, h):
    """mi_chain_gray_gauss5_median3_sobel_u8x4_dev is bit-identical to the four separate kernels,
    and (stage by stage, see test_chain_4k_frame_against_oracle) to the oracle chain."""
    for kind in ("random", "natural", "binary"):
        img_d = to_dev(make(kind, w, h, 11), dev)
        ref = ops.filter_chain(img_d)[3]
        got = ops.fused_chain(img_d)
        assert torch.equal(got, ref), (kind, int((got != ref).sum()))
    # against the oracle directly on inputs where the separable Gaussian has no LSB flip
    img = make("natural", w, h, 12)
    a = golden.grayscale(img)
    b_dev = host(ops.gaussian_blur(to_dev(a, dev), 5, 1.0))
    if np.array_equal(b_dev, golden.gaussian(a, 5, 1.0)):
        expect = golden.sobel(golden.medianblur(b_dev, 3))
        assert np.array_equal(host(ops.fused_chain(to_dev(img, dev))), expect)


def test_fused_chain_batch(dev):
    frames = ops.synth(1000, 640, 200, batch=5)
    fused = ops.fused_chain(frames)
    assert torch.equal(fused, ops.filter_chain(frames)[3])
    for f in range(5):
        assert torch.equal(fused[f], ops.fused_chain(frames[f].contiguous()))


def test_decomposition_does_not_change_results(dev):
    """Band height (and with it the marching direction of every row) is a launch detail: every
    filter, the separable Gaussian included, gives bit-identical output for any choice."""
    img = ops.synth(77, 1920, 270)
    fns = {"sobel": ops.sobel, "mean": lambda t: ops.mean_blur(t, 3), "median": lambda t: ops.median_blur(t, 3),
           "gauss5": lambda t: ops.gaussian_blur(t, 5, 1.0), "gauss15": lambda t: ops.gaussian_blur(t, 15, 2.5),
           "median5": lambda t: ops.median_blur(t, 5), "mean7": lambda t: ops.mean_blur(t, 7),
           "mean15": lambda t: ops.mean_blur(t, 15), "gauss11": lambda t: ops.gaussian_blur(t, 11, 2.0),
           "mean17": lambda t: ops.mean_blur(t, 17), "mean31": lambda t: ops.mean_blur(t, 31),
           "direct5": lambda t: ops.gaussian_blur(t, 5, 1.0, direct=True),
           "direct7": lambda t: ops.gaussian_blur(t, 7, 1.5, direct=True)}
    with ops.tuning(MI_ROWS_PER_BAND=32, MI_CHAIN_ROWS_PER_BAND=32):
        ref = {k: fn(img).clone() for k, fn in fns.items()}
        ref_chain = ops.fused_chain(img).clone()
    for rpb in (1, 2, 5, 7, 270):
        with ops.tuning(MI_ROWS_PER_BAND=rpb, MI_CHAIN_ROWS_PER_BAND=rpb):
            for k, fn in fns.items():
                assert torch.equal(fn(img), ref[k]), (k, rpb)
            assert torch.equal(ops.fused_chain(img), ref_chain), rpb


@pytest.mark.parametrize("k", [17, 23, 31])
def test_wide_mean_strips_tiles_and_oracle(dev, golden, k):
    """Mean 17x17..31x31 (the prefix-sum march): row strips filtered from their rows plus halo reproduce the
    single launch, which equals the oracle and the LDS-tile kernel it replaced (MI_MEAN_PREFIX=0)."""
    w, h = 1000, 300
    img = ops.synth(k, w, h)
    # (the matrix-core kernels take these sizes by default since round 3: MI_MEAN_MFMA=0 brings the march back)
    with ops.tuning(MI_MEAN_MFMA=0, restore={"MI_MEAN_MFMA": 11}):
        whole = ops.mean_blur(img, k)
        assert np.array_equal(host(whole), golden.meanblur(host(img), k))
        with ops.tuning(MI_MEAN_PREFIX=0, restore={"MI_MEAN_PREFIX": 1}):
            assert torch.equal(ops.mean_blur(img, k), whole)
        halo = ops.halo_rows(_lib.FILTER_MEAN, k)
        for y0, y1 in ((0, 100), (100, 117), (117, 300)):
            r0, r1 = max(0, y0 - halo), min(h, y1 + halo)
            out = torch.empty((y1 - y0, w, 4), dtype=torch.uint8, device=dev)
            ops.filter_strip(_lib.FILTER_MEAN, img[r0:r1], out, w, h, r0, y0, y1, k, 0.0)
            assert torch.equal(out, whole[y0:y1]), (y0, y1)
    assert torch.equal(ops.mean_blur(img, k), whole)          # the default path: same bytes


def test_strips_of_images_narrower_than_the_radius(dev, golden):
    """Width 4 (and 3, 5, 7: the 64-bit-address policy) with radii up to 11: the flat-index taps reach
    ceil(r / w) rows up and down, so a strip needs r + ceil(r / w) rows of halo (ops.halo_rows(.., w)); with them
    the stitched strips equal the single launch, which equals the oracle; with only r + 1 the strip entry refuses."""
    for w, h in ((4, 130), (3, 200), (5, 150), (7, 120)):
        img = ops.synth(w + h, w, h)
        for filt, k, sigma in ((_lib.FILTER_MEAN, 17, 0.0), (_lib.FILTER_MEAN, 13, 0.0), (_lib.FILTER_MEAN, 23, 0.0),
                               (_lib.FILTER_GAUSSIAN, 15, 9.0), (_lib.FILTER_GAUSSIAN, 9, 4.0), (_lib.FILTER_MEDIAN, 5, 0.0)):
            if k * k > w * h:
                continue
            whole = ops.mean_blur(img, k) if filt == _lib.FILTER_MEAN else (
                ops.gaussian_blur(img, k, sigma) if filt == _lib.FILTER_GAUSSIAN else ops.median_blur(img, k))
            if filt == _lib.FILTER_MEAN:
                assert np.array_equal(host(whole), golden.meanblur(host(img), k)), (w, h, k)
            halo = ops.halo_rows(filt, k, w)
            assert halo == k // 2 + -(-(k // 2) // w)
            for y0, y1 in ((0, 41), (41, 42), (42, 97), (97, h)):
                r0, r1 = max(0, y0 - halo), min(h, y1 + halo)
                out = torch.empty((y1 - y0, w, 4), dtype=torch.uint8, device=dev)
                ops.filter_strip(filt, img[r0:r1].contiguous(), out, w, h, r0, y0, y1, k, sigma)
                assert torch.equal(out, whole[y0:y1]), (w, h, filt, k, y0, y1)
            if halo > k // 2 + 1:
                short = k // 2 + 1
                out = torch.empty((10, w, 4), dtype=torch.uint8, device=dev)
                with pytest.raises(Exception):
                    ops.filter_strip(filt, img[60 - short:70 + short].contiguous(), out, w, h, 60 - short, 60, 70, k, sigma)


def test_wide_mean_odd_shapes_equal_the_tile_kernel(dev):
    """Shapes around the prefix-sum march's strip width (224 pixels), unaligned widths (the 64-bit-address
    policy), batches, a very wide image and images smaller than the kernel: same bytes as the LDS-tile
    kernel, which the oracle tests pin."""
    for (w, h, b) in ((32768, 256, 1), (8190, 1031, 1), (1000, 700, 3), (224, 40, 1), (225, 40, 2), (4, 4, 1), (3, 3, 1)):
        img = ops.synth(7, w, h, batch=b) if b > 1 else ops.synth(7, w, h)
        for k in (17, 31):
            default = ops.mean_blur(img, k)
            with ops.tuning(MI_MEAN_MFMA=0, restore={"MI_MEAN_MFMA": 11}):
                got = ops.mean_blur(img, k)
                with ops.tuning(MI_MEAN_PREFIX=0, restore={"MI_MEAN_PREFIX": 1}):
                    assert torch.equal(got, ops.mean_blur(img, k)), (w, h, b, k)
            assert torch.equal(default, got), (w, h, b, k)     # the matrix-core kernels where they apply


def test_large_batch_properties(dev):
    """Size-independent properties on a batch too large for the oracle: a batch result equals the
    same frames processed one by one, and a constant image stays constant inside the write set."""
    w, h, b = 1920, 1080, 8
    frames = ops.synth(1000, w, h, batch=b)
    for fn in (ops.grayscale, ops.sobel, lambda t: ops.median_blur(t, 3), lambda t: ops.gaussian_blur(t, 5, 1.0),
               lambda t: ops.mean_blur(t, 3)):
        whole = fn(frames)
        for f in (0, 3, 7):
            assert torch.equal(whole[f], fn(frames[f].contiguous()))
    const = torch.full((h, w, 4), 77, dtype=torch.uint8, device=dev)
    m = ops.median_blur(const, 3)
    assert int(m[2:-2, 2:-2, :3].min()) == 77 and int(m[2:-2, 2:-2, :3].max()) == 77
    assert int(ops.sobel(const)[2:-2, 2:-2, 0].max()) == 0


@pytest.mark.parametrize("filt,k,sigma", [(_lib.FILTER_GAUSSIAN, 15, 2.5), (_lib.FILTER_MEDIAN, 5, 0.0),
                                          (_lib.FILTER_GAUSSIAN, 31, 5.0), (_lib.FILTER_MEAN, 31, 0.0)])
def test_config4_full_size_strips_and_sampled_oracle(dev, golden, filt, k, sigma):
    """BASELINE.json config 4 at full size on one GPU: a 32768 x 32768 image (4 GiB in, 4 GiB out,
    flat pixel indices beyond 2^30), Gaussian 15x15 and median 5x5 -- and the radius-15 matrix-core kernels
    (Gaussian and mean 31x31).  The eight 4096-row strips of
    the multi-GPU decomposition, each filtered from its rows plus halo, must reproduce the single
    launch bit for bit; bands of rows at the top, at a strip boundary, in the upper half of the
    address range and at the bottom are checked against the oracle run on those rows alone (a row
    depends only on rows within the halo)."""
    w = h = 32768
    free, _ = torch.cuda.mem_get_info()
    if free < 11 * 2 ** 30:
        pytest.skip("needs 11 GiB of device memory")
    img = ops.synth(4, w, h)
    whole = run_full(filt, img, k, sigma)
    halo = ops.halo_rows(filt, k)
    out = torch.empty((4096, w, 4), dtype=torch.uint8, device=dev)
    for p in range(8):
        y0, y1 = 4096 * p, 4096 * (p + 1)
        r0, r1 = max(0, y0 - halo), min(h, y1 + halo)
        ops.filter_strip(filt, img[r0:r1], out, w, h, r0, y0, y1, k, sigma)
        assert torch.equal(out, whole[y0:y1]), p
    pad = halo + 1
    for y0, y1 in ((0, 4), (4094, 4098), (20481, 20484), (h - 4, h)):
        r0, r1 = max(0, y0 - pad), min(h, y1 + pad)
        band = host(img[r0:r1])
        ref = (golden.gaussian(band, k, sigma, threads=8) if filt == _lib.FILTER_GAUSSIAN else
               golden.meanblur(band, k, threads=8) if filt == _lib.FILTER_MEAN else golden.medianblur(band, k, threads=8))
        got = host(whole[y0:y1]).astype(np.int16)
        want = ref[y0 - r0:y1 - r0].astype(np.int16)
        if r0 > 0 and y0 - r0 < pad or r1 < h and r1 - y1 < pad:
            raise AssertionError("band padding too small")
        # the oracle saw a shorter image: its write set ends at its own last row
        if y0 == 0:
            want[0] = got[0]
        assert np.abs(got - want).max() <= (GAUSS_TOL if filt == _lib.FILTER_GAUSSIAN else 0), (y0, y1)
    del whole, out, img
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ host tier (reference names)
def test_host_tier_wrappers(dev, golden, tmp_path):
    w, h = 131, 67
    img = S.synth_image(77, w, h)
    out = np.empty_like(img)
    ops.imageGrayScaleWrapper(out, img, w, h)
    assert np.array_equal(out, golden.grayscale(img))
    ops.imageSobelEdgeWrapper(out, img, w, h)
    assert np.array_equal(out, golden.sobel(img))
    ops.imageMeanBlurWrapper(out, img, w, h)
    assert np.array_equal(out, golden.meanblur(img, 3))
    ops.imageMedianBlurWrapper(out, img, w, h, 3)
    assert np.array_equal(out, golden.medianblur(img, 3))
    ops.imageGaussianBlurWrapper(out, img, w, h, 3, 7.0)        # the reference's call, main.cpp:41: bit-exact by default
    assert np.array_equal(out, golden.gaussian(img, 3, 7.0))
    ops.imageGaussianBlurWrapper(out, img, w, h, 3, 7.0, direct=False)
    assert np.abs(out.astype(np.int16) - golden.gaussian(img, 3, 7.0).astype(np.int16)).max() <= GAUSS_TOL
    ops.imageGaussianBlurWrapper(out, img, w, h, 21, 4.0)       # any odd size, like upstream (:478-502)
    assert np.array_equal(out, golden.gaussian(img, 21, 4.0))
    # in place, as main.cpp:38-43 calls every wrapper with (img, img, ...)
    alias = img.copy()
    ops.imageSobelEdgeWrapper(alias, alias, w, h)
    assert np.array_equal(alias, golden.sobel(img))
    loaded = np.empty(w * h * 4, np.uint8)
    ops.imageLoadWrapper(img.reshape(-1), loaded, w * h)
    assert np.array_equal(loaded, img.reshape(-1))


def test_device_load_write_copy(dev):
    img = ops.synth(8, 123, 45)
    flat = ops.image_write(img)
    assert torch.equal(flat, img.reshape(-1))
    back = ops.image_load(flat)
    assert torch.equal(back, img.reshape(-1))
    odd = flat[1:1 + 4 * 99]                     # a view: unaligned source pointer, byte path
    assert torch.equal(ops.image_load(odd), odd)


# ------------------------------------------------------------------ randomized differential test
def test_random_shapes_and_filters_against_oracle(dev, golden):
    """120 seeded random (shape, filter, parameter, pattern) cases against the oracle: widths around the
    strip widths (248, 256) and their multiples, heights around the band heights, both access
    policies (widths that are and are not multiples of 4)."""
    rng = np.random.default_rng(20240607)
    special_w = [3, 4, 5, 7, 8, 247, 248, 249, 255, 256, 257, 260, 495, 496, 497, 511, 512, 513, 744, 768, 1000]
    special_h = [3, 4, 5, 6, 7, 8, 11, 12, 13, 29, 30, 31, 32, 33, 59, 60, 61, 64, 65, 127, 128, 129]
    for case in range(120):
        w = int(rng.choice(special_w)) if rng.random() < 0.6 else int(rng.integers(3, 700))
        h = int(rng.choice(special_h)) if rng.random() < 0.6 else int(rng.integers(3, 200))
        kind = str(rng.choice(PATTERNS))
        img = make(kind, w, h, int(rng.integers(1, 1 << 30)))
        d = to_dev(img, dev)
        what = int(rng.integers(0, 7))
        tag = (case, w, h, kind, what)
        if what == 0:
            assert np.array_equal(host(ops.sobel(d)), golden.sobel(img)), tag
        elif what == 1:
            k = int(rng.choice([3, 5, 7, 9, 11, 13, 15, 17]))
            assert np.array_equal(host(ops.mean_blur(d, k)), golden.meanblur(img, k)), tag + (k,)
        elif what == 2:
            k = int(rng.choice([3, 5]))
            assert np.array_equal(host(ops.median_blur(d, k)), golden.medianblur(img, k)), tag + (k,)
        elif what == 3:
            k = int(rng.choice([3, 5, 7, 9, 11, 13, 15]))
            sigma = float(rng.uniform(0.6, 4.0))
            got = host(ops.gaussian_blur(d, k, sigma)).astype(np.int16)
            assert np.abs(got - golden.gaussian(img, k, sigma).astype(np.int16)).max() <= GAUSS_TOL, tag + (k, sigma)
        elif what == 4:
            k = int(rng.choice([3, 5, 7, 9]))
            sigma = float(rng.uniform(0.6, 4.0))
            assert np.array_equal(host(ops.gaussian_blur(d, k, sigma, direct=True)), golden.gaussian(img, k, sigma)), tag + (k, sigma)
        elif what == 5:
            assert np.array_equal(host(ops.grayscale(d)), golden.grayscale(img)), tag
        else:
            stages = ops.filter_chain(d)
            assert torch.equal(ops.fused_chain(d), stages[3]), tag
            assert np.array_equal(host(stages[2]), golden.medianblur(host(stages[1]), 3)), tag
            assert np.array_equal(host(stages[3]), golden.sobel(host(stages[2]))), tag


def test_config3_full_batch(dev):
    """BASELINE.json config 3 at full size on one GPU: 4096 frames of 1920x1080 (34 GB in, 34 GB
    out) through the fused chain and through the four kernels in one launch each; sampled frames
    must equal the same frames processed alone (frame offsets beyond 2^32 bytes, 1.2 M waves)."""
    w, h, b = 1920, 1080, 4096
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * 2 ** 30:
        pytest.skip("needs 150 GiB of device memory")
    frames = torch.empty((b, h, w, 4), dtype=torch.uint8, device=dev)
    for f in range(0, b, 512):                     # one synthetic stream, cut into frames
        frames[f:f + 512] = ops.synth(3, w, h, batch=512, first_idx=f * w * h, device=dev)
    out = ops.fused_chain(frames)
    t0, t1 = torch.empty_like(frames), torch.empty_like(frames)
    sample = (0, 1, 2047, 2048, 4095)
    ops.grayscale(frames, t0)
    for f in sample:
        assert torch.equal(t0[f], ops.grayscale(frames[f])), f
    ops.gaussian_blur(t0, 5, 1.0, t1)
    ops.median_blur(t1, 3, t0)
    for f in sample:
        assert torch.equal(t0[f], ops.median_blur(ops.gaussian_blur(ops.grayscale(frames[f]), 5, 1.0), 3)), f
    ops.sobel(t0, t1)
    assert torch.equal(out, t1)
    for f in sample:
        assert torch.equal(out[f], ops.fused_chain(frames[f])), f
    del frames, out, t0, t1
    torch.cuda.empty_cache()


def test_random_shapes_for_the_tiled_kernels(dev, golden):
    """60 seeded random cases for the LDS-tiled kernels (mean 17..31, direct Gaussian 9..15): widths around
    multiples of the 64-pixel tile, heights around multiples of its 16 rows, images smaller than the kernel."""
    rng = np.random.default_rng(77)
    special_w = [3, 5, 17, 31, 63, 64, 65, 66, 127, 128, 129, 191, 192, 193, 256, 300]
    special_h = [3, 5, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, 70]
    for case in range(60):
        w = int(rng.choice(special_w)) if rng.random() < 0.7 else int(rng.integers(3, 400))
        h = int(rng.choice(special_h)) if rng.random() < 0.7 else int(rng.integers(3, 120))
        img = make(str(rng.choice(PATTERNS)), w, h, int(rng.integers(1, 1 << 30)))
        d = to_dev(img, dev)
        if rng.random() < 0.5:
            k = int(rng.choice([17, 19, 21, 23, 25, 27, 29, 31]))
            assert np.array_equal(host(ops.mean_blur(d, k)), golden.meanblur(img, k)), (case, w, h, "mean", k)
        else:
            k = int(rng.choice([9, 11, 13, 15]))
            sigma = float(rng.uniform(0.8, 5.0))
            assert np.array_equal(host(ops.gaussian_blur(d, k, sigma, direct=True)), golden.gaussian(img, k, sigma)), (case, w, h, "direct", k, sigma)


def test_device_tier_runs_under_graph_capture(dev):
    """The device tier neither allocates nor synchronises: a sequence of its calls can
    be captured into a HIP graph and replayed, with the result of the eager calls."""
    frames = ops.synth(31, 640, 360, batch=3)
    outs = [torch.empty_like(frames) for _ in range(6)]
    plane = torch.complex(torch.rand(2, 256, 128, device=dev), torch.rand(2, 256, 128, device=dev))
    spec = torch.empty_like(plane)
    odd = torch.complex(torch.rand(45, 67, device=dev), torch.rand(45, 67, device=dev))
    odd_out = torch.empty_like(odd)
    ws = torch.empty(ops.fft2d_workspace_bytes(45, 67), dtype=torch.uint8, device=dev)

    def sequence():
        ops.grayscale(frames, outs[0])
        ops.gaussian_blur(outs[0], 5, 1.0, outs[1])
        ops.median_blur(outs[1], 3, outs[2])
        ops.sobel(outs[2], outs[3])
        ops.fused_chain(frames, 1.0, outs[4])
        ops.mean_blur(frames, 17, outs[5])
        ops.fft2d(plane, out=spec)
        ops.fft2d(odd, out=odd_out, workspace=ws)

    sequence()                                   # eager: reference results (and every lazy one-time set-up)
    torch.cuda.synchronize()
    want = [o.clone() for o in outs] + [spec.clone(), odd_out.clone()]
    for o in outs + [spec, odd_out]:
        o.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sequence()
    torch.cuda.synchronize()
    assert int(outs[4].view(torch.uint8).max()) == 0          # captured, not run
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    for got, ref in zip(outs + [spec, odd_out], want):
        assert torch.equal(got, ref)


@pytest.mark.parametrize("k,sigma", [(17, 3.0), (21, 4.0), (25, 2.0), (31, 6.0)])
def test_gaussian_sizes_above_15_are_bit_exact(dev, golden, k, sigma):
    """The reference takes any odd kernel size (imageTools.cu:478-502): 17..31 in the direct mode run the
    reference-order k x k sum from LDS tiles, bit-exact against the oracle, odd widths and strips included.  The
    separable mode answers with the same kernels where the matrix-core ones do not apply (odd widths, or
    MI_GAUSS_MFMA_MAX=15 -- round 2's behaviour) and within one level where they do."""
    for w, h in ((197, 83), (256, 64)):
        img = S.synth_image(k, w, h)
        want = golden.gaussian(img, k, sigma)
        d = torch.from_numpy(img).to(dev)
        assert np.array_equal(ops.gaussian_blur(d, k, sigma, direct=True).cpu().numpy(), want)
        sep = ops.gaussian_blur(d, k, sigma).cpu().numpy()
        if w % 4:
            assert np.array_equal(sep, want)
        else:
            assert np.abs(sep.astype(np.int16) - want.astype(np.int16)).max() <= GAUSS_TOL
        with ops.tuning(MI_GAUSS_MFMA_MAX=15):
            assert np.array_equal(ops.gaussian_blur(d, k, sigma).cpu().numpy(), want)
    halo = ops.halo_rows(_lib.FILTER_GAUSSIAN, k)
    assert halo == k // 2 + 1
    y0, y1 = 20, 47
    lo, hi = max(0, y0 - halo), min(h, y1 + halo)
    out = torch.empty((y1 - y0, w, 4), dtype=torch.uint8, device=dev)
    with ops.tuning(MI_GAUSS_MFMA_MAX=15):
        ops.filter_strip(_lib.FILTER_GAUSSIAN, d[lo:hi].contiguous(), out, w, h, lo, y0, y1, k, sigma)
    assert np.array_equal(out.cpu().numpy(), want[y0:y1])
    ops.filter_strip(_lib.FILTER_GAUSSIAN_DIRECT, d[lo:hi].contiguous(), out, w, h, lo, y0, y1, k, sigma)
    assert np.array_equal(out.cpu().numpy(), want[y0:y1])


def test_very_wide_image_leaves_the_32bit_offset_path(dev, golden):
    """A band of a 12-million-pixel-wide image spans more than 2^31 bytes: the launcher must take the
    64-bit-address policy instead of wrapping the buffer offsets (or running into their sentinel)."""
    w, h = 12_000_000, 3
    img = S.synth_image(5, w, h)
    d = torch.from_numpy(img).to(dev)
    assert np.array_equal(ops.sobel(d).cpu().numpy(), golden.sobel(img))
    assert np.array_equal(ops.mean_blur(d, 3).cpu().numpy(), golden.meanblur(img, 3))


@pytest.mark.parametrize("k", [7, 9, 11, 13, 15])
def test_wide_gaussians_stay_within_one_level(dev, golden, k):
    """Separable Gaussians of 7x7..15x15 run on the matrix cores (gauss_mfma_kernel: f16 operands with two-part
    weights, fp32 accumulation; error < 0.1 before the truncating store), with the 16-bit fixed-point kernel on the
    vector ALUs (gauss_wide_kernel, MI_GAUSS_MFMA=0; < 0.07) and the fp32 kernel (MI_GAUSS_WIDE=0) behind it: all
    within 1 level of the oracle's direct fp32 sum for narrow, usual and very wide sigmas, on noise (every tap
    matters) and on a smooth image (values sit near integers, where truncation flips); a constant image comes out
    exactly constant in the first two (the plain fp32 sum may truncate to the level below); a sigma so small that a weight would not fit 16 bits takes the fp32 kernel, same bar."""
    w, h = 516, 75
    imgs = [S.synth_image(k, w, h), S.pattern_image("natural", w, h, k)]
    variants = ({}, {"MI_GAUSS_MFMA": 0}, {"MI_GAUSS_WIDE": 0})
    for sigma in (0.05, 0.7, 2.5, 10.0):
        for img in imgs:
            want = golden.gaussian(img, k, sigma).astype(np.int16)
            d = torch.from_numpy(img).to(dev)
            for knobs in variants:
                with ops.tuning(**knobs):
                    got = ops.gaussian_blur(d, k, sigma).cpu().numpy().astype(np.int16)
                assert np.abs(got - want).max() <= GAUSS_TOL, (k, sigma, knobs)
                if not knobs and sigma == 2.5:
                    # the matrix-core kernel's rate of one-level differences stays small (measured: 0.3 - 0.8 %)
                    assert (got != want).mean() < 0.02, (k, sigma)
    r = k // 2
    for knobs in variants[:2]:
        with ops.tuning(**knobs):
            for v in (0, 1, 77, 128, 254, 255):
                const = torch.full((h, w, 4), v, dtype=torch.uint8, device=dev)
                for sigma in (2.0, 0.7, 10.0):
                    out = ops.gaussian_blur(const, k, sigma)
                    inner = out[r + 1:h - r - 1, r:w - r, :3]
                    assert int(inner.min()) == v and int(inner.max()) == v, (k, sigma, v, knobs)


@pytest.mark.parametrize("k", [17, 19, 21, 23, 25, 27, 29, 31])
def test_separable_gaussians_up_to_31_stay_within_one_level(dev, golden, k):
    """MI_FILTER_GAUSSIAN for 17x17 (gauss_mfma_kernel) and 19x19 .. 31x31 (gauss_mfma_wide_kernel: aligned 16-pixel
    halves and 16-row blocks, three K = 16 products per pass and part): within one level of the oracle's direct sum for
    narrow to very wide sigmas, constants exact, independent of band height, strips and batch position; with
    MI_GAUSS_MFMA_MAX=15, or for an image the matrix-core kernels do not take (width not a multiple of 4), the direct
    kernel answers, bit-exact."""
    w, h = 516, 110
    imgs = [S.synth_image(k, w, h), S.pattern_image("natural", w, h, k)]
    for sigma in (0.3, 2.5, 6.0, 40.0):
        for img in imgs:
            want = golden.gaussian(img, k, sigma, threads=8)
            d = torch.from_numpy(img).to(dev)
            got = ops.gaussian_blur(d, k, sigma).cpu().numpy()
            assert np.abs(got.astype(np.int16) - want.astype(np.int16)).max() <= GAUSS_TOL, (k, sigma)
            if sigma == 6.0:
                assert (got != want).mean() < 0.02, (k, sigma)
                with ops.tuning(MI_GAUSS_MFMA_MAX=15, restore={"MI_GAUSS_MFMA_MAX": 31}):
                    assert np.array_equal(ops.gaussian_blur(d, k, sigma).cpu().numpy(), want), (k, sigma)
    r = k // 2
    for v in (0, 1, 77, 128, 254, 255):
        const = torch.full((h, w, 4), v, dtype=torch.uint8, device=dev)
        for sigma in (2.0, 0.7, 20.0):
            out = ops.gaussian_blur(const, k, sigma)
            inner = out[r + 1:h - r - 1, r:w - r, :3]
            assert int(inner.min()) == v and int(inner.max()) == v, (k, sigma, v)
    odd = S.synth_image(5, 131, 90)
    assert np.array_equal(ops.gaussian_blur(torch.from_numpy(odd).to(dev), k, 3.0).cpu().numpy(), golden.gaussian(odd, k, 3.0, threads=8))
    w, h = 1000, 203
    img = S.synth_image(31, w, h)
    d = torch.from_numpy(img).to(dev)
    ref = ops.gaussian_blur(d, k, 4.0)
    for rpb in (1, 16, 40, 96, 512):
        with ops.tuning(MI_ROWS_PER_BAND=rpb):
            assert torch.equal(ops.gaussian_blur(d, k, 4.0), ref), (k, rpb)
            with ops.tuning(MI_MFMA_SERPENTINE=0):      # every band downwards: the same bytes
                assert torch.equal(ops.gaussian_blur(d, k, 4.0), ref), (k, rpb)
    halo = ops.halo_rows(_lib.FILTER_GAUSSIAN, k)
    for edges in ((0, 1, 2, 9, 33, 100, 101, 203), (0, 77, 203)):
        pieces = []
        for y0, y1 in zip(edges[:-1], edges[1:]):
            r0, r1 = max(0, y0 - halo), min(h, y1 + halo)
            out = torch.empty((y1 - y0, w, 4), dtype=torch.uint8, device=dev)
            ops.filter_strip(_lib.FILTER_GAUSSIAN, d[r0:r1].contiguous(), out, w, h, r0, y0, y1, k, 4.0)
            pieces.append(out)
        assert torch.equal(torch.cat(pieces), ref), (k, edges)
    batch = torch.stack([d, torch.flip(d, dims=(0,)), d]).contiguous()
    got = ops.gaussian_blur(batch, k, 4.0)
    assert torch.equal(got[0], ref) and torch.equal(got[2], ref)


def test_matrix_core_gaussian_does_not_depend_on_the_decomposition(dev, golden):
    """gauss_mfma_kernel works on 16-row blocks and 16-pixel tiles aligned to absolute image coordinates, so an
    output pixel goes through the same operations whatever the band height, the strip a row belongs to, or the
    batch position: bit-identical for every split, on a width that is not a multiple of the tile or the strip."""
    w, h = 1000, 203
    img = S.synth_image(31, w, h)
    d = torch.from_numpy(img).to(dev)
    for k, sigma in ((15, 2.5), (9, 1.2)):
        ref = ops.gaussian_blur(d, k, sigma)
        assert np.abs(ref.cpu().numpy().astype(np.int16) - golden.gaussian(img, k, sigma).astype(np.int16)).max() <= GAUSS_TOL
        for rpb in (1, 16, 40, 96, 512):
            with ops.tuning(MI_ROWS_PER_BAND=rpb):
                assert torch.equal(ops.gaussian_blur(d, k, sigma), ref), (k, rpb)
                for strip in (64, 256):      # 64-pixel strips (three waves per SIMD), 256-pixel strips (one): the same bytes
                    with ops.tuning(MI_GM_STRIP=strip):
                        assert torch.equal(ops.gaussian_blur(d, k, sigma), ref), (k, rpb, strip)
        halo = ops.halo_rows(_lib.FILTER_GAUSSIAN, k)
        for edges in ((0, 1, 2, 9, 33, 100, 101, 203), (0, 77, 203)):
            pieces = []
            for y0, y1 in zip(edges[:-1], edges[1:]):
                r0, r1 = max(0, y0 - halo), min(h, y1 + halo)
                out = torch.empty((y1 - y0, w, 4), dtype=torch.uint8, device=dev)
                ops.filter_strip(_lib.FILTER_GAUSSIAN, d[r0:r1].contiguous(), out, w, h, r0, y0, y1, k, sigma)
                pieces.append(out)
            assert torch.equal(torch.cat(pieces), ref), (k, edges)
        batch = torch.stack([d, torch.flip(d, dims=(0,)), d]).contiguous()
        got = ops.gaussian_blur(batch, k, sigma)
        assert torch.equal(got[0], ref) and torch.equal(got[2], ref)
