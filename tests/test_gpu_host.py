"""GPU tests of the host tier's staging pipeline (mi_run_ops_u8x4 and the single-filter wrappers
that run through it): chunked, double-buffered H2D / kernels / D2H must give exactly what the
whole-image device path and the CPU oracle give, for any chunk height, with `out` aliasing `in`
as the reference's demo calls its wrappers (main.cpp:38-43).  Also BASELINE.json config 1: the
reference's own sample image, decoded on the host, gray + Gaussian 5x5, PNG out."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from cudaimageprocessing_amd import _lib, ops, synth as S  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    _lib.lib()
    return torch.device("cuda:0")


def device_chain(img, op_list, dev):
    """The same filters through the device tier, whole image, one launch per filter."""
    x = torch.from_numpy(img).to(dev)
    for f, k, s in op_list:
        if f == _lib.FILTER_GRAYSCALE:
            x = ops.grayscale(x)
        elif f == _lib.FILTER_SOBEL:
            x = ops.sobel(x)
        elif f == _lib.FILTER_GAUSSIAN:
            x = ops.gaussian_blur(x, k, s)
        elif f == _lib.FILTER_GAUSSIAN_DIRECT:
            x = ops.gaussian_blur(x, k, s, direct=True)
        elif f == _lib.FILTER_MEAN:
            x = ops.mean_blur(x, k)
        elif f == _lib.FILTER_MEDIAN:
            x = ops.median_blur(x, k)
    return x.cpu().numpy()


CHAINS = ["gray", "sobel", "gray,gauss:5:1.0,median:3,sobel", "gauss:15:2.5,median:5",
          "mean:3,mean:7,gauss_direct:3:7.0", "median:5,gray,sobel,median:3,gauss:9:2.0"]


@pytest.mark.parametrize("spec", CHAINS)
@pytest.mark.parametrize("w,h,chunk", [(67, 45, 1), (67, 45, 7), (256, 96, 16), (250, 131, 40), (64, 30, 1000)])
def test_pipeline_equals_whole_image(dev, spec, w, h, chunk):
    img = S.pattern_image("natural", w, h, 7)
    op_list = ops.parse_ops(spec)
    want = device_chain(img, op_list, dev)
    out = np.empty_like(img)
    ops.run_ops(out, img, w, h, op_list, chunk_rows=chunk)
    assert np.array_equal(out, want)
    # in place, as the reference's demo calls its wrappers
    buf = img.copy()
    ops.run_ops(buf, buf, w, h, op_list, chunk_rows=chunk)
    assert np.array_equal(buf, want)


@pytest.mark.parametrize("spec", ["mean:17", "mean:31,gauss:23:4.0", "gauss:31:6.0,mean:19", "median:5,mean:13"])
@pytest.mark.parametrize("w,h,chunk", [(4, 400, 16), (4, 400, 1), (8, 300, 33), (516, 150, 40), (1000, 203, 64)])
def test_pipeline_wide_kernels_and_narrow_images(dev, spec, w, h, chunk):
    """Chunks of the host pipeline with the radius-9..15 matrix-core kernels in the chain, and on images narrower than
    the radius, where a chunk needs r + ceil(r / w) rows of halo per stage (the flat-index taps reach further than
    one row): every chunking gives the whole-image result."""
    op_list = ops.parse_ops(spec)
    if any(k * k > w * h for _, k, _ in op_list):
        pytest.skip("kernel larger than the image")
    img = S.synth_image(w + chunk, w, h)
    want = device_chain(img, op_list, dev)
    out = np.empty_like(img)
    ops.run_ops(out, img, w, h, op_list, chunk_rows=chunk)
    assert np.array_equal(out, want)
    buf = img.copy()
    ops.run_ops(buf, buf, w, h, op_list, chunk_rows=chunk)
    assert np.array_equal(buf, want)


def test_pipeline_against_oracle(dev, golden):
    w, h = 320, 200
    img = S.synth_image(3, w, h)
    out = np.empty_like(img)
    ops.run_ops(out, img, w, h, "gray,gauss:5:1.0,median:3,sobel", chunk_rows=23)
    g = golden.grayscale(img)
    g = golden.gaussian(g, 5, 1.0)
    # separable vs the reference's direct 2-D sum: +-1 LSB after the Gaussian, so compare from
    # the library's own Gaussian output on (the later stages are exact)
    lib_gauss = np.empty_like(img)
    ops.run_ops(lib_gauss, img, w, h, "gray,gauss:5:1.0", chunk_rows=50)
    assert np.abs(lib_gauss.astype(np.int16) - g.astype(np.int16)).max() <= 1
    want = golden.sobel(golden.medianblur(lib_gauss, 3))
    assert np.array_equal(out, want)


def test_default_chunking_large_image(dev):
    """Default chunk height (about 8 MiB): several chunks on a 4096 x 3000 image."""
    w, h = 4096, 3000
    img = ops.synth(9, w, h).cpu().numpy()
    op_list = ops.parse_ops("gauss:5:1.0,median:3,sobel")
    want = device_chain(img, op_list, dev)
    buf = img.copy()
    ops.run_ops(buf, buf, w, h, op_list)
    assert np.array_equal(buf, want)
    single = np.empty_like(img)
    ops.imageMedianBlurWrapper(single, img, w, h, 5)
    assert np.array_equal(single, device_chain(img, [(_lib.FILTER_MEDIAN, 5, 0.0)], dev))
    _lib.check(_lib.lib().mi_host_release())
    ops.imageSobelEdgeWrapper(single, img, w, h)                      # buffers come back after a release
    assert np.array_equal(single, device_chain(img, [(_lib.FILTER_SOBEL, 3, 0.0)], dev))


def test_run_ops_argument_errors(dev):
    img = np.zeros((8, 8, 4), np.uint8)
    with pytest.raises(_lib.MiError):
        ops.run_ops(img, img, 8, 8, [])
    with pytest.raises(_lib.MiError):
        ops.run_ops(img, img, 8, 8, [(_lib.FILTER_MEDIAN, 7, 0.0)])
    with pytest.raises(_lib.MiError):
        ops.run_ops(img, img, 8, 8, [(_lib.FILTER_SOBEL, 3, 0.0)], chunk_rows=0)
    with pytest.raises(ValueError):
        ops.parse_ops("sharpen")


def _png_pixels(path):
    """RGBA pixels of a PNG mi_image_write produced (the reference's encoder, stb_image_write: any filter type)."""
    from PIL import Image
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    return np.asarray(Image.open(path).convert("RGBA"))


def test_config1_sample_image_gray_gauss_png(dev, golden, tmp_path):
    """BASELINE.json config 1: the reference's image/deer.jpeg (tests/golden/deer.jpeg, 2592x2000),
    gray -> Gaussian 5x5 sigma 1.0 -> PNG.  The JPEG is decoded as the reference's loadImage decodes it --
    stbi_load(..., STBI_rgb_alpha) of stb_image v2.30 (main.cpp:14), through the checker-side shim
    oracle/stb_decode.c -- and the direct Gaussian (imageGaussianBlurWrapper's own form) must match the oracle
    bit for bit; bench.py's c1 object times the same steps and prints the same CRC."""
    import zlib as _z
    import bench
    Image = pytest.importorskip("PIL.Image")
    rgba = golden.stb_decode(os.path.join(HERE, "golden", "deer.jpeg"))
    h, w = rgba.shape[:2]
    assert (w, h) == (2592, 2000)
    loaded = np.empty_like(rgba)
    ops.imageLoadWrapper(rgba, loaded, w * h)
    assert np.array_equal(loaded, rgba)
    out = np.empty_like(rgba)
    ops.run_ops(out, loaded, w, h, "gray,gauss:5:1.0")
    ref = golden.gaussian(golden.grayscale(rgba, threads=8), 5, 1.0, threads=8)
    diff = np.abs(out.astype(np.int16) - ref.astype(np.int16))
    assert diff.max() <= 1
    exact = np.empty_like(rgba)
    ops.run_ops(exact, loaded, w, h, "gray,gauss_direct:5:1.0")
    assert np.array_equal(exact, ref)
    c1 = bench.leg_c1(ops)
    assert c1["cpu"]["crc32_output"] == (_z.crc32(ref.tobytes()) & 0xffffffff)
    assert c1["gpu_host_tier"]["crc32_output"] == (_z.crc32(out.tobytes()) & 0xffffffff)
    assert c1["gpu_host_tier"]["direct_gaussian_bit_exact_vs_cpu"] and c1["gpu_host_tier"]["max_abs_diff_vs_cpu"] <= 1
    gray = np.empty_like(rgba)
    ops.imageGrayScaleWrapper(gray, loaded, w, h)
    assert np.array_equal(gray, golden.grayscale(rgba, threads=8))
    path = str(tmp_path / "deer_blur.png")
    ops.imageWriteWrapper(path, out, w, h)
    assert np.array_equal(_png_pixels(path), out)
    assert np.array_equal(np.asarray(Image.open(path)), out)


def test_host_tier_calls_from_several_threads(dev):
    """The host tier keeps cached staging and scratch buffers behind mutexes: calls from concurrent
    threads (ctypes releases the GIL) serialise and give the single-threaded results."""
    import threading
    w, h = 333, 217
    imgs = [S.synth_image(900 + i, w, h) for i in range(4)]
    spec = ops.parse_ops("gray,gauss:5:1.0,median:3,sobel")

    def work(img):
        a = np.empty_like(img); ops.run_ops(a, img, w, h, spec)
        b = np.empty_like(img); ops.imageFFTImageGenerate(b, img, w, h)
        c = np.empty_like(img); ops.compressImage(c, img, w, h, 0.2)
        d = np.empty_like(img); ops.imageMeanBlurWrapper(d, img, w, h)
        return a, b, c, d

    want = [work(img) for img in imgs]
    got = [[None] * 6 for _ in imgs]
    errors = []

    def runner(i, rep):
        try:
            got[i][rep] = work(imgs[i])
        except Exception as exc:      # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=runner, args=(i, rep)) for rep in range(6) for i in range(len(imgs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(len(imgs)):
        for rep in range(6):
            for x, y in zip(got[i][rep], want[i]):
                assert np.array_equal(x, y)


def test_multi_device_entry_points_of_the_c_abi(dev, golden):
    """mi_batch_run_ops_u8x4 / mi_strip_run_ops_u8x4: one host thread per listed device.  This box has one
    GPU, so ordinal 0 is listed one, two and three times (the threads take turns on its pipe): what is under
    test is the partition into frames / row strips with their halos, against mi_run_ops_u8x4 and the oracle."""
    spec = "gray,gauss_direct:5:1.0,median:3,sobel,mean:7,median:5"
    w, h, n = 200, 157, 5
    frames = np.stack([S.synth_image(40 + f, w, h) for f in range(n)])
    want = np.empty_like(frames)
    for f in range(n):
        ops.run_ops(want[f], frames[f], w, h, spec)
    ref = golden.medianblur(golden.meanblur(golden.sobel(golden.medianblur(golden.gaussian(golden.grayscale(frames[0]), 5, 1.0), 3)), 7), 5)
    assert np.array_equal(want[0], ref)
    for devices in ((0,), (0, 0), (0, 0, 0)):
        out = np.zeros_like(frames)
        ops.batch_run_ops(out, frames, spec, devices)
        assert np.array_equal(out, want), devices
        one = np.zeros_like(frames[0])
        ops.strip_run_ops(one, frames[1], spec, devices)
        assert np.array_equal(one, want[1]), devices
    # in place: frames always, strips only on one device
    buf = frames.copy()
    ops.batch_run_ops(buf, buf, spec, (0, 0))
    assert np.array_equal(buf, want)
    buf = frames[2].copy()
    ops.strip_run_ops(buf, buf, spec, (0,))
    assert np.array_equal(buf, want[2])
    with pytest.raises(_lib.MiError) as e:
        ops.strip_run_ops(buf, buf, spec, (0, 0))
    assert e.value.code == -4
    with pytest.raises(_lib.MiError) as e:
        ops.strip_run_ops(np.zeros_like(buf), buf, spec, (0, 7))
    assert e.value.code == -1



@pytest.mark.parametrize("threads", [1, 4])
def test_staged_whole_image_copies(dev, threads):
    """Host-tier calls that need the whole image on the device (imageFFTImageGenerate, imageLoadWrapper)
    move images of 12 MiB or more through the pinned staging buffers in chunks: same bytes as the device
    tier on the same image, for one staging thread and for several, sizes that are and are not a whole
    number of chunks."""
    with ops.tuning(MI_HOST_COPY_THREADS=threads, restore={"MI_HOST_COPY_THREADS": 4}):
        for w, h in ((2048, 2048), (2050, 1777)):
            img = S.pattern_image("natural", w, h, 9)
            out = np.empty_like(img)
            ops.imageFFTImageGenerate(out, img, w, h)
            want = ops.fft_highpass(torch.from_numpy(img).to(dev)).cpu().numpy()
            assert np.array_equal(out, want), (w, h)
        n = 5_000_001                                            # 20 MB, not a multiple of anything
        px = np.random.default_rng(3).integers(0, 256, (n, 4), dtype=np.uint8)
        loaded = np.empty_like(px)
        ops.imageLoadWrapper(px.reshape(-1), loaded, n)
        assert np.array_equal(loaded, px)


def test_host_release_concurrent_with_spectral_call(dev):
    """mi_host_release takes a device's scratch mutex and then its pipe mutex -- the order the spectral
    round trips take them in (scratch for the whole call, pipe for their staged copies of images of 12 MiB
    or more).  With the order reversed the two deadlocked; run them against each other, under a watchdog."""
    import threading
    w = h = 2048
    img = S.pattern_image("natural", w, h, 11)
    want = np.empty_like(img)
    ops.imageFFTImageGenerate(want, img, w, h)
    stop = threading.Event()
    errors = []

    def releaser():
        while not stop.is_set():
            try:
                _lib.check(_lib.lib().mi_host_release())
            except Exception as exc:      # noqa: BLE001
                errors.append(exc)
                return

    def caller():
        try:
            for _ in range(6):
                out = np.empty_like(img)
                ops.imageFFTImageGenerate(out, img, w, h)
                assert np.array_equal(out, want)
        except Exception as exc:      # noqa: BLE001
            errors.append(exc)

    r = threading.Thread(target=releaser, daemon=True)
    c = threading.Thread(target=caller, daemon=True)
    r.start(); c.start()
    c.join(timeout=90)
    stop.set()
    r.join(timeout=30)
    assert not c.is_alive() and not r.is_alive(), "deadlock between mi_host_release and mi_fft_highpass_u8x4"
    assert not errors, errors


@pytest.mark.parametrize("n_strips", [1, 2, 3, 5])
def test_device_resident_strips_behind_the_c_abi(dev, golden, n_strips):
    """mi_strip_stage_u8x4_dev: an image in row strips that stay on the devices, neighbour halos copied device
    to device into poisoned pad rows, interior rows filtered meanwhile.  One GPU here, so ordinal 0 is listed
    n times (as test_multi_device_entry_points_of_the_c_abi does): what is under test is the decomposition --
    the stitched strips of a chain of stages equal the single launch on the whole image, bit for bit, and the
    oracle."""
    w, h = 264, 301
    img = S.pattern_image("natural", w, h, 21)
    whole = torch.from_numpy(img).to(dev)
    stages = [(_lib.FILTER_SOBEL, 3, 0.0), (_lib.FILTER_MEDIAN, 5, 0.0), (_lib.FILTER_GAUSSIAN, 15, 2.5),
              (_lib.FILTER_MEAN, 9, 0.0), (_lib.FILTER_GAUSSIAN_DIRECT, 5, 1.0)]
    pad = max(_lib.lib().mi_filter_halo_rows(f, k) for f, k, _ in stages)
    devices = (0,) * n_strips
    rows = [ops.strip_rows(h, n_strips, i) for i in range(n_strips)]
    assert rows[0][0] == 0 and rows[-1][1] == h and all(rows[i][1] == rows[i + 1][0] for i in range(n_strips - 1))
    bufs = [[torch.full((pad + y1 - y0 + pad, w, 4), 0xA5, dtype=torch.uint8, device=dev) for (y0, y1) in rows] for _ in range(2)]
    for i, (y0, y1) in enumerate(rows):
        bufs[0][i][pad:pad + y1 - y0] = whole[y0:y1]
    cur, want, ref = 0, whole, img
    for f, k, s in stages:
        for b in bufs[cur ^ 1]:
            b.fill_(0x5A)                                   # the output buffers' pad rows and strips start poisoned
        ops.strip_stage(bufs[cur], bufs[cur ^ 1], w, h, devices, pad, f, k, s)
        cur ^= 1
        want = {_lib.FILTER_SOBEL: lambda t: ops.sobel(t), _lib.FILTER_MEDIAN: lambda t: ops.median_blur(t, k),
                _lib.FILTER_GAUSSIAN: lambda t: ops.gaussian_blur(t, k, s), _lib.FILTER_MEAN: lambda t: ops.mean_blur(t, k),
                _lib.FILTER_GAUSSIAN_DIRECT: lambda t: ops.gaussian_blur(t, k, s, direct=True)}[f](want)
        got = torch.cat([bufs[cur][i][pad:pad + y1 - y0] for i, (y0, y1) in enumerate(rows)])
        assert torch.equal(got, want), (n_strips, f, k)
    # the oracle on the first two (bit-exact) stages
    two = golden.medianblur(golden.sobel(img), 5)
    b2 = [[torch.full((pad + y1 - y0 + pad, w, 4), 0xA5, dtype=torch.uint8, device=dev) for (y0, y1) in rows] for _ in range(2)]
    for i, (y0, y1) in enumerate(rows):
        b2[0][i][pad:pad + y1 - y0] = whole[y0:y1]
    ops.strip_stage(b2[0], b2[1], w, h, devices, pad, _lib.FILTER_SOBEL, 3, 0.0)
    ops.strip_stage(b2[1], b2[0], w, h, devices, pad, _lib.FILTER_MEDIAN, 5, 0.0)
    got = torch.cat([b2[0][i][pad:pad + y1 - y0] for i, (y0, y1) in enumerate(rows)]).cpu().numpy()
    assert np.array_equal(got, two)
    # argument errors: a strip thinner than the halo, too few pad rows, in == out
    if n_strips == 5:                                        # 20 rows in 5 strips of 4: thinner than 8 halo rows
        small = [[torch.zeros((pad + 4 + pad, 64, 4), dtype=torch.uint8, device=dev) for _ in range(5)] for _ in range(2)]
        with pytest.raises(_lib.MiError):
            ops.strip_stage(small[0], small[1], 64, 20, devices, pad, _lib.FILTER_GAUSSIAN, 15, 2.5)
    if n_strips > 1:
        with pytest.raises(_lib.MiError):
            ops.strip_stage(bufs[0], bufs[1], w, h, devices, 1, _lib.FILTER_MEDIAN, 5, 0.0)
    with pytest.raises(_lib.MiError) as e:
        ops.strip_stage(bufs[0], bufs[0], w, h, devices, pad, _lib.FILTER_SOBEL, 3, 0.0)
    assert e.value.code == -4
