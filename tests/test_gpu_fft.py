"""GPU parity of the FFT demo path (reference kernels/imageTools.cu:506-583, 633-791) against the
CPU oracle.  The transform itself replaces cuFFT (closed source, absent from the reference tree),
so its oracle is the DFT definition evaluated in double precision: parity unpinned by reference
vectors, tolerance 1e-5 relative L2 in fp32 as BASELINE.json states."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from cudaimageprocessing_amd import _lib, ops, synth as S  # noqa: E402

REL_L2 = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    _lib.lib()
    return torch.device("cuda:0")


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def rand_plane(rows, cols, seed, batch=None):
    rng = np.random.default_rng(seed)
    shape = (rows, cols) if batch is None else (batch, rows, cols)
    return (rng.uniform(0, 255, shape) + 1j * rng.uniform(-50, 50, shape)).astype(np.complex64)


SHAPES = [(2, 2), (4, 8), (8, 2), (16, 16), (32, 16), (16, 64), (64, 64), (128, 32), (256, 256),
          (512, 64), (64, 1024), (2048, 16), (16, 4096), (4096, 8), (8, 8192), (8192, 4),
          (16384, 4), (8, 16384), (16384, 64)]


@pytest.mark.parametrize("rows,cols", SHAPES)
def test_fft2d_forward_and_inverse_match_oracle(dev, golden, rows, cols):
    x = rand_plane(rows, cols, rows * 7 + cols)
    xd = torch.from_numpy(x).to(dev)
    fwd = ops.fft2d(xd).cpu().numpy()
    ref = golden.fft2d_f64(x.astype(np.complex128), -1)
    assert rel_l2(fwd, ref) <= REL_L2
    inv = ops.fft2d(xd, inverse=True).cpu().numpy()
    assert rel_l2(inv, golden.fft2d_f64(x.astype(np.complex128), +1)) <= REL_L2
    # in place (tall planes take a different pass structure there) agrees too
    y = xd.clone()
    ops.fft2d(y, out=y)
    assert rel_l2(y.cpu().numpy(), ref) <= REL_L2


ODD_SHAPES = [(3, 5), (2, 7), (6, 10), (15, 17), (100, 64), (64, 100), (31, 1000), (243, 125),
              (900, 675), (2000, 2592 // 2), (4096, 3), (5, 8192), (4095, 6),
              (40, 300), (500, 48),     # padded lengths 128 and 1024: with the rest, every m from 8 to 8192
              (4097, 3), (6, 8191), (5000, 12)]     # m = 16384


@pytest.mark.parametrize("rows,cols", ODD_SHAPES)
def test_fft2d_any_size_matches_dft(dev, golden, rows, cols):
    """Sizes that are not powers of two (the reference's demo image is 675x900, main.cpp:34-45):
    Bluestein path against the double-precision DFT (the oracle's O(n^2) definition for the small
    cases, numpy.fft -- itself checked against the oracle in tests/test_oracle.py -- for the rest)."""
    x = rand_plane(rows, cols, rows * 3 + cols)
    xd = torch.from_numpy(x).to(dev)
    if rows * cols <= 64 * 100:
        ref_f = golden.fft2d_f64(x.astype(np.complex128), -1)
        ref_i = golden.fft2d_f64(x.astype(np.complex128), +1)
    else:
        ref_f = np.fft.fft2(x.astype(np.complex128))
        ref_i = np.fft.ifft2(x.astype(np.complex128)) * (rows * cols)
    assert rel_l2(ops.fft2d(xd).cpu().numpy(), ref_f) <= REL_L2
    assert rel_l2(ops.fft2d(xd, inverse=True).cpu().numpy(), ref_i) <= REL_L2
    y = xd.clone()
    ws = torch.empty(ops.fft2d_workspace_bytes(rows, cols), dtype=torch.uint8, device=dev)
    ops.fft2d(y, out=y, workspace=ws)                                   # in place, caller's workspace
    assert rel_l2(y.cpu().numpy(), ref_f) <= REL_L2


@pytest.mark.parametrize("rows,cols,batch", [(90, 68, 7), (675, 900, 3), (17, 4095, 2), (64, 100, 5), (3, 5, 4), (100, 3, 2)])
def test_fused_chirpz_equals_the_separate_kernels(dev, rows, cols, batch):
    """The one-kernel chirp-z line transform does the operations of the five separate kernels in the
    same order: bit-identical output, forward and inverse, for batches cut into plane groups."""
    x = torch.from_numpy(rand_plane(rows, cols, rows + 7 * cols, batch=batch)).to(dev)
    with ops.tuning(MI_FFT_FUSED_CHIRPZ=0, restore={"MI_FFT_FUSED_CHIRPZ": 1}):
        ref_f, ref_i = ops.fft2d(x).clone(), ops.fft2d(x, inverse=True).clone()
    assert torch.equal(ops.fft2d(x), ref_f)
    assert torch.equal(ops.fft2d(x, inverse=True), ref_i)
    y = x.clone()
    ops.fft2d(y, out=y, workspace=torch.empty(ops.fft2d_workspace_bytes(rows, cols), dtype=torch.uint8, device=dev))
    assert torch.equal(y, ref_f)


def test_fft2d_any_size_batch_roundtrip_and_errors(dev):
    x = rand_plane(90, 68, 11, batch=3)
    xd = torch.from_numpy(x).to(dev)
    f = ops.fft2d(xd)
    for b in range(3):
        assert torch.equal(f[b], ops.fft2d(xd[b].contiguous()))
    back = ops.fft2d(f, inverse=True).cpu().numpy() / (90 * 68)
    assert rel_l2(back, x) <= REL_L2
    assert ops.fft2d_workspace_bytes(64, 128) == 0 and ops.fft2d_workspace_bytes(90, 68) > 0
    with pytest.raises(_lib.MiError):
        ops.fft2d_workspace_bytes(9000, 64)                             # > 8192 and not a power of two
    with pytest.raises(_lib.MiError):
        ops.fft2d_workspace_bytes(16384, 16384)                         # a plane of 2 GiB
    with pytest.raises(_lib.MiError):
        ops.fft2d_workspace_bytes(32768, 2)
    with pytest.raises(_lib.MiError):
        ops.fft2d(xd, workspace=torch.empty(16, dtype=torch.uint8, device=dev))
    with pytest.raises(_lib.MiError):                                   # the no-workspace entry point stays 2^k only
        _lib.check(_lib.lib().mi_fft2d_c2c_f32_dev(xd.data_ptr(), f.data_ptr(), 90, 68, 3, -1, None))


def test_fft_highpass_image_any_size(dev, golden):
    """imageFFTImageGenerate on an image shaped like the reference's demo input, scaled down."""
    w, h = 135, 180
    img = S.pattern_image("natural", w, h, 5)
    out = np.empty_like(img)
    ops.imageFFTImageGenerate(out, img, w, h)
    ref_u8, _ = golden.fft_highpass_image(img, 10.0)
    assert np.abs(out.astype(np.int16) - ref_u8.astype(np.int16)).max() <= 1
    assert (out[..., 3] == 255).all()


def test_fft2d_photo_sized_plane(dev):
    """A 24-megapixel plane (6000 x 4000: neither side a power of two, one above 4096): forward
    transform against numpy.fft in double, then the round trip."""
    rows, cols = 4000, 6000
    x = rand_plane(rows, cols, 99)
    xd = torch.from_numpy(x).to(dev)
    ws = torch.empty(ops.fft2d_workspace_bytes(rows, cols), dtype=torch.uint8, device=dev)
    f = ops.fft2d(xd, workspace=ws)
    ref = np.fft.fft2(x.astype(np.complex128))
    assert rel_l2(f.cpu().numpy(), ref) <= REL_L2
    del ref
    back = ops.fft2d(f, inverse=True, workspace=ws)
    assert rel_l2((back / (rows * cols)).cpu().numpy(), x) <= REL_L2


@pytest.mark.parametrize("w,h", [(4500, 20), (24, 5000)])
def test_fft_highpass_image_wider_than_4096(dev, golden, w, h):
    img = S.pattern_image("natural", w, h, 6)
    out = np.empty_like(img)
    ops.imageFFTImageGenerate(out, img, w, h)
    ref_u8, _ = golden.fft_highpass_image(img, 10.0)
    assert np.abs(out.astype(np.int16) - ref_u8.astype(np.int16)).max() <= 1


@pytest.mark.parametrize("rows,cols,batch", [(1024, 64, 3), (2048, 16, 2), (4096, 8, 1), (1024, 1024, 2)])
def test_fft2d_in_place_with_workspace_takes_the_split_path(dev, rows, cols, batch):
    """Tall planes in place: with one plane of caller scratch the transform runs the same passes as
    out of place, so the two results are bit-identical (without scratch only the oracle bound holds)."""
    x = torch.from_numpy(rand_plane(rows, cols, rows + cols, batch=batch)).to(dev)
    want = ops.fft2d(x)
    assert ops.fft2d_workspace_bytes(rows, cols) == rows * cols * 8
    ws = torch.empty(rows * cols * 8, dtype=torch.uint8, device=dev)
    y = x.clone()
    ops.fft2d(y, out=y, workspace=ws)
    assert torch.equal(y, want)
    back = ops.fft2d(y, inverse=True, out=y, workspace=ws)
    assert rel_l2((back / (rows * cols)).cpu().numpy(), x.cpu().numpy()) <= REL_L2


def test_fft2d_closed_forms(dev):
    n = 64
    imp = torch.zeros((n, n), dtype=torch.complex64, device=dev)
    imp[0, 0] = 1
    assert torch.allclose(ops.fft2d(imp), torch.ones_like(imp))
    const = torch.full((n, n), 3.0, dtype=torch.complex64, device=dev)
    f = ops.fft2d(const).cpu().numpy()
    assert abs(f[0, 0] - 3.0 * n * n) < 1e-2 and np.abs(f).sum() - abs(f[0, 0]) < 1e-1
    xx = np.arange(n)[None, :].repeat(n, 0)
    cosx = torch.from_numpy(np.cos(2 * np.pi * 5 * xx / n).astype(np.complex64)).to(dev)
    f = ops.fft2d(cosx).cpu().numpy()
    assert abs(f[0, 5] - n * n / 2) < 0.05 and abs(f[0, n - 5] - n * n / 2) < 0.05
    mask = np.ones((n, n), bool)
    mask[0, 5] = mask[0, n - 5] = False
    assert np.abs(f[mask]).max() < 0.05


def test_fft2d_batch_and_roundtrip(dev, golden):
    x = rand_plane(128, 256, 3, batch=3)
    xd = torch.from_numpy(x).to(dev)
    f = ops.fft2d(xd)
    for b in range(3):
        assert torch.equal(f[b], ops.fft2d(xd[b].contiguous()))
    back = ops.fft2d(f, inverse=True).cpu().numpy() / (128 * 256)
    assert rel_l2(back, x) <= REL_L2


def test_fft2d_4096_roundtrip_parseval_and_256_subcase(dev, golden):
    """BASELINE.json config 5 size: properties at 4096^2, oracle comparison on a 256^2 sub-case."""
    n = 4096
    gray = ops.synth(5, n, n)[..., 0].to(torch.float32)          # seed 5, one byte per pixel
    x = torch.complex(gray, torch.zeros_like(gray))
    f = ops.fft2d(x)
    # Parseval: sum |X|^2 = N * sum |x|^2
    lhs = float((f.abs().double() ** 2).sum())
    rhs = float((x.abs().double() ** 2).sum()) * n * n
    assert abs(lhs - rhs) / rhs < 1e-5
    assert abs(complex(f[0, 0].item()) - complex(float(gray.double().sum()), 0)) / float(gray.double().sum()) < 1e-5
    back = ops.fft2d(f, inverse=True) / (n * n)
    err = float(torch.linalg.vector_norm((back - x).to(torch.complex128)) / torch.linalg.vector_norm(x.to(torch.complex128)))
    assert err <= REL_L2
    sub = x[:256, :256].contiguous()
    ref = golden.fft2d_f64(sub.cpu().numpy().astype(np.complex128), -1)
    assert rel_l2(ops.fft2d(sub).cpu().numpy(), ref) <= REL_L2


def test_fft2d_4096_batch16_full_spectrum_against_double_precision(dev):
    """BASELINE.json config 5 at full size: the batch-16 launch of 4096 x 4096 planes, every coefficient
    of every plane against numpy.fft.fft2 in double precision (numpy.fft itself is pinned to the oracle's
    DFT in tests/test_oracle.py), forward and back; relative L2 <= 1e-5 per plane."""
    n, batch = 4096, 16
    frames = ops.synth(5, n, n, batch=batch)[..., 0].to(torch.float32)     # seed 5, one byte per pixel
    x = torch.complex(frames, torch.zeros_like(frames))
    del frames
    f = ops.fft2d(x)
    back = ops.fft2d(f, inverse=True)
    worst_f = worst_b = 0.0
    for b in range(batch):
        xb = x[b].real.cpu().numpy().astype(np.float64)
        ref = np.fft.fft2(xb)
        worst_f = max(worst_f, rel_l2(f[b].cpu().numpy(), ref))
        worst_b = max(worst_b, rel_l2(back[b].cpu().numpy() / float(n * n), xb))
        if b == 0:      # a single-plane launch takes the same kernels: the same bits
            assert torch.equal(ops.fft2d(x[0].contiguous()), f[0])
    assert worst_f <= REL_L2 and worst_b <= REL_L2, (worst_f, worst_b)


def test_fft2d_4096_two_pass_plan_orders_groups_and_in_place(dev):
    """4096 x 4096 planes take the two-pass plan (csrc/fft_twopass.h): cols pass then rows pass (DIT), or rows
    pass then cols pass (DIF, MI_FFT_TWOPASS=2 with a workspace).  Both orders, forward and inverse, out of
    place and in place through a workspace, a batch that does not fill its last launch group, against
    numpy.fft in double precision and against the three-pass plan (MI_FFT_TWOPASS=0) it replaces."""
    n = 4096
    x0 = rand_plane(n, n, 77)
    ref_f = np.fft.fft2(x0.astype(np.complex128))
    ref_i = np.fft.ifft2(x0.astype(np.complex128)) * float(n * n)
    xd = torch.from_numpy(x0).to(dev)
    ws = torch.empty(ops.fft2d_workspace_bytes(n, n), dtype=torch.uint8, device=dev)
    assert ws.numel() >= n * n * 8
    assert ops.get_tuning("MI_FFT_TWOPASS") == 1
    results = {}
    for mode in (0, 1, 2):
        with ops.tuning(MI_FFT_TWOPASS=mode):
            f = ops.fft2d(xd)
            assert rel_l2(f.cpu().numpy(), ref_f) <= REL_L2, mode
            assert rel_l2(ops.fft2d(xd, inverse=True).cpu().numpy(), ref_i) <= REL_L2, mode
            y = xd.clone()
            ops.fft2d(y, out=y, workspace=ws)                       # in place: one plane of scratch
            assert rel_l2(y.cpu().numpy(), ref_f) <= REL_L2, mode
            # out of place WITH a workspace: mode 2 runs the rows pass first (DIF order)
            g = torch.empty_like(xd)
            _lib.check(_lib.lib().mi_fft2d_c2c_f32_ws_dev(xd.data_ptr(), g.data_ptr(), n, n, 1, _lib.FFT_FORWARD,
                                                          ws.data_ptr(), ws.numel(), None))
            torch.cuda.synchronize()
            results[mode] = (f, y, g)
    # the two orders are the same arithmetic in a different sequence: close, not necessarily equal
    assert rel_l2(results[2][2].cpu().numpy(), ref_f) <= REL_L2
    assert torch.equal(results[1][0], results[1][1])                # in place == out of place (same kernels)
    # a batch of 3 planes in launch groups of 2 (the last group is partial)
    xb = torch.stack([xd, xd.flip(0), xd * 0.5])
    with ops.tuning(MI_FFT_TWOPASS_GROUP=2):
        fb = ops.fft2d(xb)
    assert torch.equal(fb[0], results[1][0])
    assert torch.equal(fb[1], ops.fft2d(xb[1].contiguous()))
    assert torch.equal(fb[2], ops.fft2d(xb[2].contiguous()))


@pytest.mark.parametrize("rows,cols", [(1024, 1024), (2048, 2048), (1024, 4096), (4096, 1024), (2048, 4096), (4096, 2048), (1024, 2048), (2048, 1024)])
def test_fft2d_two_pass_plan_at_every_size_it_is_built_for(dev, rows, cols):
    """The two-pass plan covers rows, cols in {1024, 2048, 4096} (column tiles of 1024 points, RX = rows / 1024
    rows combined per rows-pass tile; rows = 1024 pairs the cols pass with the ordinary row transform).  Both
    orders, forward and inverse, in place with and -- for 1024 rows -- without a workspace, a batch of 3 in
    partial launch groups, against numpy.fft in double precision and the three-pass plan."""
    x0 = rand_plane(rows, cols, rows + cols)
    ref_f = np.fft.fft2(x0.astype(np.complex128))
    ref_i = np.fft.ifft2(x0.astype(np.complex128)) * float(rows * cols)
    xd = torch.from_numpy(x0).to(dev)
    ws = torch.empty(ops.fft2d_workspace_bytes(rows, cols), dtype=torch.uint8, device=dev)
    out = {}
    for mode in (0, 1, 2):
        with ops.tuning(MI_FFT_TWOPASS=mode):
            f = ops.fft2d(xd)
            assert rel_l2(f.cpu().numpy(), ref_f) <= REL_L2, mode
            assert rel_l2(ops.fft2d(xd, inverse=True).cpu().numpy(), ref_i) <= REL_L2, mode
            y = xd.clone()
            ops.fft2d(y, out=y, workspace=ws)
            assert rel_l2(y.cpu().numpy(), ref_f) <= REL_L2, mode
            g = torch.empty_like(xd)
            _lib.check(_lib.lib().mi_fft2d_c2c_f32_ws_dev(xd.data_ptr(), g.data_ptr(), rows, cols, 1, _lib.FFT_INVERSE,
                                                          ws.data_ptr(), ws.numel(), None))
            torch.cuda.synchronize()
            assert rel_l2(g.cpu().numpy(), ref_i) <= REL_L2, mode
            out[mode] = f
    y = xd.clone()
    ops.fft2d(y, out=y)                                                 # in place, no workspace
    assert rel_l2(y.cpu().numpy(), ref_f) <= REL_L2
    if rows == 1024:
        assert torch.equal(y, out[1])                                   # the cols pass on its own rows: the same kernels
    xb = torch.stack([xd, xd.flip(1), xd * 0.25])
    with ops.tuning(MI_FFT_TWOPASS_GROUP=1):                            # (the knob counts 4096 x 4096 planes)
        fb = ops.fft2d(xb)
    fb8 = ops.fft2d(xb)
    assert torch.equal(fb, fb8) and torch.equal(fb[0], out[1])
    assert rel_l2(fb[1].cpu().numpy(), np.fft.fft2(x0[:, ::-1].astype(np.complex128))) <= REL_L2


def test_fft_highpass_4096_fused_passes_against_the_step_by_step_form(dev):
    """imageFFTImageGenerate on 4096 x 4096 images: u8 -> complex rides on the loads of the forward rows pass, shift x
    gain x shift on the stores of the forward cols pass, /N + saturate + pack on the stores of the inverse rows pass
    (four passes over HBM instead of nine).  Same arithmetic per element -- the gain is the same float expression --
    but the forward transform runs rows-then-cols where the step-by-step form runs cols-then-rows, so the floats
    differ in the last bits and a value that sits on an integer may truncate one level apart: max |delta| 1 on
    fewer than one byte in 10^5.  Device tier (batch of 2), host tier, and the step-by-step spectrum itself."""
    n = 4096
    img = np.stack([S.pattern_image("natural", n, n, 31), S.synth_image(32, n, n)])
    d = torch.from_numpy(img).to(dev)
    ws = torch.empty(ops.fft_highpass_workspace_bytes(n, n, 2), dtype=torch.uint8, device=dev)
    assert ops.get_tuning("MI_FFT_FUSED_SPECTRAL") == 1
    fused = ops.fft_highpass(d, 10.0, None, ws)
    with ops.tuning(MI_FFT_FUSED_SPECTRAL=0):
        steps = ops.fft_highpass(d, 10.0, None, ws)
    diff = (fused.to(torch.int16) - steps.to(torch.int16)).abs()
    assert int(diff.max()) <= 1
    assert int((diff != 0).sum()) <= diff.numel() // 100000
    assert bool((fused[..., 3] == 255).all())
    assert torch.equal(ops.fft_highpass(d[1].contiguous(), 10.0), fused[1])      # a single image: the same kernels
    # host tier (host buffers in and out, in place like the reference's demo)
    host = img[0].copy()
    ops.imageFFTImageGenerate(host, host, n, n)
    assert np.array_equal(host, fused[0].cpu().numpy())
    # independent of both: the spectrum route in double precision on one channel of a crop is out of reach at
    # this size, so the gain itself is pinned instead -- the DC term is removed, the far corner untouched
    flat = torch.full((n, n, 4), 200, dtype=torch.uint8, device=dev)
    out = ops.fft_highpass(flat, 10.0)
    assert int(out[..., :3].max()) == 0 and bool((out[..., 3] == 255).all())      # a constant image is all DC


def test_fft_highpass_2048_fused_passes_against_the_step_by_step_form_and_the_oracle_gain(dev, golden):
    """The same fusion at 2048 x 2048 (RX = 2), where one channel is still in reach of a double-precision check:
    numpy fft2 -> the oracle's gain table -> ifft2 -> truncate, against channel 0 of the fused output."""
    n = 2048
    img = np.stack([S.pattern_image("natural", n, n, 41), S.synth_image(42, n, n), S.pattern_image("checker", n, n, 43)])
    d = torch.from_numpy(img).to(dev)
    ws = torch.empty(ops.fft_highpass_workspace_bytes(n, n, 3), dtype=torch.uint8, device=dev)
    fused = ops.fft_highpass(d, 10.0, None, ws)
    with ops.tuning(MI_FFT_FUSED_SPECTRAL=0):
        steps = ops.fft_highpass(d, 10.0, None, ws)
    diff = (fused.to(torch.int16) - steps.to(torch.int16)).abs()
    assert int(diff.max()) <= 1
    assert int((diff != 0).sum()) <= diff.numel() // 100000
    assert bool((fused[..., 3] == 255).all())
    host = img[1].copy()
    ops.imageFFTImageGenerate(host, host, n, n)
    assert np.array_equal(host, fused[1].cpu().numpy())
    # double precision, one channel: gain in the shifted frame = distance from the centre (imageTools.cu:703-719)
    x = img[0][..., 0].astype(np.float64)
    yy, xx = np.meshgrid(np.arange(n) - n // 2, np.arange(n) - n // 2, indexing="ij")
    gain = 1.0 - np.exp(-(xx * xx + yy * yy) / (2.0 * 10.0 * 10.0))
    spec = np.fft.fftshift(np.fft.fft2(x)) * gain
    back = np.fft.ifft2(np.fft.ifftshift(spec)).real
    want = np.clip(np.trunc(back), 0, 255)
    got = fused[0, ..., 0].cpu().numpy().astype(np.float64)
    near_integer = np.abs(back - np.round(back)) < 2e-3                  # fp32 pipeline: ties at integers may fall either way
    assert np.abs(got - want)[~near_integer].max() == 0
    assert np.abs(got - want).max() <= 1


def test_pointwise_stages_match_oracle(dev, golden):
    img = S.synth_image(4, 96, 64)
    planes = ops.u8_to_complex(torch.from_numpy(img).to(dev))
    gold_planes = golden.u8_to_complex(img)
    for c in range(3):
        p = planes[c].cpu().numpy()
        assert np.array_equal(p.real, gold_planes[c][..., 0]) and (p.imag == 0).all()
    for (h, w) in ((64, 96), (7, 9), (8, 5)):
        x = rand_plane(h, w, h + w)
        xd = torch.from_numpy(x).to(dev)
        assert np.array_equal(ops.fftshift_(xd.clone()).cpu().numpy(), golden.fftshift(x))
        hp = ops.highpass_(xd.clone(), 10.0).cpu().numpy()
        assert rel_l2(hp, golden.highpass(x, 10.0)) <= REL_L2          # device expf vs libm expf
        nz = ops.normalize_zero_imag_(xd.clone()).cpu().numpy()
        assert np.allclose(nz, golden.normalize_zero_imag(x), rtol=2e-7, atol=0) and (nz.imag == 0).all()
    vals = np.array([[-5.0, 0.0, 0.9, 1.0, 254.9, 255.0, 300.0, np.nan]], np.float32).astype(np.complex64)
    vd = torch.from_numpy(vals).to(dev)
    q = ops.complex_to_u8(vd, vd, vd).cpu().numpy()
    assert np.array_equal(q, golden.complex_to_u8(vals, vals, vals))


def test_minmax_and_normalize(dev, golden):
    rng = np.random.default_rng(9)
    x = rng.normal(10, 100, (257, 131)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    mm = ops.minmax(xd).cpu().numpy()
    assert tuple(mm) == golden.minmax(x)
    q = ops.normalize_to_u8(xd).cpu().numpy()
    ref = golden.normalize_to_u8(x)
    assert np.abs(q.astype(np.int16) - ref.astype(np.int16)).max() <= 1   # fp32 division on device vs host
    assert (q[..., 3] == 255).all()
    neg = torch.from_numpy(np.array([-3.0, -1.0, -2.0], np.float32)).to(dev)
    assert ops.minmax(neg).cpu().numpy().tolist() == [-3.0, -1.0]


@pytest.mark.parametrize("w,h", [(67, 45), (64, 48), (33, 40), (128, 128), (5, 9)])
def test_host_tier_highpass_equals_the_device_tier_steps(dev, w, h):
    """mi_fft_highpass_u8x4 fuses shift/filter/shift into one pass over the three planes and the
    normalisation into the final cast: byte for byte what the reference's sequence of kernels
    (imageTools.cu:659-781), run step by step through the device tier, produces -- odd sizes,
    whose middle row and column the shift leaves alone, included."""
    img = S.synth_image(w * 131 + h, w, h)
    out = np.empty_like(img)
    ops.imageFFTImageGenerate(out, img, w, h)
    planes = ops.u8_to_complex(torch.from_numpy(img).to(dev))
    ws = torch.empty(max(16, ops.fft2d_workspace_bytes(w, h)), dtype=torch.uint8, device=dev)
    for c in range(3):
        p, q = planes[c], planes[c].view(w, h)                          # cufftPlan2d(width, height)
        spec = ops.fft2d(q, workspace=ws)
        s2 = spec.view(h, w)
        ops.fftshift_(s2); ops.highpass_(s2, 10.0); ops.fftshift_(s2)
        ops.fft2d(spec, inverse=True, out=q, workspace=ws)
        ops.normalize_zero_imag_(p)
    assert np.array_equal(ops.complex_to_u8(*planes).cpu().numpy(), out)


@pytest.mark.parametrize("w,h", [(67, 45), (128, 64), (33, 40), (675, 900)])
def test_device_tier_highpass_equals_the_host_tier_and_replays_from_a_graph(dev, w, h):
    img = S.synth_image(3 * w + h, w, h)
    want = np.empty_like(img)
    ops.imageFFTImageGenerate(want, img, w, h)
    d = torch.from_numpy(img).to(dev)
    ws = torch.empty(ops.fft_highpass_workspace_bytes(w, h), dtype=torch.uint8, device=dev)
    out = torch.empty_like(d)
    assert np.array_equal(ops.fft_highpass(d, 10.0, out, ws).cpu().numpy(), want)
    inplace = d.clone()
    ops.fft_highpass(inplace, 10.0, inplace, ws)
    assert np.array_equal(inplace.cpu().numpy(), want)
    out.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.fft_highpass(d, 10.0, out, ws)
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    with pytest.raises(_lib.MiError):
        ops.fft_highpass(d, 10.0, out, ws[:1024])
    with pytest.raises(_lib.MiError):
        ops.fft_highpass(d, 0.0, out, ws)


@pytest.mark.parametrize("w,h,batch", [(67, 45, 5), (128, 64, 3), (90, 33, 8), (675, 900, 2)])
def test_device_tier_highpass_batch_equals_single_images(dev, w, h, batch):
    frames = torch.from_numpy(np.stack([S.synth_image(50 + 7 * k + w, w, h) for k in range(batch)])).to(dev)
    got = ops.fft_highpass(frames)
    for k in range(batch):
        assert torch.equal(got[k], ops.fft_highpass(frames[k].contiguous())), k
    inplace = frames.clone()
    ops.fft_highpass(inplace, out=inplace)
    assert torch.equal(inplace, got)
    with pytest.raises(_lib.MiError):
        ops.fft_highpass(frames, workspace=torch.empty(ops.fft_highpass_workspace_bytes(w, h, 1), dtype=torch.uint8, device=dev))


def test_host_tier_chirp_tables_survive_only_the_same_geometry(dev):
    """The host tier keeps the chirp-z tables in its cached device buffer between calls: results must
    not depend on what ran before (other sizes, compressImage's transposed geometry, a power of two)."""
    def run(fn, w, h):
        img = S.synth_image(w + h, w, h)
        out = np.empty_like(img)
        fn(out, img, w, h)
        return out
    first = {(w, h): run(ops.imageFFTImageGenerate, w, h) for (w, h) in ((67, 45), (45, 67), (90, 33))}
    comp = run(ops.compressImage, 67, 45)
    order = [(67, 45), (67, 45), (45, 67), (128, 64), (67, 45), (90, 33), (1024, 1024), (90, 33), (45, 67)]
    for (w, h) in order:
        if (w, h) in first:
            assert np.array_equal(run(ops.imageFFTImageGenerate, w, h), first[(w, h)]), (w, h)
        else:
            run(ops.imageFFTImageGenerate, w, h)
        assert np.array_equal(run(ops.compressImage, 67, 45), comp)


def test_fft_highpass_image_pipeline(dev, golden):
    """imageFFTImageGenerate (imageTools.cu:633-791) end to end on a power-of-two image."""
    w, h = 128, 64
    img = S.pattern_image("natural", w, h, 3)
    out = np.empty_like(img)
    ops.imageFFTImageGenerate(out, img, w, h)
    ref_u8, ref_real = golden.fft_highpass_image(img, 10.0)
    # judged on float data (the uchar cast of negative values is undefined upstream): rebuild the
    # real planes on the device and compare by relative L2
    planes = ops.u8_to_complex(torch.from_numpy(img).to(dev))
    for c in range(3):
        # the reference's plan is cufftPlan2d(width, height): the transform sees the buffer as w rows of h
        # points, the shift and the filter as h rows of w (imageTools.cu:688 against :535, :569)
        p, q = planes[c], planes[c].view(w, h)
        ops.fft2d(q, out=q)
        ops.fftshift_(p); ops.highpass_(p, 10.0); ops.fftshift_(p)
        ops.fft2d(q, inverse=True, out=q)
        ops.normalize_zero_imag_(p)
        assert rel_l2(p.cpu().numpy().real, ref_real[c]) <= 1e-4      # high-passed signal is small: abs err matters
        assert np.abs(p.cpu().numpy().real - ref_real[c]).max() < 2e-3
    assert np.abs(out.astype(np.int16) - ref_u8.astype(np.int16)).max() <= 1
    assert (out[..., 3] == 255).all()


def test_first_fft_calls_on_two_streams_of_a_fresh_process(dev):
    """The twiddle table is filled by the first FFT call on a device; a first call on a second stream
    must not see a half-written table.  Needs a process that has not transformed anything yet."""
    import subprocess
    import sys
    code = r"""
import numpy as np, torch
from cudaimageprocessing_amd import ops
rng = np.random.default_rng(3)
x = (rng.standard_normal((2, 8, 1024, 2048)) + 1j * rng.standard_normal((2, 8, 1024, 2048))).astype(np.complex64)
xd = [torch.from_numpy(x[i]).cuda() for i in range(2)]
torch.cuda.synchronize()
s = [torch.cuda.Stream(), torch.cuda.Stream()]
out = [None, None]
for i in range(2):
    with torch.cuda.stream(s[i]):
        out[i] = ops.fft2d(xd[i])
torch.cuda.synchronize()
for i in range(2):
    ref = np.fft.fft2(x[i].astype(np.complex128))
    err = np.linalg.norm(out[i].cpu().numpy() - ref) / np.linalg.norm(ref)
    assert err <= 1e-5, (i, err)
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=110, cwd=root)
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout + p.stderr


def test_second_device_has_its_own_state(dev):
    """Host-tier buffers, the twiddle table and the > 64 KiB LDS attribute are per device."""
    if _lib.lib().mi_device_count() < 2:
        pytest.skip("one GPU on this box")
    with torch.cuda.device(1):
        x = torch.from_numpy(rand_plane(4096, 64, 9)).to("cuda:1")
        f = ops.fft2d(x)
        ref = np.fft.fft2(x.cpu().numpy().astype(np.complex128))
        assert rel_l2(f.cpu().numpy(), ref) <= REL_L2
        img = np.random.default_rng(1).integers(0, 256, (64, 96, 4), dtype=np.uint8)
        a, b = np.empty_like(img), np.empty_like(img)
        ops.imageSobelEdgeWrapper(a, img, 96, 64)
        ops.imageFFTImageGenerate(b, img, 96, 64)
    c, e = np.empty_like(img), np.empty_like(img)
    ops.imageSobelEdgeWrapper(c, img, 96, 64)
    ops.imageFFTImageGenerate(e, img, 96, 64)
    assert np.array_equal(a, c) and np.array_equal(b, e)

