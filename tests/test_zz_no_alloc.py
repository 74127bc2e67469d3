"""mi_fft_highpass_u8x4_static_dev -- the entry point that allocates nothing (its scratch is a
static device array inside the library) -- against mi_fft_highpass_u8x4_dev with a caller
workspace, through the C ABI: byte-identical output on two calls in a row, the second on a smaller
image (the shared buffer keeps the first call's contents), and a clear error beyond the limit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from cudaimageprocessing_amd import _lib  # noqa: E402


def _both(img):
    lib = _lib.lib()
    h, w = img.shape[:2]
    d_in = torch.from_numpy(img).cuda()
    need = C.c_size_t(0)
    _lib.check(lib.mi_fft_highpass_workspace_bytes(w, h, C.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    ref = torch.empty_like(d_in)
    out = torch.full_like(d_in, 0x5A)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.mi_fft_highpass_u8x4_dev(d_in.data_ptr(), ref.data_ptr(), w, h, 10.0, ws.data_ptr(), need.value, stream))
    _lib.check(lib.mi_fft_highpass_u8x4_static_dev(d_in.data_ptr(), out.data_ptr(), w, h, 10.0, stream))
    torch.cuda.synchronize()
    return ref.cpu().numpy(), out.cpu().numpy()


def test_static_scratch_entry_matches_workspace_entry_twice():
    assert torch.cuda.is_available()
    rng = np.random.default_rng(7)
    # the reference's demo size (chirp-z, 48.8 MB of scratch), then a smaller power-of-two image
    for h, w in ((900, 675), (128, 256), (37, 50)):
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        ref, out = _both(img)
        assert ref.any()
        assert np.array_equal(out, ref), (h, w)


def test_static_scratch_entry_rejects_oversize():
    lib = _lib.lib()
    d = torch.zeros((2048, 2048, 4), dtype=torch.uint8, device="cuda")
    o = torch.empty_like(d)
    rc = lib.mi_fft_highpass_u8x4_static_dev(d.data_ptr(), o.data_ptr(), 2048, 2048, 10.0, None)
    assert rc == -3 and b"static buffer" in lib.mi_last_error()
