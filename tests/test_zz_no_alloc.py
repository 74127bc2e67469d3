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


def test_static_scratch_entry_allocates_nothing_and_captures():
    """The property the entry point exists for.  After the first (address look-up) call: a call on a side
    stream under stream capture records without error -- an allocation, a free or a synchronisation would
    abort the capture -- and the graph replays to the eager result; ten more calls leave the device's free
    memory exactly where it was (hipMemGetInfo)."""
    lib = _lib.lib()
    rng = np.random.default_rng(11)
    for h, w in ((900, 675), (256, 256)):                 # chirp-z and power-of-two
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        d_in = torch.from_numpy(img).cuda()
        eager = torch.empty_like(d_in)
        _lib.check(lib.mi_fft_highpass_u8x4_static_dev(d_in.data_ptr(), eager.data_ptr(), w, h, 10.0,
                                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        replayed = torch.full_like(d_in, 0x33)
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            graph.capture_begin()
            rc = lib.mi_fft_highpass_u8x4_static_dev(d_in.data_ptr(), replayed.data_ptr(), w, h, 10.0,
                                                     C.c_void_p(side.cuda_stream))
            graph.capture_end()
        assert rc == 0, lib.mi_last_error()
        torch.cuda.synchronize()
        assert not torch.equal(replayed, eager)           # capture only records
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(replayed, eager), (h, w)
        out = torch.empty_like(d_in)
        torch.cuda.synchronize()
        free0, total0 = torch.cuda.mem_get_info()
        with torch.cuda.stream(side):
            for _ in range(10):
                _lib.check(lib.mi_fft_highpass_u8x4_static_dev(d_in.data_ptr(), out.data_ptr(), w, h, 10.0,
                                                               C.c_void_p(side.cuda_stream)))
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info() == (free0, total0)
        assert torch.equal(out, eager)
        del graph
