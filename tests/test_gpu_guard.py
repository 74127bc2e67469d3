"""A fence for the hand-counted waits.  RowFeed (csrc/common.h) replaces the compiler's waits for its
LDS-DMA row bursts by `s_waitcnt vmcnt(K + L)` counted by hand; the count is right only while a row step
issues exactly the vector-memory operations the comment there lists (gauss_mfma_kernel counts the same way
for its 16-row blocks).  libmi_imagetools_guard.so is the same
library with every such wait replaced by vmcnt(0) (always safe).  Every kernel that uses RowFeed must give
the same bytes in both builds on the bench's own 16 x 3840x2160 batch: a compiler upgrade that adds or
drops a memory operation in a row loop shows up here as a difference, not as a rare wrong pixel."""
import ctypes as C
import os

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from cudaimageprocessing_amd import _lib, ops  # noqa: E402
from cudaimageprocessing_amd.build import GUARD_LIB_PATH  # noqa: E402


@pytest.fixture(scope="module")
def guard():
    assert torch.cuda.is_available()
    _lib.lib()
    assert os.path.exists(GUARD_LIB_PATH), f"{GUARD_LIB_PATH} is missing: python -m cudaimageprocessing_amd.build"
    g = C.CDLL(GUARD_LIB_PATH)
    for name in ("mi_sobel_u8x4_dev", "mi_meanblur_u8x4_dev", "mi_medianblur_u8x4_dev", "mi_gaussian_u8x4_dev",
                 "mi_chain_gray_gauss5_median3_sobel_u8x4_dev", "mi_debug_set_tuning"):
        fn = getattr(g, name)
        fn.argtypes = _lib.SIGNATURES[name]
        fn.restype = C.c_int
    return g


def _mean_valu(k):
    """the vector-ALU mean (mean_sep_kernel, a RowFeed user) instead of the matrix-core one, in either library"""
    def call(L, i, o, w, h, b):
        assert L.mi_debug_set_tuning(b"MI_MEAN_MFMA", 0) == 0
        try:
            return L.mi_meanblur_u8x4_dev(i, o, w, h, b, k, None)
        finally:
            L.mi_debug_set_tuning(b"MI_MEAN_MFMA", 11)
    return call


def _gauss_wide(k):
    """the vector-ALU kernel (gauss_wide_kernel, a RowFeed user) instead of the matrix-core one, in either library"""
    def call(L, i, o, w, h, b):
        assert L.mi_debug_set_tuning(b"MI_GAUSS_MFMA", 0) == 0
        try:
            return L.mi_gaussian_u8x4_dev(i, o, w, h, b, k, 0.4 * k, 0, None)
        finally:
            L.mi_debug_set_tuning(b"MI_GAUSS_MFMA", 1)
    return call


# (label, how to call it through a library handle)
CASES = [("sobel", lambda L, i, o, w, h, b: L.mi_sobel_u8x4_dev(i, o, w, h, b, None)),
         ("median3", lambda L, i, o, w, h, b: L.mi_medianblur_u8x4_dev(i, o, w, h, b, 3, None)),
         ("chain", lambda L, i, o, w, h, b: L.mi_chain_gray_gauss5_median3_sobel_u8x4_dev(i, o, w, h, b, 1.0, None))]
CASES += [(f"mean{k}", (lambda k: lambda L, i, o, w, h, b: L.mi_meanblur_u8x4_dev(i, o, w, h, b, k, None))(k)) for k in (3, 5, 7, 9, 11, 13, 15)]
CASES += [(f"gauss{k}", (lambda k: lambda L, i, o, w, h, b: L.mi_gaussian_u8x4_dev(i, o, w, h, b, k, 0.4 * k, 0, None))(k))
          for k in (3, 5, 7, 9, 11, 13, 15)]
CASES += [(f"gausswide{k}", _gauss_wide(k)) for k in (7, 11, 15)]
CASES += [(f"mean{k}", (lambda k: lambda L, i, o, w, h, b: L.mi_meanblur_u8x4_dev(i, o, w, h, b, k, None))(k)) for k in (19, 31)]
CASES += [(f"gauss{k}", (lambda k: lambda L, i, o, w, h, b: L.mi_gaussian_u8x4_dev(i, o, w, h, b, k, 0.2 * k, 0, None))(k)) for k in (17, 23, 31)]
def _strip64(fn):
    """the radius-8 matrix-core kernel on 64-pixel strips (its other instantiation, with its own counted waits)"""
    def call(L, i, o, w, h, b):
        assert L.mi_debug_set_tuning(b"MI_GM_STRIP", 64) == 0
        try:
            return fn(L, i, o, w, h, b)
        finally:
            L.mi_debug_set_tuning(b"MI_GM_STRIP", 128)
    return call


CASES += [("gauss15strip64", _strip64(lambda L, i, o, w, h, b: L.mi_gaussian_u8x4_dev(i, o, w, h, b, 15, 6.0, 0, None))),
          ("mean13strip64", _strip64(lambda L, i, o, w, h, b: L.mi_meanblur_u8x4_dev(i, o, w, h, b, 13, None)))]
CASES += [("mean17", lambda L, i, o, w, h, b: L.mi_meanblur_u8x4_dev(i, o, w, h, b, 17, None)), ("meanvalu11", _mean_valu(11)), ("meanvalu15", _mean_valu(15))]


@pytest.mark.parametrize("label,call", CASES, ids=[c[0] for c in CASES])
def test_counted_waits_equal_full_waits(guard, label, call):
    w, h, b = 3840, 2160, 16
    frames = ops.synth(2, w, h, batch=b)
    prod = torch.full_like(frames, 0x11)
    full = torch.full_like(frames, 0x22)
    assert call(_lib.lib(), frames.data_ptr(), prod.data_ptr(), w, h, b) == 0, _lib.last_error()
    assert call(guard, frames.data_ptr(), full.data_ptr(), w, h, b) == 0
    torch.cuda.synchronize()
    assert torch.equal(prod, full), label
