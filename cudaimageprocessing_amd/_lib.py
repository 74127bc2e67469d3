"""ctypes binding of libmi_imagetools.so (the C ABI declared in include/mi_imagetools.h).

There is no fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# MI_IMAGETOOLS_LIB points the binding at another build of the library (kernel experiments)
LIB_PATH = os.environ.get("MI_IMAGETOOLS_LIB") or os.path.join(_PKG, "libmi_imagetools.so")

MI_OK = 0
GAUSS_SEPARABLE, GAUSS_DIRECT = 0, 1
FILTER_GRAYSCALE, FILTER_SOBEL, FILTER_GAUSSIAN, FILTER_MEAN, FILTER_MEDIAN, FILTER_GAUSSIAN_DIRECT = range(6)
FFT_FORWARD, FFT_INVERSE = -1, 1

_vp, _i, _f, _sz, _u64, _u32 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64, C.c_uint32

# name -> argtypes; every function returns int.  Kept in step with include/mi_imagetools.h
# (tests/test_abi.py checks the two against each other).
SIGNATURES = {
    "mi_image_load_dev": [_vp, _vp, _sz, _vp],
    "mi_image_write_dev": [_vp, _vp, _i, _i, _vp],
    "mi_grayscale_u8x4_dev": [_vp, _vp, _i, _i, _i, _vp],
    "mi_sobel_u8x4_dev": [_vp, _vp, _i, _i, _i, _vp],
    "mi_gaussian_u8x4_dev": [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "mi_meanblur_u8x4_dev": [_vp, _vp, _i, _i, _i, _i, _vp],
    "mi_medianblur_u8x4_dev": [_vp, _vp, _i, _i, _i, _i, _vp],
    "mi_chain_gray_gauss5_median3_sobel_u8x4_dev": [_vp, _vp, _i, _i, _i, _f, _vp],
    "mi_filter_strip_u8x4_dev": [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    "mi_filter_halo_rows": [_i, _i],
    "mi_synth_u8x4_dev": [_vp, _u64, _u64, _u64, _vp],
    "mi_debug_sobel_mag_table_dev": [_vp, _u32, _vp],
    "mi_debug_sobel_mag_pairs_dev": [_vp, _i, _i, _vp],
    "mi_u8_to_complex_dev": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "mi_fft2d_c2c_f32_dev": [_vp, _vp, _i, _i, _i, _i, _vp],
    "mi_fft2d_workspace_bytes": [_i, _i, _vp],
    "mi_fft2d_c2c_f32_ws_dev": [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp],
    "mi_fftshift_dev": [_vp, _i, _i, _vp],
    "mi_highpass_dev": [_vp, _i, _i, _f, _vp],
    "mi_normalize_zero_imag_dev": [_vp, _i, _i, _vp],
    "mi_complex_to_u8_dev": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "mi_minmax_f32_dev": [_vp, _sz, _vp, _vp],
    "mi_normalize_to_u8_dev": [_vp, _vp, _sz, _vp, _vp],
    "mi_image_load": [_vp, _vp, _sz],
    "mi_image_write": [C.c_char_p, _vp, _i, _i],
    "mi_grayscale_u8x4": [_vp, _vp, _i, _i],
    "mi_sobel_u8x4": [_vp, _vp, _i, _i],
    "mi_gaussian_u8x4": [_vp, _vp, _i, _i, _i, _f, _i],
    "mi_meanblur_u8x4": [_vp, _vp, _i, _i, _i],
    "mi_medianblur_u8x4": [_vp, _vp, _i, _i, _i],
    "mi_run_ops_u8x4": [_vp, _vp, _i, _i, _vp, _i],
    "mi_run_ops_chunked_u8x4": [_vp, _vp, _i, _i, _vp, _i, _i],
    "mi_batch_run_ops_u8x4": [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i],
    "mi_strip_run_ops_u8x4": [_vp, _vp, _i, _i, _vp, _i, _vp, _i],
    "mi_host_release": [],
    "mi_fft_highpass_u8x4": [_vp, _vp, _i, _i, _f],
    "mi_fft_highpass_workspace_bytes": [_i, _i, _vp],
    "mi_fft_highpass_u8x4_dev": [_vp, _vp, _i, _i, _f, _vp, _sz, _vp],
    "mi_fft_highpass_u8x4_static_dev": [_vp, _vp, _i, _i, _f, _vp],
    "mi_fft_highpass_batch_workspace_bytes": [_i, _i, _i, _vp],
    "mi_fft_highpass_batch_u8x4_dev": [_vp, _vp, _i, _i, _i, _f, _vp, _sz, _vp],
    "mi_compress_u8x4": [_vp, _vp, _i, _i, _f],
    "mi_spectrum_keep_top_dev": [_vp, _sz, _sz, _vp],
    "mi_selftest": [],
    "mi_gaussian_kernel": [_i, _f, _vp],
    "mi_strip_rows": [_i, _i, _i, _vp, _vp],
    "mi_strip_stage_u8x4_dev": [_vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _f],
    "mi_debug_set_tuning": [C.c_char_p, _i],
    "mi_debug_get_tuning": [C.c_char_p, _vp],
    "mi_version": [],
    "mi_device_count": [],
}


class Op(C.Structure):
    """struct mi_op of include/mi_imagetools.h"""
    _fields_ = [("filter", C.c_int), ("ksize", C.c_int), ("sigma", C.c_float)]


class MiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libmi_imagetools error {code}: {message}")
        self.code = code


_lib = None


def lib():
    """The loaded library; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m cudaimageprocessing_amd.build` "
                "(hipcc, gfx950).  This package has no CPU fallback.")
        # One HIP runtime per process: torch bundles its own libamdhip64 (soname
        # libamdhip64.so.7, the one this library asks for).  Loaded after torch, the library
        # binds to that copy; loaded before it, the process would end up with two runtimes and
        # the second one finds no device.  torch is this package's device-memory/stream
        # plumbing anyway, so import it first.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = args
            fn.restype = _i
        handle.mi_last_error.argtypes = []
        handle.mi_last_error.restype = C.c_char_p
        _lib = handle
    return _lib


def last_error():
    return lib().mi_last_error().decode(errors="replace")


def check(rc):
    if rc != MI_OK:
        raise MiError(rc, last_error())
