"""Host-side mirror of the reference's imageTools interface over the C ABI.

Two families, both thin (all arithmetic happens in libmi_imagetools.so):

* device tier -- functions on CUDA(HIP) torch tensors of dtype uint8 and shape (..., H, W, 4);
  asynchronous on torch's current stream.  torch is used only for device memory and streams.
* host tier -- the reference's wrapper names (kernels/imageTools.h:10-18) on numpy arrays,
  blocking, same argument order: imageGrayScaleWrapper(returnImage, imageLoaded, width, height).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


# ---------------------------------------------------------------------------- device tier
def _torch():
    import torch
    return torch


def _stream():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _frames(t):
    torch = _torch()
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError("device-tier ops take CUDA/HIP tensors; there is no CPU fallback")
    if t.dtype != torch.uint8 or t.dim() < 3 or t.shape[-1] != 4 or not t.is_contiguous():
        raise ValueError("expected a contiguous uint8 tensor of shape (..., H, W, 4)")
    h, w = int(t.shape[-3]), int(t.shape[-2])
    batch = 1
    for d in t.shape[:-3]:
        batch *= int(d)
    return w, h, batch


def _out_like(t, out):
    if out is None:
        return _torch().empty_like(t)
    if out.shape != t.shape or out.dtype != t.dtype or not out.is_cuda or not out.is_contiguous():
        raise ValueError("out must match the input's shape, dtype and device")
    return out


def grayscale(img, out=None):
    w, h, b = _frames(img)
    out = _out_like(img, out)
    check(lib().mi_grayscale_u8x4_dev(img.data_ptr(), out.data_ptr(), w, h, b, _stream()))
    return out


def sobel(img, out=None):
    w, h, b = _frames(img)
    out = _out_like(img, out)
    check(lib().mi_sobel_u8x4_dev(img.data_ptr(), out.data_ptr(), w, h, b, _stream()))
    return out


def gaussian_blur(img, kernel_size, sigma, out=None, direct=False):
    w, h, b = _frames(img)
    out = _out_like(img, out)
    mode = _lib.GAUSS_DIRECT if direct else _lib.GAUSS_SEPARABLE
    check(lib().mi_gaussian_u8x4_dev(img.data_ptr(), out.data_ptr(), w, h, b, int(kernel_size),
                                     float(sigma), mode, _stream()))
    return out


def mean_blur(img, kernel_size=3, out=None):
    w, h, b = _frames(img)
    out = _out_like(img, out)
    check(lib().mi_meanblur_u8x4_dev(img.data_ptr(), out.data_ptr(), w, h, b, int(kernel_size), _stream()))
    return out


def median_blur(img, kernel_size=3, out=None):
    w, h, b = _frames(img)
    out = _out_like(img, out)
    check(lib().mi_medianblur_u8x4_dev(img.data_ptr(), out.data_ptr(), w, h, b, int(kernel_size), _stream()))
    return out


def filter_chain(img, bufs=None, gauss_k=5, gauss_sigma=1.0, median_k=3):
    """BASELINE.json config 2: gray -> Gaussian -> median -> Sobel.  Returns the four stage
    outputs; `bufs` (4 tensors like img) lets a benchmark reuse its buffers."""
    torch = _torch()
    if bufs is None:
        bufs = [torch.empty_like(img) for _ in range(4)]
    grayscale(img, bufs[0])
    gaussian_blur(bufs[0], gauss_k, gauss_sigma, bufs[1])
    median_blur(bufs[1], median_k, bufs[2])
    sobel(bufs[2], bufs[3])
    return bufs


def fused_chain(img, gauss_sigma=1.0, out=None):
    """The same chain in one kernel (mi_chain_gray_gauss5_median3_sobel_u8x4_dev): only the final
    Sobel image is produced, bit-identical to filter_chain(img)[3]."""
    w, h, b = _frames(img)
    out = _out_like(img, out)
    check(lib().mi_chain_gray_gauss5_median3_sobel_u8x4_dev(img.data_ptr(), out.data_ptr(), w, h, b,
                                                            float(gauss_sigma), _stream()))
    return out


def filter_strip(filter_id, strip_in, out, w, h_total, in_row0, y_begin, y_end, ksize=3, sigma=0.0):
    """One row strip of a larger image (mi_filter_strip_u8x4_dev): strip_in holds global rows
    [in_row0, in_row0 + strip_in.shape[0]); out receives rows [y_begin, y_end)."""
    in_rows = int(strip_in.shape[0])
    check(lib().mi_filter_strip_u8x4_dev(int(filter_id), strip_in.data_ptr(), out.data_ptr(), int(w),
                                         int(h_total), int(in_row0), in_rows, int(y_begin), int(y_end),
                                         int(ksize), float(sigma), _stream()))
    return out


def halo_rows(filter_id, ksize=3, w=None):
    """Rows of halo a strip needs per side; `w` (pixels) only matters for an image narrower than the radius, whose
    flat-index taps reach ceil(r / w) rows further instead of one."""
    n = lib().mi_filter_halo_rows(int(filter_id), int(ksize))
    if n < 0:
        raise ValueError(f"unknown filter {filter_id}")
    r = int(ksize) // 2
    if w is not None and n > 1 and int(w) < r:
        n = r + -(-r // int(w))
    return n


def set_tuning(name, value):
    """One tuning knob of the library (mi_debug_set_tuning; `name` = its environment variable's name)."""
    check(lib().mi_debug_set_tuning(str(name).encode(), int(value)))


def get_tuning(name):
    """The knob's current value (mi_debug_get_tuning): the default, the environment's, or an earlier set."""
    import ctypes
    v = ctypes.c_int(0)
    check(lib().mi_debug_get_tuning(str(name).encode(), ctypes.byref(v)))
    return int(v.value)


class tuning:
    """Context manager: `with ops.tuning(MI_ROWS_PER_BAND=8): ...` sets the knobs and puts back the
    values they had on entry (so a knob set through the environment survives); `restore` overrides
    what is put back for the knobs it names."""

    def __init__(self, restore=None, **knobs):
        self.knobs, self.restore, self.saved = knobs, restore or {}, {}

    def __enter__(self):
        for k, v in self.knobs.items():
            self.saved[k] = get_tuning(k)
            set_tuning(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.knobs:
            set_tuning(k, self.restore.get(k, self.saved[k]))
        return False


def synth(seed, w, h, batch=None, first_idx=0, device="cuda"):
    """Synthetic frames generated on the device (same stream of pixels as oracle synth)."""
    torch = _torch()
    shape = (h, w, 4) if batch is None else (batch, h, w, 4)
    out = torch.empty(shape, dtype=torch.uint8, device=device)
    count = w * h * (1 if batch is None else batch)
    check(lib().mi_synth_u8x4_dev(out.data_ptr(), int(seed), int(first_idx), count, _stream()))
    return out


def sobel_mag_table(count, device="cuda"):
    torch = _torch()
    out = torch.empty(count, dtype=torch.uint8, device=device)
    check(lib().mi_debug_sobel_mag_table_dev(out.data_ptr(), count, _stream()))
    return out


def sobel_mag_pairs(gmax, variant=0, device="cuda"):
    torch = _torch()
    n = 2 * gmax + 1
    out = torch.empty((n, n), dtype=torch.uint8, device=device)
    check(lib().mi_debug_sobel_mag_pairs_dev(out.data_ptr(), gmax, variant, _stream()))
    return out


def image_load(byte_tensor, out=None):
    torch = _torch()
    n_px = byte_tensor.numel() // 4
    if out is None:
        out = torch.empty(n_px * 4, dtype=torch.uint8, device=byte_tensor.device)
    check(lib().mi_image_load_dev(byte_tensor.data_ptr(), out.data_ptr(), n_px, _stream()))
    return out


def image_write(pixels, out=None):
    torch = _torch()
    w, h, b = _frames(pixels)
    assert b == 1
    if out is None:
        out = torch.empty(w * h * 4, dtype=torch.uint8, device=pixels.device)
    check(lib().mi_image_write_dev(out.data_ptr(), pixels.data_ptr(), w, h, _stream()))
    return out


# FFT path --------------------------------------------------------------------------------
def _plane(t):
    torch = _torch()
    if not (t.is_cuda and t.dtype == torch.complex64 and t.is_contiguous() and t.dim() >= 2):
        raise ValueError("expected a contiguous complex64 CUDA tensor of shape (..., rows, cols)")
    rows, cols = int(t.shape[-2]), int(t.shape[-1])
    batch = 1
    for d in t.shape[:-2]:
        batch *= int(d)
    return rows, cols, batch


def u8_to_complex(img):
    torch = _torch()
    w, h, b = _frames(img)
    assert b == 1
    planes = [torch.empty((h, w), dtype=torch.complex64, device=img.device) for _ in range(3)]
    check(lib().mi_u8_to_complex_dev(img.data_ptr(), planes[0].data_ptr(), planes[1].data_ptr(),
                                     planes[2].data_ptr(), w, h, _stream()))
    return planes


def fft2d_workspace_bytes(rows, cols):
    """Device scratch a rows x cols transform needs (0 for powers of two)."""
    import ctypes
    n = ctypes.c_size_t(0)
    check(lib().mi_fft2d_workspace_bytes(rows, cols, ctypes.byref(n)))
    return int(n.value)


def fft2d(x, inverse=False, out=None, workspace=None):
    """Unnormalised 2-D complex DFT of the last two dims (cufftExecC2C's convention).  Sizes that
    are not powers of two need scratch: `workspace` (a uint8 device tensor of at least
    fft2d_workspace_bytes) or, when None, one allocated here through torch."""
    rows, cols, batch = _plane(x)
    if out is None:
        out = _torch().empty_like(x)
    direction = _lib.FFT_INVERSE if inverse else _lib.FFT_FORWARD
    need = fft2d_workspace_bytes(rows, cols)
    pow2 = (rows & (rows - 1)) == 0 and (cols & (cols - 1)) == 0
    if need == 0 or (pow2 and (out.data_ptr() != x.data_ptr() or workspace is None)):
        # powers of two: scratch is optional; it is used only when the caller brings it (in place, tall planes)
        check(lib().mi_fft2d_c2c_f32_dev(x.data_ptr(), out.data_ptr(), rows, cols, batch, direction, _stream()))
        return out
    if workspace is None:
        workspace = _torch().empty(need, dtype=_torch().uint8, device=x.device)
    check(lib().mi_fft2d_c2c_f32_ws_dev(x.data_ptr(), out.data_ptr(), rows, cols, batch, direction,
                                        workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                        _stream()))
    return out


def fftshift_(x):
    rows, cols, batch = _plane(x)
    assert batch == 1
    check(lib().mi_fftshift_dev(x.data_ptr(), cols, rows, _stream()))
    return x


def highpass_(x, cutoff=10.0):
    rows, cols, batch = _plane(x)
    assert batch == 1
    check(lib().mi_highpass_dev(x.data_ptr(), cols, rows, float(cutoff), _stream()))
    return x


def normalize_zero_imag_(x):
    rows, cols, batch = _plane(x)
    assert batch == 1
    check(lib().mi_normalize_zero_imag_dev(x.data_ptr(), cols, rows, _stream()))
    return x


def complex_to_u8(red, green, blue):
    torch = _torch()
    rows, cols, _ = _plane(red)
    out = torch.empty((rows, cols, 4), dtype=torch.uint8, device=red.device)
    check(lib().mi_complex_to_u8_dev(red.data_ptr(), green.data_ptr(), blue.data_ptr(), out.data_ptr(),
                                     cols, rows, _stream()))
    return out


def minmax(img):
    torch = _torch()
    assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
    out = torch.empty(2, dtype=torch.float32, device=img.device)
    check(lib().mi_minmax_f32_dev(img.data_ptr(), img.numel(), out.data_ptr(), _stream()))
    return out


def normalize_to_u8(img, mm=None):
    torch = _torch()
    if mm is None:
        mm = minmax(img)
    out = torch.empty(tuple(img.shape) + (4,), dtype=torch.uint8, device=img.device)
    check(lib().mi_normalize_to_u8_dev(img.data_ptr(), out.data_ptr(), img.numel(), mm.data_ptr(), _stream()))
    return out


# ---------------------------------------------------------------------------- host tier
def _host_img(a, w, h):
    if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]):
        raise TypeError("host-tier wrappers take C-contiguous numpy uint8 arrays")
    if a.size != w * h * 4:
        raise ValueError(f"array has {a.size} bytes, expected {w * h * 4}")
    return a.ctypes.data


def imageLoadWrapper(image, imageLoaded, imgSize):
    check(lib().mi_image_load(image.ctypes.data, imageLoaded.ctypes.data, int(imgSize)))


def imageWriteWrapper(filename, pixels, width, height):
    check(lib().mi_image_write(str(filename).encode(), _host_img(pixels, width, height), width, height))


def imageGrayScaleWrapper(returnImage, imageLoaded, width, height):
    check(lib().mi_grayscale_u8x4(_host_img(imageLoaded, width, height), _host_img(returnImage, width, height), width, height))


def imageSobelEdgeWrapper(returnImage, imageLoaded, width, height):
    check(lib().mi_sobel_u8x4(_host_img(imageLoaded, width, height), _host_img(returnImage, width, height), width, height))


def imageGaussianBlurWrapper(returnImage, imageLoaded, width, height, kernalSize, sigma, direct=True):
    """The reference's wrapper: by default the reference's own k x k summation order (the same bytes
    as upstream); direct=False takes the separable kernel (within 1 LSB)."""
    check(lib().mi_gaussian_u8x4(_host_img(imageLoaded, width, height), _host_img(returnImage, width, height),
                                 width, height, int(kernalSize), float(sigma),
                                 _lib.GAUSS_DIRECT if direct else _lib.GAUSS_SEPARABLE))


def imageMeanBlurWrapper(returnImage, imageLoaded, width, height, kernalSize=3):
    check(lib().mi_meanblur_u8x4(_host_img(imageLoaded, width, height), _host_img(returnImage, width, height),
                                 width, height, int(kernalSize)))


def imageMedianBlurWrapper(returnImage, imageLoaded, width, height, kernalSize=3):
    check(lib().mi_medianblur_u8x4(_host_img(imageLoaded, width, height), _host_img(returnImage, width, height),
                                   width, height, int(kernalSize)))


def compressImage(outputImage, inputImage, width, height, keep_fraction=0.10):
    """compressImage (kernels/imageTools.h:17; undefined upstream): keep the largest DFT coefficients."""
    check(lib().mi_compress_u8x4(_host_img(inputImage, width, height), _host_img(outputImage, width, height),
                                 width, height, float(keep_fraction)))


def spectrum_keep_top_(x, keep):
    """Zero all but the `keep` largest-magnitude values of a complex64 device tensor, in place."""
    check(lib().mi_spectrum_keep_top_dev(x.data_ptr(), x.numel(), int(keep), _stream()))
    return x


def test():
    """bool test() (kernels/imageTools.h:19; undefined upstream): the library's self-test."""
    return lib().mi_selftest() == _lib.MI_OK


_OP_NAMES = {"gray": _lib.FILTER_GRAYSCALE, "grayscale": _lib.FILTER_GRAYSCALE, "sobel": _lib.FILTER_SOBEL,
             "gauss": _lib.FILTER_GAUSSIAN, "gaussian": _lib.FILTER_GAUSSIAN, "gauss_direct": _lib.FILTER_GAUSSIAN_DIRECT,
             "mean": _lib.FILTER_MEAN, "median": _lib.FILTER_MEDIAN}


def parse_ops(spec):
    """'gray,gauss:5:1.0,median:3,sobel' -> [(filter, ksize, sigma)] (the demo CLI's syntax)."""
    out = []
    for item in spec.split(","):
        parts = item.strip().split(":")
        if parts[0] not in _OP_NAMES:
            raise ValueError(f"unknown op {parts[0]!r} (known: {', '.join(sorted(_OP_NAMES))})")
        f = _OP_NAMES[parts[0]]
        ksize = int(parts[1]) if len(parts) > 1 else (0 if f == _lib.FILTER_GRAYSCALE else 3)
        sigma = float(parts[2]) if len(parts) > 2 else 1.0
        out.append((f, ksize, sigma))
    return out


def run_ops(returnImage, imageLoaded, width, height, ops_list, chunk_rows=None):
    """Several filters on one host image in a single trip over PCIe (mi_run_ops_u8x4).
    ops_list: [(filter, ksize, sigma)] or a parse_ops string."""
    if isinstance(ops_list, str):
        ops_list = parse_ops(ops_list)
    arr = (_lib.Op * len(ops_list))(*[_lib.Op(int(f), int(k), float(s)) for f, k, s in ops_list])
    src, dst = _host_img(imageLoaded, width, height), _host_img(returnImage, width, height)
    if chunk_rows is None:
        check(lib().mi_run_ops_u8x4(src, dst, width, height, arr, len(ops_list)))
    else:
        check(lib().mi_run_ops_chunked_u8x4(src, dst, width, height, arr, len(ops_list), int(chunk_rows)))


def _op_array(ops_list):
    if isinstance(ops_list, str):
        ops_list = parse_ops(ops_list)
    return (_lib.Op * len(ops_list))(*[_lib.Op(int(f), int(k), float(s)) for f, k, s in ops_list]), len(ops_list)


def batch_run_ops(out, frames, ops_list, devices=(0,)):
    """mi_batch_run_ops_u8x4: host frames (n, h, w, 4) sharded by frame over `devices` (one process)."""
    n, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    arr, n_ops = _op_array(ops_list)
    dev = (C.c_int * len(devices))(*devices)
    check(lib().mi_batch_run_ops_u8x4(frames.ctypes.data, out.ctypes.data, w, h, n, arr, n_ops, dev, len(devices)))


def strip_run_ops(out, image, ops_list, devices=(0,)):
    """mi_strip_run_ops_u8x4: one host image (h, w, 4) in row strips over `devices` (one process)."""
    h, w = int(image.shape[0]), int(image.shape[1])
    arr, n_ops = _op_array(ops_list)
    dev = (C.c_int * len(devices))(*devices)
    check(lib().mi_strip_run_ops_u8x4(image.ctypes.data, out.ctypes.data, w, h, arr, n_ops, dev, len(devices)))


def strip_rows(h, n, i):
    """Row range [y0, y1) of strip i of n (mi_strip_rows)."""
    y0, y1 = C.c_int(0), C.c_int(0)
    check(lib().mi_strip_rows(int(h), int(n), int(i), C.byref(y0), C.byref(y1)))
    return y0.value, y1.value


def strip_stage(bufs_in, bufs_out, w, h, devices, pad_rows, filter_id, ksize=3, sigma=0.0):
    """mi_strip_stage_u8x4_dev: one filter over device-resident row strips.  bufs_in[i] / bufs_out[i]: uint8
    tensors of (pad_rows + rows_i + pad_rows, w, 4) on devices[i], the strip's rows starting at row pad_rows."""
    n = len(devices)
    pin = (C.c_void_p * n)(*[b.data_ptr() for b in bufs_in])
    pout = (C.c_void_p * n)(*[b.data_ptr() for b in bufs_out])
    dev = (C.c_int * n)(*devices)
    check(lib().mi_strip_stage_u8x4_dev(pin, pout, int(w), int(h), dev, n, int(pad_rows), int(filter_id), int(ksize), float(sigma)))


def fft_highpass_workspace_bytes(width, height, batch=1):
    n = C.c_size_t(0)
    check(lib().mi_fft_highpass_batch_workspace_bytes(int(width), int(height), int(batch), C.byref(n)))
    return int(n.value)


def fft_highpass(img, cutoff=10.0, out=None, workspace=None):
    """imageFFTImageGenerate on device images, (h, w, 4) or a batch (b, h, w, 4) of uint8: the device tier
    of mi_fft_highpass_u8x4."""
    torch = _torch()
    assert img.is_cuda and img.dtype == torch.uint8 and img.is_contiguous() and img.dim() in (3, 4) and img.shape[-1] == 4
    h, w = int(img.shape[-3]), int(img.shape[-2])
    batch = int(img.shape[0]) if img.dim() == 4 else 1
    if out is None:
        out = torch.empty_like(img)
    if workspace is None:
        workspace = torch.empty(fft_highpass_workspace_bytes(w, h, batch), dtype=torch.uint8, device=img.device)
    check(lib().mi_fft_highpass_batch_u8x4_dev(img.data_ptr(), out.data_ptr(), w, h, batch, float(cutoff),
                                               workspace.data_ptr(), workspace.numel(), _stream()))
    return out


def imageFFTImageGenerate(returnImage, imageLoaded, width, height, cutoff=10.0):
    check(lib().mi_fft_highpass_u8x4(_host_img(imageLoaded, width, height), _host_img(returnImage, width, height),
                                     width, height, float(cutoff)))


def generateGaussianKernal(kernalSize, sigma):
    out = np.empty((kernalSize, kernalSize), np.float32)
    check(lib().mi_gaussian_kernel(int(kernalSize), float(sigma), out.ctypes.data))
    return out
