"""ctypes front end of oracle/libgolden.so (see golden.c: TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All images are numpy uint8 arrays of shape (H, W, 4) (RGBA, the reference's uchar4 layout,
kernels/imageTools.cu:107-119).  PARITY UNPINNED -- see the header of golden.c.
"""
import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgolden.so")


def build(force=False):
    src = os.path.join(_HERE, "golden.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgolden.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_STB_SO = os.path.join(_HERE, "libstbdecode.so")
_stb = None


def build_stb(force=False):
    src = os.path.join(_HERE, "stb_decode.c")
    if force or not os.path.exists(_STB_SO) or os.path.getmtime(_STB_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libstbdecode.so"], stdout=subprocess.DEVNULL)
    return _STB_SO


def stb_decode(path):
    """(H, W, 4) uint8 RGBA of an image file, decoded as the reference's loadImage decodes it
    (stbi_load(..., STBI_rgb_alpha), main.cpp:14) with the same stb_image v2.30."""
    global _stb
    if _stb is None:
        _stb = C.CDLL(build_stb())
        _stb.gold_stb_load_rgba.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _stb.gold_stb_load_rgba.restype = C.c_void_p
        _stb.gold_stb_free.argtypes = [C.c_void_p]
        _stb.gold_stb_free.restype = None
    w, h = C.c_int(0), C.c_int(0)
    p = _stb.gold_stb_load_rgba(os.fsencode(path), C.byref(w), C.byref(h))
    if not p:
        raise OSError(f"stb_image cannot decode {path}")
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
    finally:
        _stb.gold_stb_free(p)


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    u8p, f32p, f64p = C.c_void_p, C.c_void_p, C.c_void_p
    i, i64, u64, f = C.c_int, C.c_int64, C.c_uint64, C.c_float
    sig = {
        "gold_synth_u8x4": [u8p, u64, u64, u64],
        "gold_image_load": [u8p, u8p, u64],
        "gold_image_write": [u8p, u8p, i, i],
        "gold_grayscale": [u8p, u8p, i, i, i, i],
        "gold_grayscale_table": [u8p, i],
        "gold_sobel": [u8p, u8p, i, i, i, i],
        "gold_gaussian_kernel": [i, f, f32p],
        "gold_gaussian": [u8p, u8p, i, i, i, f32p, i, i],
        "gold_meanblur": [u8p, u8p, i, i, i, i, i],
        "gold_medianblur": [u8p, u8p, i, i, i, i, i],
        "gold_u8_to_complex": [u8p, f32p, f32p, f32p, i64],
        "gold_fft2d_f64": [f64p, i, i, i],
        "gold_fft2d_c2c": [f32p, f32p, i, i, i],
        "gold_fftshift": [f32p, i, i],
        "gold_highpass": [f32p, i, i, f],
        "gold_normalize_zero_imag": [f32p, i, i],
        "gold_complex_to_u8": [f32p, f32p, f32p, u8p, i64],
        "gold_fft_highpass_image": [u8p, u8p, f32p, i, i, f],
        "gold_minmax": [f32p, i64, f32p, f32p],
        "gold_normalize_to_u8": [f32p, u8p, i64, f, f],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = None
    lib.gold_sobel_mag.argtypes = [C.c_uint32]
    lib.gold_sobel_mag.restype = C.c_int
    lib.gold_highpass_gain.argtypes = [i, i, i, i, f]
    lib.gold_highpass_gain.restype = f
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a):
    return a.ctypes.data


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 4, "expected (H, W, 4) uint8"
    return a


def _bands(h, threads):
    threads = max(1, min(int(threads), h))
    edges = [h * t // threads for t in range(threads + 1)]
    return [(edges[t], edges[t + 1]) for t in range(threads) if edges[t + 1] > edges[t]]


def _run_rows(fn, h, threads):
    """Run fn(y0, y1) over row bands; ctypes releases the GIL so bands run on real cores."""
    bands = _bands(h, threads)
    if len(bands) == 1:
        fn(*bands[0])
        return
    with ThreadPoolExecutor(len(bands)) as ex:
        list(ex.map(lambda b: fn(*b), bands))


def synth(seed, w, h, first_idx=0):
    out = np.empty((h, w, 4), np.uint8)
    lib().gold_synth_u8x4(_p(out), seed, first_idx, w * h)
    return out


def image_load(bytes_, n_px):
    src = np.ascontiguousarray(bytes_, dtype=np.uint8).reshape(-1)
    out = np.empty(n_px * 4, np.uint8)
    lib().gold_image_load(_p(src), _p(out), n_px)
    return out


def image_write(pixels):
    px = _img(pixels)
    out = np.empty(px.size, np.uint8)
    lib().gold_image_write(_p(out), _p(px), px.shape[1], px.shape[0])
    return out


def grayscale(img, threads=1):
    img = _img(img)
    h, w = img.shape[:2]
    out = np.empty_like(img)
    _run_rows(lambda a, b: lib().gold_grayscale(_p(img), _p(out), w, h, a, b), h, threads)
    return out


def grayscale_table(fused=True):
    out = np.empty(1 << 24, np.uint8)
    lib().gold_grayscale_table(_p(out), 1 if fused else 0)
    return out


def sobel(img, threads=1):
    img = _img(img)
    h, w = img.shape[:2]
    out = np.empty_like(img)
    _run_rows(lambda a, b: lib().gold_sobel(_p(img), _p(out), w, h, a, b), h, threads)
    return out


def sobel_mag(s):
    return lib().gold_sobel_mag(int(s))


def gaussian_kernel(k, sigma):
    assert k > 0 and k % 2 == 1
    out = np.empty((k, k), np.float32)
    lib().gold_gaussian_kernel(k, float(sigma), _p(out))
    return out


def gaussian(img, k, sigma, threads=1):
    img = _img(img)
    h, w = img.shape[:2]
    kern = gaussian_kernel(k, sigma)
    out = np.empty_like(img)
    _run_rows(lambda a, b: lib().gold_gaussian(_p(img), _p(out), w, h, k, _p(kern), a, b), h, threads)
    return out


def meanblur(img, k=3, threads=1):
    img = _img(img)
    h, w = img.shape[:2]
    out = np.empty_like(img)
    _run_rows(lambda a, b: lib().gold_meanblur(_p(img), _p(out), w, h, k, a, b), h, threads)
    return out


def medianblur(img, k=3, threads=1):
    img = _img(img)
    h, w = img.shape[:2]
    out = np.empty_like(img)
    _run_rows(lambda a, b: lib().gold_medianblur(_p(img), _p(out), w, h, k, a, b), h, threads)
    return out


def chain(img, threads=1, gauss_k=5, gauss_sigma=1.0, median_k=3):
    """BASELINE.json config 2: gray -> Gaussian -> median -> Sobel, each stage seeing the
    previous stage's zero border exactly as chaining the reference wrappers would."""
    a = grayscale(img, threads)
    b = gaussian(a, gauss_k, gauss_sigma, threads)
    c = medianblur(b, median_k, threads)
    d = sobel(c, threads)
    return a, b, c, d


def u8_to_complex(img):
    img = _img(img)
    n = img.shape[0] * img.shape[1]
    planes = [np.empty((img.shape[0], img.shape[1], 2), np.float32) for _ in range(3)]
    lib().gold_u8_to_complex(_p(img), _p(planes[0]), _p(planes[1]), _p(planes[2]), n)
    return planes


def fft2d_f64(data, sign):
    """data: complex128 (rows, cols); returns the unnormalised 2-D DFT (sign -1 fwd, +1 inv)."""
    d = np.ascontiguousarray(data, dtype=np.complex128).copy()
    rows, cols = d.shape
    lib().gold_fft2d_f64(_p(d), rows, cols, sign)
    return d


def fft2d_c2c(data, sign):
    """data: complex64 (rows, cols); computed in double, rounded once to fp32."""
    d = np.ascontiguousarray(data, dtype=np.complex64)
    rows, cols = d.shape
    out = np.empty_like(d)
    lib().gold_fft2d_c2c(_p(d), _p(out), rows, cols, sign)
    return out


def fftshift(data):
    d = np.ascontiguousarray(data, dtype=np.complex64).copy()
    h, w = d.shape
    lib().gold_fftshift(_p(d), w, h)
    return d


def highpass(data, cutoff):
    d = np.ascontiguousarray(data, dtype=np.complex64).copy()
    h, w = d.shape
    lib().gold_highpass(_p(d), w, h, float(cutoff))
    return d


def highpass_gain(x, y, w, h, cutoff):
    return lib().gold_highpass_gain(x, y, w, h, float(cutoff))


def normalize_zero_imag(data):
    d = np.ascontiguousarray(data, dtype=np.complex64).copy()
    h, w = d.shape
    lib().gold_normalize_zero_imag(_p(d), w, h)
    return d


def complex_to_u8(red, green, blue):
    r = np.ascontiguousarray(red, dtype=np.complex64)
    g = np.ascontiguousarray(green, dtype=np.complex64)
    b = np.ascontiguousarray(blue, dtype=np.complex64)
    h, w = r.shape
    out = np.empty((h, w, 4), np.uint8)
    lib().gold_complex_to_u8(_p(r), _p(g), _p(b), _p(out), h * w)
    return out


def fft_highpass_image(img, cutoff=10.0):
    img = _img(img)
    h, w = img.shape[:2]
    out = np.empty_like(img)
    real = np.empty((3, h, w), np.float32)
    lib().gold_fft_highpass_image(_p(img), _p(out), _p(real), w, h, float(cutoff))
    return out, real


def minmax(img):
    a = np.ascontiguousarray(img, dtype=np.float32)
    mn, mx = C.c_float(), C.c_float()
    lib().gold_minmax(_p(a), a.size, C.addressof(mn), C.addressof(mx))
    return mn.value, mx.value


def normalize_to_u8(img, mn=None, mx=None):
    a = np.ascontiguousarray(img, dtype=np.float32)
    if mn is None or mx is None:
        mn, mx = minmax(a)
    out = np.empty(a.shape + (4,), np.uint8)
    lib().gold_normalize_to_u8(_p(a), _p(out), a.size, mn, mx)
    return out


def compress_image(img, keep_fraction):
    """compressImage (kernels/imageTools.h:17, main.cpp:44; declared, never defined upstream), as
    this build defines it on imageFFTImageGenerate's round-trip scaffold (imageTools.cu:633-791):
    per channel, forward DFT (double), keep the ceil(keep*N) coefficients of largest |X|^2 (ties
    at the threshold all stay), inverse DFT, /N, real part saturated to u8 (gold_complex_to_u8).
    numpy restatement: parity unpinned -- the reference holds no definition and no output."""
    img = _img(img)
    h, w = img.shape[:2]
    n = h * w
    keep = int(min(n, max(1, np.ceil(np.float32(keep_fraction).astype(np.float64) * n))))
    planes = []
    for c in range(3):
        spec = np.fft.fft2(img[..., c].astype(np.float64))
        mag2 = spec.real ** 2 + spec.imag ** 2
        if keep < n:
            thr = np.partition(mag2.ravel(), n - keep)[n - keep]
            spec = np.where(mag2 >= thr, spec, 0.0)
        back = np.fft.ifft2(spec)
        planes.append(back.real.astype(np.float32).astype(np.complex64))
    return complex_to_u8(planes[0], planes[1], planes[2])
