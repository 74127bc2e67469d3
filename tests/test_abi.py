"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol that
include/mi_imagetools.h declares, the ctypes table matches the header, argument errors are
reported without touching a GPU, and the host-only entry points (Gaussian weights, PNG writer)
work.  No kernel is launched here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi_imagetools.h")


@pytest.fixture(scope="module")
def mi():
    from cudaimageprocessing_amd import _lib, build
    build.build()
    return _lib


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = re.findall(r"(?:int|const char \*)\s*(mi_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S)
    return {name: args for name, args in decls}


def test_header_symbols_are_exported(mi):
    decl = declared_functions()
    assert len(decl) >= 30
    handle = C.CDLL(mi.LIB_PATH)
    for name in decl:
        assert hasattr(handle, name), f"{name} declared in mi_imagetools.h but not exported"
    out = subprocess.check_output(["nm", "-D", "--defined-only", mi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (mi_\w+)", out))
    assert set(decl) <= exported


def test_ctypes_table_matches_header(mi):
    decl = declared_functions()
    table = dict(mi.SIGNATURES)
    assert set(table) | {"mi_last_error"} == set(decl)
    for name, args in decl.items():
        if name == "mi_last_error":
            continue
        n_args = 0 if args.strip() in ("", "void") else len(args.split(","))
        assert n_args == len(table[name]), name


def test_cpp_shim_symbols_exported(mi):
    out = subprocess.check_output(["nm", "-D", "-C", "--defined-only", mi.LIB_PATH]).decode()
    for name in ("imageLoadWrapper", "imageWriteWrapper", "imageGrayScaleWrapper", "imageSobelEdgeWrapper",
                 "imageGaussianBlurWrapper", "imageMeanBlurWrapper", "imageFFTImageGenerate",
                 "generateGaussianKernal", "imageMedianBlurWrapper"):
        assert re.search(rf" T {name}\(", out), name


def test_argument_errors_need_no_gpu(mi):
    lib = mi.lib()
    buf = np.zeros(64 * 4, np.uint8)
    p = buf.ctypes.data
    assert lib.mi_grayscale_u8x4_dev(None, p, 4, 4, 1, None) == -1
    assert "null" in mi.last_error()
    assert lib.mi_sobel_u8x4_dev(p, p + 1024, 2, 8, 1, None) == -1            # W < 3
    assert lib.mi_sobel_u8x4_dev(p, p, 4, 4, 1, None) == -4                   # in/out overlap
    assert lib.mi_gaussian_u8x4_dev(p, p + 1024, 4, 4, 1, 4, 1.0, 0, None) == -2   # even size
    assert lib.mi_gaussian_u8x4_dev(p, p + 1024, 4, 4, 1, 257, 1.0, 0, None) == -2  # too large (255 is the limit)
    assert lib.mi_gaussian_u8x4_dev(p, p + 1024, 4, 4, 1, 5, 0.0, 0, None) == -1   # sigma
    assert lib.mi_gaussian_u8x4_dev(p, p + 1024, 4, 4, 1, 5, 1.0, 7, None) == -1   # mode
    assert lib.mi_meanblur_u8x4_dev(p, p + 1024, 4, 4, 1, 0, None) == -2
    assert lib.mi_medianblur_u8x4_dev(p, p + 1024, 4, 4, 1, 7, None) == -2
    assert lib.mi_grayscale_u8x4_dev(p, p + 1024, 4, 4, 0, None) == -1
    assert lib.mi_fft2d_c2c_f32_dev(p, p, 12, 16, 1, -1, None) == -3          # not a power of two
    assert lib.mi_fft2d_c2c_f32_dev(p, p, 16, 16, 1, 0, None) == -1
    # strip: halo not covered
    assert lib.mi_filter_strip_u8x4_dev(mi.FILTER_SOBEL, p, p + 4096, 4, 16, 4, 4, 4, 8, 3, 0.0, None) == -1
    assert lib.mi_filter_halo_rows(mi.FILTER_GRAYSCALE, 0) == 0
    assert lib.mi_filter_halo_rows(mi.FILTER_SOBEL, 3) == 1
    assert lib.mi_filter_halo_rows(mi.FILTER_MEDIAN, 5) == 3
    assert lib.mi_filter_halo_rows(mi.FILTER_GAUSSIAN, 15) == 8
    assert lib.mi_version() >= 100


def test_gaussian_kernel_matches_oracle(mi, golden):
    from cudaimageprocessing_amd import ops
    for k, sigma in ((3, 7.0), (5, 1.0), (15, 2.5), (1, 0.5), (31, 6.0)):
        assert np.array_equal(ops.generateGaussianKernal(k, sigma), golden.gaussian_kernel(k, sigma))
    with pytest.raises(mi.MiError):
        ops.generateGaussianKernal(4, 1.0)


def test_png_writer(mi, tmp_path):
    from PIL import Image
    from cudaimageprocessing_amd import ops, synth
    img = synth.synth_image(9, 33, 21)
    img[..., 3] = img[..., 0]                    # alpha is kept (4-channel PNG, imageTools.cu:152)
    path = tmp_path / "out.jpg"                  # the reference writes PNG bytes whatever the name
    ops.imageWriteWrapper(path, img, 33, 21)
    assert open(path, "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    back = np.asarray(Image.open(path))
    assert np.array_equal(back, img)


def test_missing_gpu_fails_loudly(mi):
    """Without a GPU the host tier must report a HIP error, never fall back to CPU code."""
    if mi.lib().mi_device_count() > 0:
        pytest.skip("a GPU is present")
    a = np.zeros((4, 4, 4), np.uint8)
    b = np.zeros_like(a)
    rc = mi.lib().mi_grayscale_u8x4(a.ctypes.data, b.ctypes.data, 4, 4)
    assert rc > 0 and "HIP error" in mi.last_error()


def test_imgtool_decodes_jpeg_like_the_committed_stb_fixture(tmp_path):
    """demo/imgtool --decode is stb_image v2.30's decoder (what the reference's loadImage calls, main.cpp:14);
    tests/golden/lord_decoded_by_stb.png was made with it.  Needs no GPU."""
    import subprocess
    Image = pytest.importorskip("PIL.Image")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "demo", "imgtool")
    if not os.path.exists(tool):
        pytest.skip("demo/imgtool not built")
    dst = str(tmp_path / "lord.png")
    p = subprocess.run([tool, "--decode", os.path.join(root, "tests", "golden", "lord.jpeg"), dst], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stdout + p.stderr
    got = np.asarray(Image.open(dst).convert("RGB"))
    want = np.asarray(Image.open(os.path.join(root, "tests", "golden", "lord_decoded_by_stb.png")).convert("RGB"))
    assert np.array_equal(got, want)



def test_image_write_reproduces_the_reference_file_byte_for_byte(tmp_path):
    """imageWriteWrapper ends in stbi_write_png (imageTools.cu:152, stb_image_write v1.16); mi_image_write calls
    the same encoder.  The reference ships one file its wrapper wrote: image/output.jpg (a PNG), committed as
    tests/golden/upstream_output_of_lord.png.  Decoded and written again through the C ABI, it comes out as
    the same 829 886 bytes -- the write path pinned by a reference-held artefact, no GPU involved.
    The zlib encoder (MI_PNG_ZLIB=1) writes another file with the same pixels."""
    import numpy as np
    from PIL import Image
    from cudaimageprocessing_amd import ops
    up = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "upstream_output_of_lord.png")
    px = np.ascontiguousarray(np.asarray(Image.open(up).convert("RGBA")))
    h, w = px.shape[:2]
    out = str(tmp_path / "again.png")
    ops.imageWriteWrapper(out, px, w, h)
    assert open(out, "rb").read() == open(up, "rb").read()
    with ops.tuning(MI_PNG_ZLIB=1):
        ops.imageWriteWrapper(out, px, w, h)
    assert open(out, "rb").read() != open(up, "rb").read()
    assert np.array_equal(np.asarray(Image.open(out).convert("RGBA")), px)
