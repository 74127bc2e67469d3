"""The built library's kernels, read from the code objects inside libmi_imagetools.so (no GPU needed): the kernels whose
waits on memory are counted by hand -- LDS-DMA row feeds, the matrix-core filters, the fused chain -- must not spill.
A spill's scratch loads and stores travel on the same counter as the row loads (vmcnt): in-order return keeps the
counted waits correct (the N youngest operations may be outstanding, anything older has landed), but every such wait
then also waits for spills, and a kernel tuned to two or three waves per SIMD has no registers to give.  This is the
fence for compiler upgrades that tests/test_gpu_guard.py is for the waits themselves."""
import os
import re
import shutil
import subprocess

import pytest

from cudaimageprocessing_amd import _lib

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels_of(lib_path, tmp_path):
    """{mangled kernel name: (scratch bytes per lane, vgprs, agprs, lds bytes)} over every code object in the library"""
    fat = tmp_path / "fat.bin"
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib_path], check=True)
    data = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    out = {}
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(data)])):
        piece, co = tmp_path / f"bundle{n}.bin", tmp_path / f"dev{n}.co"
        piece.write_bytes(data[a:b])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={piece}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == "agpr_count" and cur.get("name"):       # a new kernel record starts with .agpr_count (keys are sorted)
                cur = {}
            cur[k] = v
            if k == "vgpr_count" and "name" in cur:
                out[cur["name"]] = (int(cur.get("private_segment_fixed_size", -1)), int(v), int(cur.get("agpr_count", 0)),
                                    int(cur.get("group_segment_fixed_size", -1)))
    return out


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not all(os.path.exists(f"{LLVM}/{t}") for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        pytest.skip("ROCm LLVM tools not found")
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    return kernels_of(_lib.LIB_PATH, tmp_path_factory.mktemp("co"))


def test_every_filter_kernel_is_in_the_library(kernels):
    names = " ".join(kernels)
    for part in ("chain_kernel", "gauss_mfma_kernel", "gauss_mfma_wide_kernel", "median5_kernel", "mean_prefix_kernel", "gauss_wide_kernel",
                 "fft2p_rows_kernel", "fft2p_cols_kernel"):
        assert part in names, part


def test_kernels_with_counted_waits_do_not_spill(kernels):
    # known to spill since rounds 1-2, and fallbacks since the matrix-core kernels took their sizes over (unaligned images,
    # or the knobs that switch those off): the 15x15 kernels on the vector ALUs and round 1's fp32 13x13 / 15x15
    known = re.compile(r"mean_sep_kernelILi7E|gauss_wide_kernelILi7E|gauss_sep_kernelILi[67]E")
    fenced = re.compile(r"chain_kernel|gauss_mfma|sobel_kernel|mean3_kernel|median3_kernel|median5_kernel|gauss_sep_kernel|"
                        r"mean_sep_kernel|gauss_wide_kernel|mean_prefix_kernel|gauss_direct")
    bad = {n: v for n, v in kernels.items() if fenced.search(n) and not known.search(n) and v[0] != 0}
    assert not bad, bad


def test_matrix_core_kernels_keep_their_occupancy(kernels):
    """Two waves per SIMD need at most 256 registers a wave, eight waves a CU at most 20 KiB of LDS each."""
    for n, (scratch, vgpr, agpr, lds) in kernels.items():
        if "gauss_mfma_wide_kernel" in n or re.search(r"gauss_mfma_kernelILi\dELi8E", n):
            assert scratch == 0 and vgpr + agpr <= 256 and lds <= 20480, (n, scratch, vgpr, agpr, lds)
        if re.search(r"gauss_mfma_kernelILi\dELi4E", n):
            assert scratch == 0 and vgpr + agpr <= 168 and lds <= 10240, (n, scratch, vgpr, agpr, lds)
