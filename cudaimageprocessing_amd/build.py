"""Builds libmi_imagetools.so (hand-written HIP for gfx950) in-tree with hipcc."""
import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "libmi_imagetools.so")

SOURCES = ["api.hip", "host.hip", "filters.hip", "spectral.hip", "fft.hip", "gauss_direct.hip", "gauss_mfma.hip", "noalloc.hip", "strips.hip", "selftest.cpp", "shim.cpp", "stb_write.cpp"]
HEADERS = ["common.h", "internal.h", "median5_network.h", "fft_twopass.h",
           os.path.join("..", "..", "include", "mi_imagetools.h"),
           os.path.join("..", "..", "include", "imageTools.h")]

# -ffp-contract=off: the grayscale and direct-Gaussian kernels spell every fused
# multiply-add explicitly so that they match the oracle bit for bit.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-x", "hip"]


# MI_EXTRA_CXXFLAGS: extra compiler flags for experiment builds (tools/make_variant.sh), e.g. -DMI_CHAIN_DBG=1
import shlex  # noqa: E402
FLAGS += shlex.split(os.environ.get("MI_EXTRA_CXXFLAGS", ""))


# Per-file flags.  fft.hip: the SLP vectoriser pairs the butterflies' scalar fp32 operations into v_pk_* ones --
# which issue at the scalar rate on this part -- and pays for it in register shuffles (198 v_mov in the cols
# kernel of the two-pass plan, 128 registers with spills to scratch; without it 85 registers, no spills).
# gauss_direct.hip: the same for the multiply-add chains of the direct Gaussians (see the head of that file).
FILE_FLAGS = {"fft.hip": ["-fno-slp-vectorize"], "gauss_direct.hip": ["-fno-slp-vectorize"]}


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the shared library."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(_CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(_CSRC, os.path.splitext(s)[0] + ".o")
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(s, []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if out and (verbose or p.returncode):
            sys.stderr.write(out.decode(errors="replace"))
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-lz", "-pthread"]
    subprocess.check_call(cmd)
    return LIB_PATH


GUARD_LIB_PATH = os.path.join(_PKG, "libmi_imagetools_guard.so")


def build_guard(force=False):
    """libmi_imagetools_guard.so: the same library with filters.hip and gauss_mfma.hip compiled
    -DMI_ROWFEED_FULL_WAIT=1 (every hand-counted s_waitcnt -- RowFeed in common.h, the block feed of
    gauss_mfma_kernel -- replaced by vmcnt(0)).  Test infrastructure: tests/test_gpu_guard.py holds the product
    library to it bit for bit.  The other objects are shared."""
    lib = build()
    guarded = ["filters.hip", "gauss_mfma.hip"]
    deps = [os.path.join(_CSRC, f) for f in guarded + ["common.h", "internal.h", "median5_network.h"]] + [lib]
    if not force and os.path.exists(GUARD_LIB_PATH) and all(os.path.getmtime(GUARD_LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return GUARD_LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs = []
    for f in guarded:
        obj = os.path.join(_CSRC, os.path.splitext(f)[0] + "_guard.o")
        procs.append(subprocess.Popen([hipcc] + FLAGS + FILE_FLAGS.get(f, []) + ["-DMI_ROWFEED_FULL_WAIT=1", "-c", os.path.join(_CSRC, f), "-o", obj]))
    if any(p.wait() != 0 for p in procs):
        raise RuntimeError("hipcc failed (guard build)")
    objs = [os.path.join(_CSRC, os.path.splitext(s)[0] + ("_guard.o" if s in guarded else ".o")) for s in SOURCES]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", GUARD_LIB_PATH] + objs + ["-lz", "-pthread"])
    return GUARD_LIB_PATH


DEMO_SRC = os.path.normpath(os.path.join(_PKG, "..", "demo", "imgtool.cpp"))
DEMO_BIN = os.path.normpath(os.path.join(_PKG, "..", "demo", "imgtool"))


def build_demo(force=False):
    """demo/imgtool: the C++ demo program, linked against the in-tree library."""
    if not os.path.exists(DEMO_SRC):
        return None
    lib = build()
    if (not force and os.path.exists(DEMO_BIN) and os.path.getmtime(DEMO_BIN) > os.path.getmtime(DEMO_SRC)
            and os.path.getmtime(DEMO_BIN) > os.path.getmtime(lib)):
        return DEMO_BIN
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-x", "c++", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", DEMO_SRC, "-o", DEMO_BIN,
                           "-L" + _PKG, "-lmi_imagetools", "-lz", "-Wl,-rpath,$ORIGIN/../cudaimageprocessing_amd"])
    return DEMO_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_demo(force="--force" in sys.argv))
    print(build_guard(force="--force" in sys.argv))
