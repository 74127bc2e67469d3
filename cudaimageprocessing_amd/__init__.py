"""MI355X-native (gfx950) image filters behind the imageTools API of ctaylo41/CudaImageProcessing.

The arithmetic lives in libmi_imagetools.so (hand-written HIP, C ABI in include/mi_imagetools.h);
this package is the Python host side: ctypes bindings (`_lib`), the operator mirror (`ops`),
the synthetic-image generator (`synth`) and batch/strip sharding over torch.distributed (`shard`).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "synth", "shard", "build"]
