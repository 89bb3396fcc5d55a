"""cine_hip: host-side bindings of the MI355X cine-reconstruction kernels.

The numerics live in ``csrc/*.hip`` (hand-written gfx950 kernels) behind the
C ABI declared in ``include/cine_hip.h``; this package loads
``libcine_hip.so`` with ctypes and passes raw device pointers and the caller's
HIP stream.  There is no CPU fallback: any compute entry point raises
``CineHipError`` when the library or a GPU is missing.
"""
from . import synth  # noqa: F401  (host-only, numpy)

__all__ = ["synth"]
