"""cine_hip: host-side bindings of the MI355X cine-reconstruction kernels.

The numerics live in ``csrc/*.hip`` (hand-written gfx950 kernels) behind the
C ABI declared in ``include/cine_hip.h``; this package loads
``libcine_hip.so`` with ctypes and passes raw device pointers and the caller's
HIP stream.  There is no CPU fallback: any compute entry point raises
``CineHipError`` when the library or a GPU is missing.
"""
import os as _os

# The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The training path runs its weight gradients on a side
# stream (csrc/grad.h: SideLane) and RCCL brings streams of its own: with four queues the side stream ends up sharing a queue with the
# main one and the overlap is gone (cfg-2 training step 40.4 -> 45.7 ms once a process group exists).  Takes effect only if the runtime
# has not been initialised yet; an explicit setting of the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import synth  # noqa: F401,E402  (host-only, numpy)

__all__ = ["synth"]
