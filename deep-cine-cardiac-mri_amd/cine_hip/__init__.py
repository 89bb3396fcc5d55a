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
def _hw_queue_default() -> None:
    import sys
    import warnings
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"
    torch = sys.modules.get("torch")          # never imports torch itself: the question is whether the runtime is ALREADY up
    try:
        started = bool(torch is not None and torch.cuda.is_initialized())
    except Exception:                          # pragma: no cover
        started = False
    if started:
        warnings.warn("cine_hip was imported after the HIP runtime had been initialised and GPU_MAX_HW_QUEUES is not set: the runtime keeps its "
                      "default of 4 hardware queues, on which the weight-gradient side stream, the U-Net branches and RCCL's streams share queues "
                      "(measured: cfg-2 training step 40.4 -> 45.7 ms).  Export GPU_MAX_HW_QUEUES=16 before the first GPU call, or import "
                      "reconstruction.models / cine_hip first.", RuntimeWarning, stacklevel=3)


_hw_queue_default()

from . import synth  # noqa: F401,E402  (host-only, numpy)

__all__ = ["synth"]
