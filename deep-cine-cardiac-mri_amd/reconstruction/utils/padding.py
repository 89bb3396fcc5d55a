"""MWCNN padding helpers (interface of the reference's utils/padding.py).

``pad_for_mwcnn`` pads the last two dims to a multiple of 2^n_scales, the extra element of an odd
size going on the LEFT (reference padding.py:26-47); paddings are returned as plain ints (the
reference returns 0-d tensors, which force a host sync when used as slice bounds).  Float32 GPU tensors are
padded by ``cine_pad2d``; the path's own pads are fused into ``cine_mwcnn_pad`` / ``cine_xpd_pack``.
"""
from typing import List, Tuple

import torch
import torch.nn.functional as F

from cine_hip import ops
from cine_hip import autograd as ag


def pad_for_mwcnn(x: torch.Tensor, n_scales: int) -> Tuple[torch.Tensor, List[int]]:
    if x.dim() < 2:
        raise ValueError("Number of dimensions cannot be less than 2")
    m = 2 ** n_scales
    paddings: List[int] = []
    for d in (x.shape[-1], x.shape[-2]):
        n_pad = 0 if d % m == 0 else (d // m + 1) * m - d
        left = n_pad // 2 if (d % 2 == 0 or n_pad == 0) else 1 + n_pad // 2
        paddings += [left, n_pad // 2]
    if x.is_cuda and x.dtype == torch.float32:             # csrc/ew_kernels.hip: cine_pad2d (an autograd Function when x requires grad)
        return (ag.Pad2dFn.apply(x, *paddings) if ag.needs_grad(x) else ops.pad2d(x, *paddings)), paddings
    return F.pad(x, paddings), paddings


def unpad_from_mwcnn(x: torch.Tensor, pad: List[int]) -> torch.Tensor:
    pad = [int(p) for p in pad]
    return x[..., pad[2]:x.shape[-2] - pad[3], pad[0]:x.shape[-1] - pad[1]]
