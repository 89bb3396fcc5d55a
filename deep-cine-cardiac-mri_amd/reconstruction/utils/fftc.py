"""Centered ortho FFTs on (..., 2) tensors, run by the gfx950 line-FFT kernels.

Interface of the reference's utils/fftc.py (fft1c :5, ifft1c :32, fft2c :59,
ifft2c :86, roll :141, fftshift :166, ifftshift :191).  The transforms need GPU
tensors; the shift helpers are index permutations: a HIP kernel on float32 GPU tensors (inside the
path they are folded into the FFT kernels' loads and stores), torch.roll on host tensors.
"""
from typing import List, Optional

import torch

from cine_hip import ops
from cine_hip import autograd as ag


def _check(data: torch.Tensor) -> None:
    if not data.shape[-1] == 2:
        raise ValueError("Tensor does not have separate complex dim.")


def _scale(norm: Optional[str], n: int, inverse: bool) -> float:
    """Factor on top of the ortho kernels for torch.fft's other normalisations (reference passes ``norm`` through:
    fftc.py:23-25, 78; run_inference.py:66 uses norm=None)."""
    if norm == "ortho":
        return 1.0
    if norm is None or norm == "backward":        # forward unscaled, inverse 1/n
        return n ** -0.5 if inverse else n ** 0.5
    if norm == "forward":                          # forward 1/n, inverse unscaled
        return n ** 0.5 if inverse else n ** -0.5
    raise ValueError(f"Invalid normalization mode: {norm!r}")


def _run(data: torch.Tensor, two_d: bool, inverse: bool, norm: Optional[str]) -> torch.Tensor:
    _check(data)
    n = data.shape[-2] * (data.shape[-3] if two_d else 1)
    s = _scale(norm, n, inverse)
    # the reference's callers also hand over CPU tensors (run_inference.py:66): they are staged through the GPU -- the
    # arithmetic still runs in the HIP kernels; without a GPU ops raises (no CPU fallback)
    x = data if data.is_cuda or not torch.cuda.is_available() else data.cuda()
    if ag.needs_grad(x):                           # a tensor that requires grad: the same kernels behind an autograd Function
        out = ag.CenteredFftFn.apply(x, two_d, inverse, s)
        return out if out.device == data.device else out.to(data.device)
    out = ops.fft2c(x, inverse=inverse) if two_d else ops.fft1c(x, inverse=inverse)
    if s != 1.0:
        ops.scale_(out, s)
    return out if out.device == data.device else out.to(data.device)


def fft1c(data: torch.Tensor, norm: Optional[str] = "ortho") -> torch.Tensor:
    return _run(data, False, False, norm)


def ifft1c(data: torch.Tensor, norm: Optional[str] = "ortho") -> torch.Tensor:
    return _run(data, False, True, norm)


def fft2c(data: torch.Tensor, norm: Optional[str] = "ortho") -> torch.Tensor:
    return _run(data, True, False, norm)


def ifft2c(data: torch.Tensor, norm: Optional[str] = "ortho") -> torch.Tensor:
    return _run(data, True, True, norm)


def roll(x: torch.Tensor, shift: List[int], dim: List[int]) -> torch.Tensor:
    if len(shift) != len(dim):
        raise ValueError("len(shift) must match len(dim)")
    if x.is_cuda and x.dtype == torch.float32:                 # csrc/ew_kernels.hip: cine_roll, one launch per rolled dimension
        if ag.needs_grad(x):
            return ag.RollFn.apply(x, tuple(int(s) for s in shift), tuple(int(d) for d in dim))
        return ops.roll(x, [int(s) for s in shift], [int(d) for d in dim])
    return torch.roll(x, shifts=tuple(int(s) for s in shift), dims=tuple(int(d) for d in dim))


def fftshift(x: torch.Tensor, dim: Optional[List[int]] = None) -> torch.Tensor:
    dim = list(range(x.dim())) if dim is None else dim
    return roll(x, [x.shape[d] // 2 for d in dim], dim)


def ifftshift(x: torch.Tensor, dim: Optional[List[int]] = None) -> torch.Tensor:
    dim = list(range(x.dim())) if dim is None else dim
    return roll(x, [(x.shape[d] + 1) // 2 for d in dim], dim)
