"""Centered ortho FFTs on (..., 2) tensors, run by the gfx950 line-FFT kernels.

Interface of the reference's utils/fftc.py (fft1c :5, ifft1c :32, fft2c :59,
ifft2c :86, roll :141, fftshift :166, ifftshift :191).  The transforms need GPU
tensors; the shift helpers are pure index permutations and work anywhere.
"""
from typing import List, Optional

import torch

from cine_hip import ops


def _check(data: torch.Tensor) -> None:
    if not data.shape[-1] == 2:
        raise ValueError("Tensor does not have separate complex dim.")


def fft1c(data: torch.Tensor, norm: str = "ortho") -> torch.Tensor:
    _check(data); _ortho(norm)
    return ops.fft1c(data, inverse=False)


def ifft1c(data: torch.Tensor, norm: str = "ortho") -> torch.Tensor:
    _check(data); _ortho(norm)
    return ops.fft1c(data, inverse=True)


def fft2c(data: torch.Tensor, norm: str = "ortho") -> torch.Tensor:
    _check(data); _ortho(norm)
    return ops.fft2c(data, inverse=False)


def ifft2c(data: torch.Tensor, norm: str = "ortho") -> torch.Tensor:
    _check(data); _ortho(norm)
    return ops.fft2c(data, inverse=True)


def _ortho(norm: str) -> None:
    if norm != "ortho":
        raise NotImplementedError("only norm='ortho' (the mode every reference call site uses) is implemented")


def roll(x: torch.Tensor, shift: List[int], dim: List[int]) -> torch.Tensor:
    if len(shift) != len(dim):
        raise ValueError("len(shift) must match len(dim)")
    return torch.roll(x, shifts=tuple(int(s) for s in shift), dims=tuple(int(d) for d in dim))


def fftshift(x: torch.Tensor, dim: Optional[List[int]] = None) -> torch.Tensor:
    dim = list(range(x.dim())) if dim is None else dim
    return roll(x, [x.shape[d] // 2 for d in dim], dim)


def ifftshift(x: torch.Tensor, dim: Optional[List[int]] = None) -> torch.Tensor:
    dim = list(range(x.dim())) if dim is None else dim
    return roll(x, [(x.shape[d] + 1) // 2 for d in dim], dim)
