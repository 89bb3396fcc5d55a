"""Oracle: centered, ortho-normalised FFTs on (..., 2) real-pair tensors.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference ``reconstruction/utils/fftc.py``:
  * ``ifftshift`` rolls by (n + 1) // 2        (fftc.py:191-213)
  * ``fftshift``  rolls by  n // 2             (fftc.py:166-188)
  * ``fft1c`` / ``ifft1c`` act on dim -2       (fftc.py:5-56)
  * ``fft2c`` / ``ifft2c`` act on dims -3, -2  (fftc.py:59-110)
all with norm="ortho".  The reference builds its rolls out of narrow+cat
(fftc.py:119-163); ``torch.roll`` is the same permutation.
"""
from typing import Sequence

import torch


def _need_pair(x: torch.Tensor) -> None:
    # fftc.py:18-19 / 45-46 / 72-73 / 99-100
    if x.shape[-1] != 2:
        raise ValueError("Tensor does not have separate complex dim.")


def roll(x: torch.Tensor, shift: Sequence[int], dim: Sequence[int]) -> torch.Tensor:
    """fftc.py:141-163."""
    if len(shift) != len(dim):
        raise ValueError("len(shift) must match len(dim)")
    return torch.roll(x, shifts=tuple(int(s) for s in shift), dims=tuple(dim))


def fftshift(x: torch.Tensor, dim: Sequence[int] = None) -> torch.Tensor:
    """fftc.py:166-188: shift = n // 2."""
    if dim is None:
        dim = list(range(x.dim()))
    return roll(x, [x.shape[d] // 2 for d in dim], dim)


def ifftshift(x: torch.Tensor, dim: Sequence[int] = None) -> torch.Tensor:
    """fftc.py:191-213: shift = (n + 1) // 2."""
    if dim is None:
        dim = list(range(x.dim()))
    return roll(x, [(x.shape[d] + 1) // 2 for d in dim], dim)


def _centered(x: torch.Tensor, real_dims, inverse: bool) -> torch.Tensor:
    _need_pair(x)
    x = ifftshift(x, dim=real_dims)
    z = torch.view_as_complex(x.contiguous())
    # complex view drops the trailing pair dim: real dim d (<0) -> d + 1
    cdims = tuple(d + 1 for d in real_dims)
    z = (torch.fft.ifftn if inverse else torch.fft.fftn)(z, dim=cdims, norm="ortho")
    return fftshift(torch.view_as_real(z), dim=real_dims)


def fft1c(x: torch.Tensor) -> torch.Tensor:
    """fftc.py:5-29."""
    return _centered(x, [-2], inverse=False)


def ifft1c(x: torch.Tensor) -> torch.Tensor:
    """fftc.py:32-56."""
    return _centered(x, [-2], inverse=True)


def fft2c(x: torch.Tensor) -> torch.Tensor:
    """fftc.py:59-83."""
    return _centered(x, [-3, -2], inverse=False)


def ifft2c(x: torch.Tensor) -> torch.Tensor:
    """fftc.py:86-110."""
    return _centered(x, [-3, -2], inverse=True)


def xpd_temporal_fft(z: torch.Tensor, dim: int = 1) -> torch.Tensor:
    """XPDNet's XF transform on complex tensors, xpdnet.py:466:
    ifftshift(fft(fftshift(z))) -- the opposite shift order from fft1c, which
    differs for odd lengths."""
    n = z.shape[dim]
    return torch.fft.ifftshift(
        torch.fft.fft(torch.fft.fftshift(z, dim=dim), n, dim, "ortho"), dim=dim)


def xpd_temporal_ifft(z: torch.Tensor, dim: int = 1) -> torch.Tensor:
    """xpdnet.py:500: fftshift(ifft(ifftshift(z)))."""
    n = z.shape[dim]
    return torch.fft.fftshift(
        torch.fft.ifft(torch.fft.ifftshift(z, dim=dim), n, dim, "ortho"), dim=dim)
