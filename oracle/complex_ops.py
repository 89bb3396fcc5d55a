"""Oracle: complex arithmetic on trailing-pair tensors and coil combination.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference ``reconstruction/utils/math.py`` and
``reconstruction/utils/coil_combine.py``.
"""
import torch


def _need_pair(*xs: torch.Tensor) -> None:
    for x in xs:
        if x.shape[-1] != 2:
            raise ValueError("Tensor does not have separate complex dim.")


def complex_mul(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """math.py:5-25: (a+ib)(c+id), broadcasting, result stacked on a new last dim."""
    if not (x.shape[-1] == y.shape[-1] == 2):
        raise ValueError("Tensors do not have separate complex dim.")
    a, b = x[..., 0], x[..., 1]
    c, d = y[..., 0], y[..., 1]
    return torch.stack((a * c - b * d, a * d + b * c), dim=-1)


def complex_conj(x: torch.Tensor) -> torch.Tensor:
    """math.py:28-45."""
    _need_pair(x)
    return torch.stack((x[..., 0], -x[..., 1]), dim=-1)


def complex_abs_sq(x: torch.Tensor) -> torch.Tensor:
    """math.py:65-79."""
    _need_pair(x)
    return (x * x).sum(dim=-1)


def complex_abs(x: torch.Tensor) -> torch.Tensor:
    """math.py:48-62."""
    return complex_abs_sq(x).sqrt()


def rss(x: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """coil_combine.py:5-18."""
    return torch.sqrt((x * x).sum(dim))


def rss_complex(x: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """coil_combine.py:21-34."""
    return torch.sqrt(complex_abs_sq(x).sum(dim))


def real_to_complex_multi_ch(x: torch.Tensor, n: int) -> torch.Tensor:
    """math.py:97-118: [re_0..re_{n-1}, im_0..im_{n-1}] -> complex with n channels."""
    if x.shape[-1] != 2 * n:
        raise ValueError("Real and imaginary parts do not have the same size")
    return torch.complex(x[..., :n], x[..., n:])


def complex_to_real_multi_ch(z: torch.Tensor) -> torch.Tensor:
    """math.py:121-135."""
    return torch.cat((z.real, z.imag), dim=-1)


def mask_center(x: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
    """data/transforms.py:95-108: keep rows [lo, hi) of dim 2, zero the rest."""
    out = torch.zeros_like(x)
    out[:, :, lo:hi] = x[:, :, lo:hi]
    return out
