"""Root-sum-of-squares coil combination (interface of the reference's utils/coil_combine.py)."""
import torch

from .math import complex_abs_sq


def rss(data: torch.Tensor, dim: int = 0) -> torch.Tensor:
    return (data * data).sum(dim).sqrt()


def rss_complex(data: torch.Tensor, dim: int = 0) -> torch.Tensor:
    return complex_abs_sq(data).sum(dim).sqrt()
