"""Root-sum-of-squares coil combination (interface of the reference's utils/coil_combine.py).

On float32 GPU tensors: one HIP kernel (csrc/ew_kernels.hip: ``cine_rss``, squares summed over ``dim`` in index order, then the
root); host tensors take the reference's tensor expression.  The fused path computes its RSS inside ``cine_rss_normalise`` /
``cine_zero_filled_rss``.
"""
import torch

from cine_hip import ops
from cine_hip.autograd import needs_grad

from .math import complex_abs_sq


def rss(data: torch.Tensor, dim: int = 0) -> torch.Tensor:
    if data.is_cuda and data.dtype == torch.float32 and not needs_grad(data):       # (a tensor that requires grad: the differentiable expression)
        return ops.rss(data, dim, is_complex=False)
    return (data * data).sum(dim).sqrt()


def rss_complex(data: torch.Tensor, dim: int = 0) -> torch.Tensor:
    if data.is_cuda and data.dtype == torch.float32 and data.shape[-1] == 2 and dim % data.dim() != data.dim() - 1 and not needs_grad(data):
        return ops.rss(data, dim, is_complex=True)
    return complex_abs_sq(data).sum(dim).sqrt()
