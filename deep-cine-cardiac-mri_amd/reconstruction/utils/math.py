"""Complex helpers on trailing-pair tensors (interface of the reference's utils/math.py).

The fused kernels never call these -- sens-multiply, conjugate and magnitude live
inside the FFT passes -- they exist so user code written against the reference
keeps working.  On float32 GPU tensors every one of them is a HIP kernel (csrc/ew_kernels.hip:
``cine_complex_mul`` with broadcasting, ``cine_complex_conj``, ``cine_complex_abs[_sq]``); host tensors -- the
reference's numpy-side dataset code -- take the tensor expressions of the reference.  A tensor that REQUIRES GRAD (a loss written with
these helpers) never goes to a raw kernel: ``complex_abs`` / ``complex_conj`` run as autograd Functions on the same kernels, the others take
the reference's differentiable tensor expression (the raw kernels return tensors without a ``grad_fn``).
"""
import numpy as np
import torch

from cine_hip import ops
from cine_hip import autograd as ag


def _check(*xs):
    for x in xs:
        if not x.shape[-1] == 2:
            raise ValueError("Tensor does not have separate complex dim.")


def complex_mul(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    if not x.shape[-1] == y.shape[-1] == 2:
        raise ValueError("Tensors do not have separate complex dim.")
    if x.is_cuda and y.is_cuda and x.dtype == y.dtype == torch.float32 and not ag.needs_grad(x, y):
        return ops.complex_mul(x, y)
    z = torch.view_as_complex(x.contiguous()) * torch.view_as_complex(y.contiguous())
    return torch.view_as_real(z)


def complex_conj(x: torch.Tensor) -> torch.Tensor:
    _check(x)
    if x.is_cuda and x.dtype == torch.float32:
        return ag.ConjFn.apply(x) if ag.needs_grad(x) else ops.complex_conj(x)
    return x * x.new_tensor([1.0, -1.0])


def complex_abs(data: torch.Tensor) -> torch.Tensor:
    _check(data)
    if data.is_cuda and data.dtype == torch.float32:
        return ag.AbsFn.apply(data) if ag.needs_grad(data) else ops.complex_abs(data)
    return complex_abs_sq(data).sqrt()


def complex_abs_sq(data: torch.Tensor) -> torch.Tensor:
    _check(data)
    if data.is_cuda and data.dtype == torch.float32 and not ag.needs_grad(data):
        return ops.complex_abs_sq(data)
    return (data * data).sum(dim=-1)


def tensor_to_complex_np(data: torch.Tensor) -> np.ndarray:
    data = data.numpy()
    return data[..., 0] + 1j * data[..., 1]


def real_to_complex_multi_ch(x: torch.Tensor, n: int) -> torch.Tensor:
    if not x.shape[-1] == 2 * n:
        raise ValueError("Real and imaginary parts do not have the same size")
    return torch.complex(x[..., :n], x[..., n:])


def complex_to_real_multi_ch(x: torch.Tensor) -> torch.Tensor:
    return torch.cat([x.real, x.imag], dim=-1)
