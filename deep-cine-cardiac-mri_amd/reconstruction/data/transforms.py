"""``mask_center`` and ``apply_mask`` of the reference's data/transforms.py (:66-108), device-agnostic, plus the
on-device forms used by the bench (``cine_hip.ops.apply_mask`` / ``mask_center``).

Everything else in the reference's transforms.py (``to_tensor``, ``center_crop*``, the ``*DataTransform`` classes: HDF5 /
BART based dataset preparation) is outside the accelerated path; those names are forwarded to the reference checkout
(``CINE_REFERENCE_ROOT``) on first use, so ``pl_modules`` and ``mri_data`` keep working against this module.
"""
import torch

from cine_hip import synth as _synth
from .._fallthrough import load_shadowed as _load_shadowed


def apply_mask(data: torch.Tensor, mask_func, seed=None):
    """reference transforms.py:66-92: one row mask per frame from ``mask_func`` (host RNG, as in the reference), applied
    on the device (cine_apply_mask) when ``data`` lives there."""
    if not data.is_cuda:
        return _synth.apply_mask(data, mask_func, seed)
    import numpy as np
    from cine_hip import ops
    shape = np.array(data.shape)
    shape[1] = 1
    mask = mask_func(shape, seed)                                    # (t, 1, h, 1, 1) float, values 0 / 1
    return ops.apply_mask(data, mask.to(torch.uint8).to(data.device)), mask.to(data.device)


def mask_center(x: torch.Tensor, mask_from: int, mask_to: int) -> torch.Tensor:
    """Keep rows [mask_from, mask_to) of dim 2, zero the rest (reference transforms.py:95-108)."""
    out = torch.zeros_like(x)
    out[:, :, mask_from:mask_to] = x[:, :, mask_from:mask_to]
    return out


def filtered_crop_center_and_slices(data, shape, n_slices, filter_size):
    """reference transforms.py:186-220.  A GPU tensor of (t, c, h, w, 2) float32 pairs is cropped and Gaussian-filtered by the
    HIP kernels (cine_crop_select / cine_gauss_axis); numpy input goes to the reference's scipy implementation."""
    if isinstance(data, torch.Tensor) and data.is_cuda:
        from cine_hip import frontend
        return frontend.filtered_crop_center_and_slices(data, shape, n_slices, filter_size)
    return _load_shadowed("reconstruction.data", "transforms").filtered_crop_center_and_slices(data, shape, n_slices, filter_size)


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    try:
        return getattr(_load_shadowed("reconstruction.data", "transforms"), name)
    except AttributeError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None
