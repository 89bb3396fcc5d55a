"""``mask_center`` and ``apply_mask`` of the reference's data/transforms.py, device-agnostic.

Everything else in the reference's transforms.py (HDF5 / BART based dataset
transforms) is outside the accelerated path and is not restated here.
"""
import torch

from cine_hip.synth import apply_mask  # noqa: F401  (reference transforms.py:66-92)


def mask_center(x: torch.Tensor, mask_from: int, mask_to: int) -> torch.Tensor:
    """Keep rows [mask_from, mask_to) of dim 2, zero the rest (reference transforms.py:95-108)."""
    out = torch.zeros_like(x)
    out[:, :, mask_from:mask_to] = x[:, :, mask_from:mask_to]
    return out
