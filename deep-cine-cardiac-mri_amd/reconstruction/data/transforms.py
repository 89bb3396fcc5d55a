"""``mask_center``, ``apply_mask``, the center crops (what ``training_step`` / ``validation_step`` call,
pl_modules/varnet_module.py:101) and ``filtered_crop_center_and_slices`` of the reference's data/transforms.py, plus the
on-device forms used by the bench (``cine_hip.ops.apply_mask`` / ``mask_center``).

Names this build does NOT implement (``to_tensor``, the ``*DataTransform`` classes: HDF5 / BART based dataset preparation) are
forwarded to the reference checkout (``CINE_REFERENCE_ROOT``) on first use, so ``pl_modules`` and ``mri_data`` keep working
against this module; nothing this build implements ever routes to the reference.
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


def center_crop(data: torch.Tensor, shape):
    """reference transforms.py:111-133: centered window of the last two dims (a view: autograd flows through it)."""
    if not (0 < shape[0] <= data.shape[-2] and 0 < shape[1] <= data.shape[-1]):
        raise ValueError("Invalid shapes.")
    w_from = (data.shape[-2] - shape[0]) // 2
    h_from = (data.shape[-1] - shape[1]) // 2
    return data[..., w_from:w_from + shape[0], h_from:h_from + shape[1]]


def complex_center_crop(data: torch.Tensor, shape):
    """reference transforms.py:136-158: the same window on dims -3 / -2 of a (..., h, w, 2) tensor."""
    if not (0 < shape[0] <= data.shape[-3] and 0 < shape[1] <= data.shape[-2]):
        raise ValueError("Invalid shapes.")
    w_from = (data.shape[-3] - shape[0]) // 2
    h_from = (data.shape[-2] - shape[1]) // 2
    return data[..., w_from:w_from + shape[0], h_from:h_from + shape[1], :]


def center_crop_to_smallest(x: torch.Tensor, y: torch.Tensor):
    """reference transforms.py:161-183: crop both images to the smaller height and the smaller width."""
    smallest_width = min(x.shape[-1], y.shape[-1])
    smallest_height = min(x.shape[-2], y.shape[-2])
    return center_crop(x, (smallest_height, smallest_width)), center_crop(y, (smallest_height, smallest_width))


def filtered_crop_center_and_slices(data, shape, n_slices, filter_size):
    """reference transforms.py:186-220 on the HIP kernels (cine_crop_select / cine_gauss_axis).  A GPU tensor of (t, c, h, w, 2)
    float32 pairs stays on the device; a complex numpy array (what mri_data.py:288 passes) is staged through the GPU and
    comes back as complex64 arrays.  Without a GPU this raises -- there is no CPU path."""
    from cine_hip import frontend
    if isinstance(data, torch.Tensor) and data.is_cuda:
        return frontend.filtered_crop_center_and_slices(data, shape, n_slices, filter_size)
    import numpy as np
    from cine_hip._lib import CineHipError
    if not torch.cuda.is_available():
        raise CineHipError("filtered_crop_center_and_slices: the crop / Gaussian filter run in the HIP kernels and no GPU is "
                           "visible (there is no CPU fallback)")
    arr = data.detach().cpu().numpy() if isinstance(data, torch.Tensor) else np.asarray(data)
    if not np.iscomplexobj(arr):
        raise ValueError("filtered_crop_center_and_slices expects complex (t, c, h, w) data")
    pairs = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(arr.astype(np.complex64)))).cuda()
    crop, filt = frontend.filtered_crop_center_and_slices(pairs, shape, n_slices, filter_size)
    back = lambda v: torch.view_as_complex(v.contiguous()).cpu().numpy()
    return back(crop), back(filt)


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    try:
        return getattr(_load_shadowed("reconstruction.data", "transforms"), name)
    except AttributeError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None
