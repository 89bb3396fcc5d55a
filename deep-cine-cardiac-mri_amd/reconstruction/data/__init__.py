"""``reconstruction.data``: the hot path needs only ``transforms.mask_center`` / ``apply_mask`` (reference
models/varnet.py:9, data/transforms.py:66-108), which this build ships.  The dataset side (``mri_data``, ``subsample``,
``volume_sampler``: HDF5 / BART / samplers) stays the reference's: with ``CINE_REFERENCE_ROOT`` set those modules
resolve to the reference checkout, and the names its callers import from the package
(``pl_modules/data_module.py:15``: ``SliceDataset``, ``CombinedSliceDataset``, ``VolumeSampler``; reference
data/__init__.py:1-2) are forwarded lazily, so importing the models never needs h5py / bart.
"""
from .._fallthrough import extend_path as _extend_path

_extend_path(__path__, "data")

from . import transforms  # noqa: E402,F401

_LAZY = {"SliceDataset": "mri_data", "CombinedSliceDataset": "mri_data", "VolumeSampler": "volume_sampler"}


def __getattr__(name):
    mod = _LAZY.get(name)
    if mod is None:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
    import importlib
    try:
        return getattr(importlib.import_module(f"{__name__}.{mod}"), name)
    except ModuleNotFoundError as e:
        if e.name == f"{__name__}.{mod}":
            raise ImportError(f"reconstruction.data.{name} is the reference's dataset code; set CINE_REFERENCE_ROOT to the "
                              "f78bono/deep-cine-cardiac-mri checkout (INTEGRATION.md)") from e
        raise
