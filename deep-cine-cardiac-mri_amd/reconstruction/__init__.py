"""Drop-in ``reconstruction`` package backed by the MI355X HIP kernels.

Put ``deep-cine-cardiac-mri_amd/`` ahead of the reference checkout on
``sys.path`` and ``import reconstruction.models`` / ``reconstruction.utils``
resolve here (same class names, constructor arguments, forward signatures,
tensor layouts and state-dict keys as f78bono/deep-cine-cardiac-mri), while
the sub-packages this build does not replace (``pl_modules``, ``data.mri_data``
...) keep resolving to the reference when ``CINE_REFERENCE_ROOT`` points at it
(see INTEGRATION.md).
"""
import os as _os

_ref = _os.environ.get("CINE_REFERENCE_ROOT")
if _ref:
    _cand = _os.path.join(_ref, "reconstruction")
    if _os.path.isdir(_cand) and _cand not in __path__:
        __path__.append(_cand)
