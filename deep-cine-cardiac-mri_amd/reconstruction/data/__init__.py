"""Only what the hot path imports from ``reconstruction.data`` (reference varnet.py:9)."""
from . import transforms  # noqa: F401
