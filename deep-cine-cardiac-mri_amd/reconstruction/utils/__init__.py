"""HIP-backed counterparts of the reference's ``reconstruction.utils`` numeric helpers."""
from .._fallthrough import extend_path as _extend_path

_extend_path(__path__, "utils")      # modules this build does not ship resolve to the reference checkout

from .fftc import fft1c, ifft1c, fft2c, ifft2c, fftshift, ifftshift, roll
from .math import (complex_abs, complex_abs_sq, complex_conj, complex_mul,
                   complex_to_real_multi_ch, real_to_complex_multi_ch, tensor_to_complex_np)
from .coil_combine import rss, rss_complex
from .losses import SSIMLoss
from .padding import pad_for_mwcnn, unpad_from_mwcnn

__all__ = ["fft1c", "ifft1c", "fft2c", "ifft2c", "fftshift", "ifftshift", "roll",
           "complex_abs", "complex_abs_sq", "complex_conj", "complex_mul",
           "complex_to_real_multi_ch", "real_to_complex_multi_ch", "tensor_to_complex_np",
           "rss", "rss_complex", "SSIMLoss", "pad_for_mwcnn", "unpad_from_mwcnn"]
