"""The steps in front of the hot path, on the device.

``prepare_slice``   reference data/mri_data.py:283-293, 302-303: raw k-space -> image space -> crop, frame selection,
                    Gaussian filter (data/transforms.py:186-220) -> k-space of the filtered crop, and the coil-combined
                    magnitude target.
``espirit_maps``    what the reference gets from the BART toolbox (``bart ecalib -r N``, mri_data.py:296,
                    transforms.py:429): ESPIRiT sensitivity maps (Uecker et al., MRM 71:990-1001, 2014) with ecalib's
                    defaults -- 6 x 6 kernels, singular-value threshold 0.001, eigenvalue crop 0.8, first map.
``ecalib``          the same behind ecalib's array convention ((1, x, y, coil) complex in, (x, y, coil) out), so the
                    reference's two call sites change by one line.

Arithmetic runs in libcine_hip.so (crop / filter / FFT / lag kernels / per-pixel eigen-iteration).  The two small dense
steps of the calibration -- the Gram matrix of the k-space patches and its Hermitian eigen-decomposition (k*k*coils square,
540 for 15 coils) -- are library calls on the same device (torch.matmul -> rocBLAS, torch.linalg.eigh -> hipSOLVER) in
complex128: the threshold keeps singular values down to 1e-3 of the largest, i.e. Gram eigenvalues down to 1e-6.
"""
from typing import Optional, Sequence, Tuple

import torch

from . import ops
from ._lib import CineHipError, check, lib


def _c2r(x: torch.Tensor) -> torch.Tensor:
    """complex64 tensor -> float32 (..., 2) pairs (a view)."""
    return torch.view_as_real(x) if x.is_complex() else x


def crop_select(x: torch.Tensor, n_slices: int, shape: Sequence[int]) -> torch.Tensor:
    """x (t, c, h, w, 2) -> (n_slices, c, shape[0], shape[1], 2): data[:n_slices, :, centered crop] (transforms.py:209-214)."""
    x = ops._dev(x, "crop_select input")
    t, c, h, w, _ = x.shape
    if not (0 < shape[0] <= h and 0 < shape[1] <= w):
        raise ValueError("Invalid shapes.")                                  # transforms.py:206-207
    n_slices = min(int(n_slices), t)
    out = torch.empty((n_slices, c, shape[0], shape[1], 2), device=x.device, dtype=x.dtype)
    check(lib().cine_crop_select(x.data_ptr(), out.data_ptr(), t, c, h, w, n_slices, shape[0], shape[1], ops._stream()), "cine_crop_select")
    return out


def gaussian_filter(x: torch.Tensor, sigma: Sequence[float]) -> torch.Tensor:
    """scipy.ndimage.gaussian_filter(x.real / x.imag, sigma) over the leading len(sigma) axes of x (..., 2), as
    transforms.py:216-218 applies it (one pass per axis with sigma > 0, in axis order)."""
    x = ops._dev(x, "gaussian_filter input")
    if len(sigma) != x.dim() - 1:
        raise ValueError("one sigma per axis")
    cur = x
    for ax, s in enumerate(sigma):
        if float(s) <= 1e-15:
            continue
        outer = 1
        for d in cur.shape[:ax]:
            outer *= d
        inner = 1
        for d in cur.shape[ax + 1:-1]:
            inner *= d
        nxt = torch.empty_like(cur)
        check(lib().cine_gauss_axis(cur.data_ptr(), nxt.data_ptr(), outer, cur.shape[ax], inner, float(s), ops._stream()), "cine_gauss_axis")
        cur = nxt
    return cur.clone() if cur is x else cur


def filtered_crop_center_and_slices(data: torch.Tensor, shape, n_slices: int, filter_size) -> Tuple[torch.Tensor, torch.Tensor]:
    """Device form of reference data/transforms.py:186-220 for (t, c, h, w, 2) float32 pairs."""
    crop = crop_select(data, n_slices, shape)
    return crop, gaussian_filter(crop, filter_size)


def combine_target(images: torch.Tensor, sens: torch.Tensor, crop_target: Sequence[int]) -> torch.Tensor:
    """center_crop(|sum_c images * conj(sens)|, crop_target) (mri_data.py:302-303): images (t, c, h, w, 2), sens (c, h, w, 2)."""
    images, sens = ops._dev(images, "images"), ops._dev(sens, "sens")
    t, c, h, w, _ = images.shape
    if tuple(sens.shape) != (c, h, w, 2):
        raise ValueError("sens must be (c, h, w, 2)")
    if not (0 < crop_target[0] <= h and 0 < crop_target[1] <= w):
        raise ValueError("Invalid shapes.")                                  # transforms.py:150-151
    out = torch.empty((t, crop_target[0], crop_target[1]), device=images.device, dtype=torch.float32)
    check(lib().cine_combine_target(images.data_ptr(), sens.data_ptr(), out.data_ptr(), t, c, h, w, crop_target[0], crop_target[1],
                                    ops._stream()), "cine_combine_target")
    return out


def prepare_slice(kspace_txyc: torch.Tensor, crop_shape=(200, 200), n_slices: int = 15,
                  filter_size=(0.7, 0.0, 0.3, 0.3), scaling: float = 1e6):
    """reference data/mri_data.py:283-293 on the device.  kspace_txyc: raw (t, x, y, coil) complex64 (the HDF5 ``y`` array)
    on the GPU.  Returns (kspace (t, coil, X, Y, 2) float32 of the filtered crop, filtered images (t, coil, X, Y, 2))."""
    if not kspace_txyc.is_cuda:
        raise CineHipError("prepare_slice: the HIP path needs a GPU tensor (no CPU fallback)")
    k = _c2r((kspace_txyc.to(torch.complex64) * scaling).permute(0, 3, 1, 2).contiguous())
    images = ops.fft2c(k, inverse=True)                                      # ifftn(norm=None) * sqrt(N) == ortho (:288-289)
    _, filt = filtered_crop_center_and_slices(images, crop_shape, n_slices, filter_size)
    # :291 transforms back with the shifts the other way round -- ifftshift(fftn(fftshift(x))) -- which differs from fft2c
    # (fftshift(fftn(ifftshift(x)))) along axes of ODD length by one sample on either side: fftshift(x) = roll(ifftshift(x), -1)
    # and ifftshift(y) = roll(fftshift(y), +1) there.  Even lengths (the reference's 200 x 200 crop): no difference.
    odd = [d for d, n in ((-3, filt.shape[-3]), (-2, filt.shape[-2])) if n % 2]
    x = torch.roll(filt, shifts=[-1] * len(odd), dims=odd).contiguous() if odd else filt
    kk = ops.fft2c(x)                                                        # fftn(norm=None) / sqrt(N) == ortho (:291-292)
    return (torch.roll(kk, shifts=[1] * len(odd), dims=odd).contiguous() if odd else kk), filt


def espirit_maps(kspace: torch.Tensor, r: int = 24, k: int = 6, thresh: float = 1e-3, crop: float = 0.8,
                 iters: int = 100) -> Tuple[torch.Tensor, torch.Tensor]:
    """kspace (coil, ny, nx, 2) float32 (centered, ortho; e.g. the time average of a cine slice) ->
    (maps (coil, ny, nx, 2), eigenvalue map (ny, nx)).  r: side of the central calibration region (ecalib -r)."""
    kspace = ops._dev(kspace, "espirit_maps kspace")
    c, ny, nx, _ = kspace.shape
    if c > 32:
        raise CineHipError("espirit_maps: at most 32 coils")
    ry, rx = min(int(r), ny), min(int(r), nx)
    if ry < k or rx < k or ny < 2 * k - 1 or nx < 2 * k - 1:
        raise ValueError("calibration region / image smaller than the kernel")
    y0, x0 = ny // 2 - ry // 2, nx // 2 - rx // 2
    acs = torch.view_as_complex(kspace)[:, y0:y0 + ry, x0:x0 + rx].to(torch.complex128)
    # rows = all k x k patches, columns ordered (py, px, coil)
    a = acs.unfold(1, k, 1).unfold(2, k, 1).permute(1, 2, 3, 4, 0).reshape((ry - k + 1) * (rx - k + 1), k * k * c)
    gram = a.conj().transpose(0, 1) @ a                                      # (k k c)^2, Hermitian
    try:
        ev, vec = torch.linalg.eigh(gram)
    except RuntimeError as e:                                                # no silent host fallback
        raise CineHipError(f"espirit_maps: torch.linalg.eigh failed on {gram.device}: {e}") from e
    keep = ev >= (thresh * thresh) * ev[-1]                                  # sigma >= thresh * sigma_max
    v = vec[:, keep]
    proj = torch.view_as_real((v @ v.conj().transpose(0, 1)).to(torch.complex64)).contiguous()
    kpad = torch.empty((c * c, ny, nx, 2), device=kspace.device, dtype=torch.float32)
    check(lib().cine_espirit_lag_kernels(proj.data_ptr(), kpad.data_ptr(), c, k, ny, nx, ops._stream()), "cine_espirit_lag_kernels")
    m = ops.fft2c(kpad, inverse=True)
    maps = torch.empty((c, ny, nx, 2), device=kspace.device, dtype=torch.float32)
    lam = torch.empty((ny, nx), device=kspace.device, dtype=torch.float32)
    check(lib().cine_espirit_eig(m.data_ptr(), maps.data_ptr(), lam.data_ptr(), c, ny * nx, int(iters), float(crop), ops._stream()),
          "cine_espirit_eig")
    return maps, lam


def ecalib(time_avg_kspace, *, r: int):
    """``r`` is required: the reference's two call sites differ (`-r 200`, mri_data.py:296; `-r 15`, transforms.py:429).
    Stand-in for ``bart.bart(2, 'ecalib -r N', time_avg_kspace)[0][..., 0]`` at the reference's call sites
    (mri_data.py:295-297, transforms.py:427-430): (1, x, y, coil) complex (numpy or tensor) -> (x, y, coil) complex of
    the same kind; the calibration itself runs on the GPU."""
    import numpy as np
    is_np = isinstance(time_avg_kspace, np.ndarray)
    t = torch.as_tensor(time_avg_kspace).to(torch.complex64)
    if t.dim() != 4 or t.shape[0] != 1:
        raise ValueError("ecalib expects (1, x, y, coil)")
    dev = t.device if t.is_cuda else torch.device("cuda")
    k = torch.view_as_real(t[0].permute(2, 0, 1).contiguous().to(dev)).contiguous()
    maps, _ = espirit_maps(k, r=r)
    out = torch.view_as_complex(maps).permute(1, 2, 0).contiguous()
    return out.cpu().numpy() if is_np else out.to(t.device)


def prepare_example(source, mask=None, sens=None, fname: str = "", crop_shape=(200, 200), crop_target=(180, 180), n_slices: int = 15,
                    filter_size=(0.7, 0.0, 0.3, 0.3), scaling: float = 1e6, ecalib_r: int = 200):
    """``SliceDataset.__getitem__`` of the reference (data/mri_data.py:267-311) in one piece, on the device: scale -> IFFT2 -> crop +
    frame selection + Gaussian filter -> FFT2 (k-space of the filtered crop) -> sensitivity maps from the time-averaged k-space
    (ESPIRiT, where the reference shells out to ``bart ecalib -r 200``; pass ``sens`` (coil, X, Y) complex to use given maps) ->
    coil-combined magnitude target -> center crop.  ``source``: the raw (t, x, y, coil) complex array / tensor, or an already-open
    h5py-like mapping holding it under ``"y"`` (and optionally ``"mask"``): the reader stays the caller's.
    Returns the reference's sample tuple (kspace (t, coil, X, Y) complex64, mask, target (t, cx, cy) float32, attrs, fname, dataslice)
    as numpy arrays, like the reference (the ``*DataTransform`` classes take it from there, data/transforms.py:300-352)."""
    import numpy as np
    if hasattr(source, "keys") and "y" in source:
        raw = np.asarray(source["y"])
        if mask is None and "mask" in source:
            mask = np.asarray(source["mask"])
    else:
        raw = source
    if not torch.cuda.is_available():
        raise CineHipError("prepare_example: the front-end kernels need a GPU (no CPU fallback)")
    y = torch.as_tensor(raw).to(torch.complex64).cuda()
    kspace, filt = prepare_slice(y, crop_shape, n_slices, filter_size, scaling)            # (t, c, X, Y, 2) each
    if sens is None:
        time_avg = torch.view_as_complex(kspace.mean(dim=0, keepdim=True).contiguous()).permute(0, 2, 3, 1)     # (1, X, Y, coil), :295
        smaps = ecalib(time_avg, r=ecalib_r).permute(2, 0, 1).contiguous()                  # (coil, X, Y), :297-298
    else:
        smaps = torch.as_tensor(sens).to(torch.complex64).cuda()
    target = combine_target(filt, torch.view_as_real(smaps).contiguous(), crop_target)      # :302-303
    k_np = torch.view_as_complex(kspace.contiguous()).cpu().numpy()
    return k_np, mask, target.cpu().numpy(), {}, fname, 0
