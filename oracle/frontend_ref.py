"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the steps in FRONT of the hot path:
the reference's data front-end and an ESPIRiT calibration that stands in for BART's ``ecalib``.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Front-end (pinned: tests/golden/frontend.npz is made by running the reference's own lines, tests/golden/make_golden.py):
  reference data/mri_data.py:283-303   raw k-space -> centered IFFT2 -> crop + frame selection + Gaussian filter
                                       (data/transforms.py:186-220, scipy.ndimage.gaussian_filter) -> centered FFT2,
                                       coil-combined magnitude target, center crop (transforms.py:136-158)

ESPIRiT -- PARITY UNPINNED: the reference calls the BART toolbox (``bart ecalib -r N``, data/mri_data.py:296,
data/transforms.py:429; BART is a third-party C program, not vendored, no version pinned in requirements.txt) and the
image has no BART.  ``espirit_maps`` below is the published algorithm (Uecker et al., "ESPIRiT -- an eigenvalue approach
to autocalibrating parallel MRI", MRM 71:990-1001, 2014, sections "Calibration matrix", "Null-space / row-space" and
"Eigenvalue decomposition in image space") with ecalib's documented defaults (6 x 6 kernels, singular-value threshold
0.001, eigenvalue crop 0.8, first map, phase referenced to the first coil), written the direct way: explicit patch matrix,
full SVD, one dense Hermitian eigen-decomposition per pixel.  It is checked against the analytic coil maps of the
synthetic phantom (tests/test_oracle_golden.py); the HIP path (lag-kernel FFTs + power iteration) is checked against it.
"""
import numpy as np


# --------------------------------------------------------------------------------------------- front-end
def _gauss_weights(sigma: float, truncate: float = 4.0) -> np.ndarray:
    """scipy.ndimage._filters._gaussian_kernel1d (order 0): radius int(truncate * sigma + 0.5), normalised, float64."""
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1)
    w = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return w / w.sum()


def _reflect_index(i: np.ndarray, n: int) -> np.ndarray:
    """scipy 'reflect' boundary (d c b a | a b c d | d c b a): half-sample symmetric, any overhang."""
    p = 2 * n
    i = np.mod(i, p)
    return np.where(i >= n, p - 1 - i, i)


def gaussian_filter_axis(x: np.ndarray, sigma: float, axis: int) -> np.ndarray:
    """One axis pass of scipy.ndimage.gaussian_filter (mode 'reflect', truncate 4): float64 accumulation, float32 store."""
    if sigma <= 1e-15:
        return x
    w = _gauss_weights(sigma)
    r = (len(w) - 1) // 2
    n = x.shape[axis]
    acc = np.zeros(x.shape, np.float64)
    base = np.arange(n)
    for k in range(-r, r + 1):
        acc += w[k + r] * np.take(x, _reflect_index(base + k, n), axis=axis).astype(np.float64)
    return acc.astype(x.dtype)


def filtered_crop_center_and_slices(data: np.ndarray, shape, n_slices: int, filter_size):
    """reference data/transforms.py:186-220: data (t, c, x, y) complex -> (crop, Gaussian-filtered crop)."""
    y0, x0 = (data.shape[-2] - shape[0]) // 2, (data.shape[-1] - shape[1]) // 2
    crop = data[:n_slices, :, y0:y0 + shape[0], x0:x0 + shape[1]]
    re, im = np.ascontiguousarray(crop.real), np.ascontiguousarray(crop.imag)
    for ax, s in enumerate(filter_size):
        re, im = gaussian_filter_axis(re, s, ax), gaussian_filter_axis(im, s, ax)
    return crop, re + 1j * im


def _ifft2c_np(k):   # mri_data.py:289 (norm=None times sqrt(N) == ortho)
    return np.fft.fftshift(np.fft.ifftn(np.fft.ifftshift(k, axes=(-2, -1)), axes=(-2, -1), norm="ortho"), axes=(-2, -1))


def _fft2c_np(x):    # mri_data.py:292
    return np.fft.ifftshift(np.fft.fftn(np.fft.fftshift(x, axes=(-2, -1)), axes=(-2, -1), norm="ortho"), axes=(-2, -1))


def prepare_slice(kspace_txyc: np.ndarray, crop_shape=(200, 200), n_slices=15, filter_size=(0.7, 0.0, 0.3, 0.3), scaling=1e6):
    """reference data/mri_data.py:283-293: raw (t, x, y, c) complex64 -> (k-space (t, c, X, Y) complex64 of the filtered crop,
    filtered images (t, c, X, Y))."""
    k = np.asarray(kspace_txyc, np.complex64) * np.float32(scaling)
    k = k.transpose(0, 3, 1, 2)
    images = _ifft2c_np(k).astype(np.complex64)
    _, filt = filtered_crop_center_and_slices(images, crop_shape, n_slices, filter_size)
    return _fft2c_np(filt).astype(np.complex64), filt.astype(np.complex64)


def combine_target(images_filter: np.ndarray, sens: np.ndarray, crop_target=(180, 180)) -> np.ndarray:
    """reference data/mri_data.py:302-303: |sum_c img * conj(sens)| then center crop (transforms.py:136-158)."""
    t = np.abs(np.sum(images_filter * np.conjugate(sens[None]), axis=1)).astype(np.float32)
    h0, w0 = (t.shape[-2] - crop_target[0]) // 2, (t.shape[-1] - crop_target[1]) // 2
    return t[..., h0:h0 + crop_target[0], w0:w0 + crop_target[1]]


# --------------------------------------------------------------------------------------------- ESPIRiT
def calibration_matrix(acs: np.ndarray, k: int = 6) -> np.ndarray:
    """acs (c, r, r) complex -> A ((r-k+1)^2, k*k*c): one row per k x k k-space patch, columns ordered (py, px, coil)."""
    c, ry, rx = acs.shape
    rows = []
    for y in range(ry - k + 1):
        for x in range(rx - k + 1):
            rows.append(acs[:, y:y + k, x:x + k].transpose(1, 2, 0).reshape(-1))
    return np.asarray(rows)


def espirit_maps(kspace: np.ndarray, r: int = 24, k: int = 6, thresh: float = 1e-3, crop: float = 0.8, with_second: bool = False):
    """kspace (c, N, M) complex (centered, ortho) -> (maps (c, N, M) complex64, eigenvalue map (N, M) float32).

    1. A = patch matrix of the central r x r region; V_par = right singular vectors with sigma >= thresh * sigma_max.
    2. Every kept vector, reshaped to (k, k, c), is a set of k-space kernels; zero-padded to N x M and transformed to
       image space it gives G_j(r) in C^c.  The signal's sensitivities are eigenvectors with eigenvalue 1 of
       M(r) = (1 / k^2) sum_j conj(G_j(r)) G_j(r)^T  (a c x c Hermitian matrix per pixel).
    3. maps(r) = dominant eigenvector, unit norm, first coil real and non-negative; zero where lambda_max < crop.
    """
    c, ny, nx = kspace.shape
    r_y, r_x = min(r, ny), min(r, nx)
    y0, x0 = ny // 2 - r_y // 2, nx // 2 - r_x // 2
    acs = np.asarray(kspace[:, y0:y0 + r_y, x0:x0 + r_x], np.complex128)
    a = calibration_matrix(acs, k)
    _, s, vh = np.linalg.svd(a, full_matrices=False)
    vpar = vh[s >= thresh * s[0]].conj().T                       # (k*k*c, n)
    n = vpar.shape[1]
    kern = vpar.reshape(k, k, c, n)
    # image-space kernels: a k-space offset p (relative to the patch corner) is the modulation exp(-2 pi i p r / N) of the
    # centered image grid; evaluated directly (k*k terms per pixel)
    ry = (np.arange(ny) - ny // 2) / ny
    rx = (np.arange(nx) - nx // 2) / nx
    ey = np.exp(-2j * np.pi * np.outer(np.arange(k), ry))          # (k, ny)
    ex = np.exp(-2j * np.pi * np.outer(np.arange(k), rx))          # (k, nx)
    g = np.einsum("pqcn,py,qx->yxcn", kern, ey, ex)               # (ny, nx, c, n)
    m = np.einsum("yxcn,yxdn->yxcd", g.conj(), g) / (k * k)       # Hermitian (c x c) per pixel
    w, v = np.linalg.eigh(m)
    lam = w[..., -1]
    vec = v[..., -1]                                              # (ny, nx, c)
    ph = np.exp(-1j * np.angle(vec[..., :1]))
    vec = vec * ph
    vec = vec * (lam[..., None] >= crop)
    if with_second:                                               # second eigenvalue: how fast a power iteration converges
        return vec.transpose(2, 0, 1).astype(np.complex64), lam.astype(np.float32), w[..., -2].astype(np.float32)
    return vec.transpose(2, 0, 1).astype(np.complex64), lam.astype(np.float32)
