"""Synthetic cine k-space, sampling masks and seeded weights (host side, numpy).

There is no dataset or checkpoint on the build/bench machines, so benchmarks
and parity tests run on a deterministic synthetic cine volume:

  moving-disc complex phantom (t, h, w)  x  analytic coil maps (c, h, w)
    -> centered ortho FFT2 -> per-frame Cartesian row mask -> masked k-space

The mask generators restate the sampling semantics of the reference
``reconstruction/data/subsample.py`` (RandomMaskFunc :75-151,
EquispacedMaskFunc :154-215) and ``transforms.apply_mask`` (:66-92); they are
checked bit-for-bit against the reference in tests/test_synth_golden.py.
This is host-side input preparation, exactly as in the reference (numpy in a
DataLoader worker); it is not part of the device hot path.
"""
import zlib
from typing import Optional, Sequence, Tuple

import numpy as np
import torch


# --------------------------------------------------------------------------- masks
class RandomMaskFunc:
    """Gaussian-density Cartesian row mask, one draw per frame.

    subsample.py:75-151.  ``center_fractions`` holds the NUMBER of always-sampled
    central lines (script default [10]); ``int(Nx / acc)`` lines per frame in
    total.  Rows are drawn with ``np.random.choice`` from the GLOBAL numpy RNG
    (subsample.py:139), so ``np.random.seed(s)`` before the call reproduces the
    reference mask.
    """

    def __init__(self, center_fractions: Sequence[float], accelerations: Sequence[int]):
        if len(center_fractions) != len(accelerations):
            raise ValueError("Number of center fractions should match number of accelerations")
        self.center_fractions = center_fractions
        self.accelerations = accelerations
        self.rng = np.random.RandomState()

    def choose_acceleration(self):
        choice = self.rng.randint(0, len(self.accelerations))     # subsample.py:66-71
        return self.center_fractions[choice], self.accelerations[choice]

    def __call__(self, shape: Sequence[int], seed=None) -> torch.Tensor:
        if len(shape) < 3:
            raise ValueError("Shape should have 3 or more dimensions")
        if seed is None:
            sample_n, acc = self.choose_acceleration()
        else:                                                     # temp_seed, :15-28
            state = self.rng.get_state()
            self.rng.seed(seed)
            try:
                sample_n, acc = self.choose_acceleration()
            finally:
                self.rng.set_state(state)
        n_frames, _, nx, _, _ = shape
        # tail-adjusted Gaussian density over rows (:118-127)
        pdf = np.exp(-(0.5 / (nx / 10.0) ** 2) * (np.arange(nx) - nx / 2) ** 2)
        pdf += (nx / (2.0 * acc)) * 1.0 / nx
        n_lines = int(nx / acc)
        lo, hi = nx // 2 - sample_n // 2, nx // 2 + sample_n // 2
        if sample_n:                                              # :129-134
            pdf[lo:hi] = 0
            pdf /= np.sum(pdf)
            n_lines -= sample_n
        mask = np.zeros((n_frames, nx))
        for i in range(n_frames):                                 # :136-140
            mask[i, np.random.choice(nx, n_lines, False, pdf)] = 1
        if sample_n:
            mask[:, lo:hi] = 1                                    # :142-144
        out_shape = [1] * len(shape)
        out_shape[-3] = nx
        out_shape[0] = n_frames
        return torch.from_numpy(mask.reshape(*out_shape).astype(np.float32))


class EquispacedMaskFunc(RandomMaskFunc):
    """subsample.py:154-215 (same mask for every frame; fraction-valued centre)."""

    def __call__(self, shape: Sequence[int], seed=None) -> torch.Tensor:
        if len(shape) < 3:
            raise ValueError("Shape should have 3 or more dimensions")
        state = self.rng.get_state() if seed is not None else None
        if seed is not None:
            self.rng.seed(seed)
        try:
            frac, acc = self.choose_acceleration()
            rows = shape[-3]
            n_low = int(round(rows * frac))
            mask = np.zeros(rows, dtype=np.float32)
            pad = (rows - n_low + 1) // 2
            mask[pad:pad + n_low] = True
            adj = (acc * (n_low - rows)) / (n_low * acc - rows)
            offset = self.rng.randint(0, round(adj))
            picks = np.around(np.arange(offset, rows - 1, adj)).astype(np.uint)
            mask[picks] = True
        finally:
            if state is not None:
                self.rng.set_state(state)
        out_shape = [1] * len(shape)
        out_shape[-3] = rows
        return torch.from_numpy(mask.reshape(*out_shape).astype(np.float32))


def create_mask_for_mask_type(kind: str, center_fractions, accelerations):
    """subsample.py:218-235."""
    if kind == "random":
        return RandomMaskFunc(center_fractions, accelerations)
    if kind == "equispaced":
        return EquispacedMaskFunc(center_fractions, accelerations)
    raise Exception(f"{kind} not supported")


def apply_mask(data: torch.Tensor, mask_func, seed=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """transforms.py:66-92.  data (t, c, h, w, 2); one mask per frame, shared by coils."""
    shape = np.array(data.shape)
    shape[1] = 1
    mask = mask_func(shape, seed)
    return data * mask + 0.0, mask


# --------------------------------------------------------------------------- phantom
def _fft2c_np(x: np.ndarray) -> np.ndarray:
    ax = (-2, -1)
    return np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(x, axes=ax), norm="ortho"), axes=ax)


def cine_phantom(t: int, h: int, w: int, seed: int = 0) -> np.ndarray:
    """Complex (t, h, w) phantom: a static body ellipse, a pulsating 'ventricle'
    ring and a few drifting discs; imag = 0.2 * real plus a smooth phase roll."""
    rs = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    frames = []
    discs = [(rs.uniform(-.5, .5), rs.uniform(-.5, .5), rs.uniform(.05, .15),
              rs.uniform(.3, 1.0), rs.uniform(0, 2 * np.pi)) for _ in range(6)]
    for k in range(t):
        ph = 2 * np.pi * k / max(t, 1)
        img = 0.35 * ((xx / .85) ** 2 + (yy / .7) ** 2 < 1)
        r_out = .32 + .05 * np.sin(ph)
        r_in = .18 + .07 * np.sin(ph)
        rr = np.sqrt((xx + .1) ** 2 + (yy - .05) ** 2)
        img = img + .55 * ((rr < r_out) & (rr > r_in)) + .25 * (rr <= r_in)
        for (cx, cy, rad, amp, p0) in discs:
            dx, dy = .06 * np.cos(ph + p0), .06 * np.sin(ph + p0)
            img = img + amp * .3 * (((xx - cx - dx) ** 2 + (yy - cy - dy) ** 2) < rad ** 2)
        frames.append(img)
    real = np.stack(frames).astype(np.float64)
    phase = np.exp(1j * 0.3 * (xx + 0.5 * yy))[None]
    return ((real + 0.2j * real) * phase).astype(np.complex64)


def coil_maps(c: int, h: int, w: int) -> np.ndarray:
    """Analytic smooth sensitivities: Gaussian magnitude x linear phase on a ring
    of coil centres, RSS-normalised.  (c, h, w) complex64."""
    yy, xx = np.meshgrid(np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    maps = []
    for k in range(c):
        a = 2 * np.pi * k / c
        cx, cy = 1.1 * np.cos(a), 1.1 * np.sin(a)
        mag = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / 1.6)
        ph = np.exp(1j * (0.8 * np.cos(a) * xx + 0.8 * np.sin(a) * yy + a))
        maps.append(mag * ph)
    s = np.stack(maps)
    s = s / np.sqrt((np.abs(s) ** 2).sum(0, keepdims=True))
    return s.astype(np.complex64)


def _pairs(z: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.stack((z.real, z.imag), axis=-1).astype(np.float32))


def make_cine_slice(t: int = 15, c: int = 15, h: int = 200, w: int = 200, accel: int = 4,
                    center_lines: int = 10, seed: int = 0, mask_type: str = "random", noise_std: float = 0.0):
    """One synthetic cine example in the layout the models consume.

    Returns dict with
      masked_kspace (1, t, c, h, w, 2) f32, mask (1, t, 1, h, 1, 1) uint8,
      sens_maps (1, 1, c, h, w, 2) f32, target (1, t, h, w) f32 (|phantom|),
      kspace (1, t, c, h, w, 2) f32 fully sampled.
    Masks come from ``np.random.seed(seed)`` + RandomMaskFunc, as a reference
    run with the same seed would draw them (mask shape / dtype per
    transforms.py:341-343).  ``noise_std`` > 0 adds complex white noise of that
    standard deviation per k-space sample (RandomState(seed + 7919)), as a
    measurement has: the noise-free phantom has image columns that are exactly
    empty, and networks without an input normalisation (XPDNet's MWCNN) turn
    the rounding noise of such planes into O(1) output.
    """
    img = cine_phantom(t, h, w, seed)
    sens = coil_maps(c, h, w)
    k = _fft2c_np(img[:, None] * sens[None])                      # (t, c, h, w)
    kspace = _pairs(k)
    if noise_std > 0:
        rs = np.random.RandomState(seed + 7919)
        kspace = kspace + torch.from_numpy((noise_std * rs.standard_normal(tuple(kspace.shape))).astype(np.float32))
    np.random.seed(seed)
    mf = create_mask_for_mask_type(mask_type, [center_lines], [accel])
    masked, mask = apply_mask(kspace, mf, None)
    return {
        "masked_kspace": masked.unsqueeze(0).contiguous(),
        "mask": mask.unsqueeze(0).byte().contiguous(),
        "sens_maps": _pairs(sens).unsqueeze(0).unsqueeze(0).contiguous(),
        "target": torch.from_numpy(np.abs(img)).float().unsqueeze(0),
        "kspace": kspace.unsqueeze(0).contiguous(),
    }


# --------------------------------------------------------------------------- weights
def fill_parameters_(module: torch.nn.Module, seed: int = 1, keep: Sequence[str] = ("lambda",)):
    """Deterministic, torch-RNG-independent weight fill keyed by parameter NAME.

    Each unique parameter (first name under which ``named_parameters`` yields
    it) is drawn uniform(-b, b), b = 1/sqrt(fan_in), from
    RandomState(crc32(name) ^ seed).  Parameters whose name contains an entry
    of ``keep`` (the DC lambdas) keep their initial value.  The reference model
    and the HIP model share parameter names, so the same call gives both the
    same weights on any machine.
    """
    with torch.no_grad():
        for name, p in module.named_parameters():
            if any(k in name for k in keep):
                continue
            rs = np.random.RandomState((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
            fan_in = int(np.prod(p.shape[1:])) if p.dim() > 1 else int(p.shape[0])
            b = 1.0 / np.sqrt(max(fan_in, 1))
            vals = rs.uniform(-b, b, size=tuple(p.shape)).astype(np.float32)
            p.copy_(torch.from_numpy(vals).to(p.device))
    return module
