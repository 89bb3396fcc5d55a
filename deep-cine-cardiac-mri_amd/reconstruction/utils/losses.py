"""``SSIMLoss`` with the reference's interface and numerics (utils/losses.py:6-58), device-agnostic.

Time-averaged ``1 - SSIM`` over dim 2 of (b, 1, t, h, w) tensors: 7x7 uniform window (valid region only), sample
covariance (N / (N-1)), K1 = 0.01, K2 = 0.03, and -- as the reference does (:34) -- the data range of every frame is
the maximum of that TARGET frame, whatever ``data_range`` was passed.  GPU tensors of the shape ``training_step`` passes
((1, 1, t, h, w), pl_modules/varnet_module.py:110-112) go through the HIP kernels (cine_ssim_loss / cine_ssim_loss_bwd: window
sums in float64, hand-written backward); anything else -- CPU tensors in the host-side tests -- through the same formula in torch ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class SSIMLoss(nn.Module):
    def __init__(self, win_size: int = 7, k1: float = 0.01, k2: float = 0.03):
        super().__init__()
        self.win_size = win_size
        self.k1, self.k2 = k1, k2
        self.register_buffer("w", torch.ones(1, 1, win_size, win_size) / win_size ** 2)
        npix = win_size ** 2
        self.cov_norm = npix / (npix - 1)

    def forward(self, Xt: torch.Tensor, Yt: torch.Tensor, data_range: torch.Tensor = None) -> torch.Tensor:
        if Xt.is_cuda and Xt.dim() == 5 and Xt.shape[0] == 1 and Xt.shape[1] == 1 and Xt.dtype == torch.float32 and Yt.shape == Xt.shape:
            from cine_hip.autograd import SsimLossFn
            return SsimLossFn.apply(Xt[0, 0], Yt[0, 0], self.win_size, self.k1, self.k2)
        w = self.w.to(device=Xt.device, dtype=Xt.dtype)
        total = 0.0
        frames = Xt.shape[2]
        for t in range(frames):
            X, Y = Xt[:, :, t], Yt[:, :, t]
            rng = Y.max().reshape(1, 1, 1, 1)
            c1, c2 = (self.k1 * rng) ** 2, (self.k2 * rng) ** 2
            ux, uy = F.conv2d(X, w), F.conv2d(Y, w)
            uxx, uyy, uxy = F.conv2d(X * X, w), F.conv2d(Y * Y, w), F.conv2d(X * Y, w)
            vx = self.cov_norm * (uxx - ux * ux)
            vy = self.cov_norm * (uyy - uy * uy)
            vxy = self.cov_norm * (uxy - ux * uy)
            s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
            total = total + (1 - s.mean())
        return total / frames
