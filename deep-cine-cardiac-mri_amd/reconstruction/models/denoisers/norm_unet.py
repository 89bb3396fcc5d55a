"""NormUnet: group-normalise, pad to x16, U-Net, unpad, un-normalise -- on the HIP kernels.

Mirrors the reference's denoisers/norm_unet.py (NormUnet :12-114, NormUnet3D
:117-219): same constructor and ``unet.*`` state-dict keys.
"""
import torch
from torch import nn

from cine_hip import autograd as ag
from cine_hip import ops
from .unet import Unet


class NormUnet(nn.Module):
    def __init__(self, chans: int, num_pools: int, in_chans: int = 2, out_chans: int = 2, drop_prob: float = 0.0):
        super().__init__()
        self.unet = Unet(in_chans=in_chans, out_chans=out_chans, chans=chans, num_pool_layers=num_pools,
                         drop_prob=drop_prob, dims=2)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.shape[-1] == 2:
            raise ValueError("Last dimension must be 2 for complex.")
        b, c, h, w, _ = x.shape
        if c != 1:
            raise NotImplementedError("HIP NormUnet handles one complex channel per sample (all reference call sites)")
        if ag.grad_mode(self):       # training: pack -> U-Net -> unpack as one autograd node with a HIP backward
            return ag.norm_unet(x.reshape(b, h, w, 2), self.unet.hip_weights()).view(b, 1, h, w, 2)
        planes, stats = ops.normunet_pack(x.reshape(b, h, w, 2))
        planes = self.unet(planes)
        return ops.normunet_unpack(planes, stats, h, w).view(b, 1, h, w, 2)


class NormUnet3D(nn.Module):
    """NormUnet over (t, h, w) volumes: group norm, pad all three dims to x16, Conv3d U-Net, unpad, un-normalise."""

    def __init__(self, chans: int, num_pools: int, in_chans: int = 2, out_chans: int = 2, drop_prob: float = 0.0):
        super().__init__()
        self.unet = Unet(in_chans=in_chans, out_chans=out_chans, chans=chans, num_pool_layers=num_pools,
                         drop_prob=drop_prob, dims=3)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.shape[-1] == 2:
            raise ValueError("Last dimension must be 2 for complex.")
        b, c, t, h, w, _ = x.shape
        if c != 1:
            raise NotImplementedError("HIP NormUnet3D handles one complex channel per sample (all reference call sites)")
        if ag.grad_mode(self):
            return self._forward_train(x.reshape(b, t, h, w, 2)).view(b, 1, t, h, w, 2)
        planes, stats = ops.normunet3d_pack(x.reshape(b, t, h, w, 2))
        planes = self.unet(planes)
        return ops.normunet3d_unpack(planes, stats, t, h, w).view(b, 1, t, h, w, 2)

    def _forward_train(self, x: torch.Tensor) -> torch.Tensor:
        """The two halves around the 3-D U-Net (reference norm_unet.py:149-219) in differentiable torch element-wise ops -- group mean and
        unbiased std per (sample, re / im), centred zero pad of (t, h, w) to multiples of 16, crop, un-normalise -- around ag.unet3d, whose
        backward pass runs on the HIP kernels.  x (b, t, h, w, 2) -> (b, t, h, w, 2)."""
        b, t, h, w, _ = x.shape
        vol = x.permute(0, 4, 1, 2, 3)                                   # (b, 2, t, h, w)
        flat = vol.reshape(b, 2, -1)
        mean = flat.mean(dim=2).view(b, 2, 1, 1, 1)
        std = flat.std(dim=2).view(b, 2, 1, 1, 1)
        vol = (vol - mean) / std
        pads = []
        for size in (w, h, t):                                           # F.pad order: last dimension first
            extra = ((size - 1) | 15) + 1 - size
            pads += [extra // 2, extra - extra // 2]
        out = ag.unet3d(nn.functional.pad(vol, pads).contiguous(), self.unet.hip_weights())
        out = out[:, :, pads[4]:pads[4] + t, pads[2]:pads[2] + h, pads[0]:pads[0] + w]
        return (out * std + mean).permute(0, 2, 3, 4, 1).contiguous()
