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
        if ag.grad_mode(self):
            raise NotImplementedError("training through the 3-D U-Net is not on the HIP path yet")
        if c != 1:
            raise NotImplementedError("HIP NormUnet3D handles one complex channel per sample (all reference call sites)")
        planes, stats = ops.normunet3d_pack(x.reshape(b, t, h, w, 2))
        planes = self.unet(planes)
        return ops.normunet3d_unpack(planes, stats, t, h, w).view(b, 1, t, h, w, 2)
