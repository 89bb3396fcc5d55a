"""Oracle: U-Net / NormUnet image regularisers (CPU, torch.nn).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference ``reconstruction/models/denoisers/unet.py`` and
``norm_unet.py``.  Attribute names are kept identical to the reference so a
reference ``state_dict`` loads with ``strict=True``:
  Unet.down_sample_layers[i].layers.{0,4}, Unet.conv.layers.{0,4},
  Unet.up_transpose_conv[i].layers.0, Unet.up_conv[i](.0).layers.{0,4},
  Unet.up_conv[-1].1 (1x1 conv with bias).
"""
import math
from typing import List, Tuple

import torch
from torch import nn
from torch.nn import functional as F


def _ops(dims: int):
    assert dims in (2, 3), "Dimensions must be either 2 or 3"   # unet.py:43-44
    if dims == 2:
        return nn.Conv2d, nn.ConvTranspose2d, nn.InstanceNorm2d, nn.Dropout2d, F.avg_pool2d
    return nn.Conv3d, nn.ConvTranspose3d, nn.InstanceNorm3d, nn.Dropout3d, F.avg_pool3d


class ConvBlock(nn.Module):
    """unet.py:128-182: [conv3 no-bias, InstanceNorm, LeakyReLU(0.2), Dropout] x 2."""

    def __init__(self, in_chans: int, out_chans: int, drop_prob: float, dims: int):
        super().__init__()
        conv, _, norm, drop, _ = _ops(dims)
        self.layers = nn.Sequential(
            conv(in_chans, out_chans, kernel_size=3, padding=1, bias=False),
            norm(out_chans),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
            drop(drop_prob),
            conv(out_chans, out_chans, kernel_size=3, padding=1, bias=False),
            norm(out_chans),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
            drop(drop_prob),
        )

    def forward(self, x):
        return self.layers(x)


class TransposeConvBlock(nn.Module):
    """unet.py:185-232: transpose conv k2 s2 no-bias, InstanceNorm, LeakyReLU(0.2)."""

    def __init__(self, in_chans: int, out_chans: int, dims: int):
        super().__init__()
        _, tconv, norm, _, _ = _ops(dims)
        self.layers = nn.Sequential(
            tconv(in_chans, out_chans, kernel_size=2, stride=2, bias=False),
            norm(out_chans),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
        )

    def forward(self, x):
        return self.layers(x)


class Unet(nn.Module):
    """unet.py:6-125."""

    def __init__(self, chans: int = 32, num_pool_layers: int = 4, in_chans: int = 2,
                 out_chans: int = 2, drop_prob: float = 0.0, dims: int = 2):
        super().__init__()
        conv = _ops(dims)[0]
        self.chans, self.num_pool_layers = chans, num_pool_layers
        self.in_chans, self.out_chans = in_chans, out_chans
        self.drop_prob, self.dims = drop_prob, dims

        # unet.py:51-56
        self.down_sample_layers = nn.ModuleList([ConvBlock(in_chans, chans, drop_prob, dims)])
        ch = chans
        for _ in range(num_pool_layers - 1):
            self.down_sample_layers.append(ConvBlock(ch, ch * 2, drop_prob, dims))
            ch *= 2
        self.conv = ConvBlock(ch, ch * 2, drop_prob, dims)

        # unet.py:58-71
        self.up_conv = nn.ModuleList()
        self.up_transpose_conv = nn.ModuleList()
        for _ in range(num_pool_layers - 1):
            self.up_transpose_conv.append(TransposeConvBlock(ch * 2, ch, dims))
            self.up_conv.append(ConvBlock(ch * 2, ch, drop_prob, dims))
            ch //= 2
        self.up_transpose_conv.append(TransposeConvBlock(ch * 2, ch, dims))
        self.up_conv.append(nn.Sequential(
            ConvBlock(ch * 2, ch, drop_prob, dims),
            conv(ch, out_chans, kernel_size=1, stride=1),
        ))

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        pool = _ops(self.dims)[4]
        skips = []
        x = image
        for layer in self.down_sample_layers:            # unet.py:94-97
            x = layer(x)
            skips.append(x)
            x = pool(x, kernel_size=2, stride=2, padding=0)
        x = self.conv(x)                                 # unet.py:99
        for up, conv in zip(self.up_transpose_conv, self.up_conv):
            skip = skips.pop()
            x = up(x)
            # unet.py:106-120: ZERO pad by one at the far end of every dim
            # whose size disagrees with the skip (comment there says reflect).
            pad = [0] * (2 * self.dims)
            for k in range(self.dims):
                if x.shape[-1 - k] != skip.shape[-1 - k]:
                    pad[2 * k + 1] = 1
            if any(pad):
                x = F.pad(x, pad)
            x = conv(torch.cat([x, skip], dim=1))        # unet.py:122-123
        return x


def _pad16(n: int) -> Tuple[int, List[int]]:
    mult = ((n - 1) | 15) + 1                            # norm_unet.py:80-81
    return mult, [math.floor((mult - n) / 2), math.ceil((mult - n) / 2)]


class NormUnet(nn.Module):
    """norm_unet.py:12-114."""

    def __init__(self, chans: int, num_pools: int, in_chans: int = 2, out_chans: int = 2,
                 drop_prob: float = 0.0):
        super().__init__()
        self.unet = Unet(in_chans=in_chans, out_chans=out_chans, chans=chans,
                         num_pool_layers=num_pools, drop_prob=drop_prob, dims=2)

    @staticmethod
    def norm(x):
        """norm_unet.py:59-69: per sample, per (re | im) group; UNBIASED std, no eps."""
        b, c = x.shape[:2]
        g = x.reshape(b, 2, -1)
        shape = (b, c) + (1,) * (x.dim() - 2)
        mean = g.mean(dim=2).view(shape)
        std = g.std(dim=2).view(shape)
        return (x - mean) / std, mean, std

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[-1] != 2:
            raise ValueError("Last dimension must be 2 for complex.")
        b, c, h, w, _ = x.shape
        x = x.permute(0, 4, 1, 2, 3).reshape(b, 2 * c, h, w)      # :48-51
        x, mean, std = self.norm(x)
        hm, hp = _pad16(h)
        wm, wp = _pad16(w)
        x = F.pad(x, wp + hp)                                     # :76-86
        x = self.unet(x)
        x = x[..., hp[0]:hm - hp[1], wp[0]:wm - wp[1]]            # :88-96
        x = x * std + mean                                        # :71-74
        return x.view(b, 2, c, h, w).permute(0, 2, 3, 4, 1).contiguous()   # :53-57


class NormUnet3D(nn.Module):
    """norm_unet.py:117-219: same with (t, h, w) volumes and Conv3d."""

    def __init__(self, chans: int, num_pools: int, in_chans: int = 2, out_chans: int = 2,
                 drop_prob: float = 0.0):
        super().__init__()
        self.unet = Unet(in_chans=in_chans, out_chans=out_chans, chans=chans,
                         num_pool_layers=num_pools, drop_prob=drop_prob, dims=3)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[-1] != 2:
            raise ValueError("Last dimension must be 2 for complex.")
        b, c, t, h, w, _ = x.shape
        x = x.permute(0, 5, 1, 2, 3, 4).reshape(b, 2 * c, t, h, w)     # :149-152
        x, mean, std = NormUnet.norm(x)                                # :160-170
        tm, tp = _pad16(t)
        hm, hp = _pad16(h)
        wm, wp = _pad16(w)
        x = F.pad(x, wp + hp + tp)                                     # :177-189
        x = self.unet(x)
        x = x[..., tp[0]:tm - tp[1], hp[0]:hm - hp[1], wp[0]:wm - wp[1]]
        x = x * std + mean
        return x.view(b, 2, c, t, h, w).permute(0, 2, 3, 4, 5, 1).contiguous()
