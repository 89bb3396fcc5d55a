"""Oracle: dynamic end-to-end VarNet (2D / 3D / XT / XF) on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference ``reconstruction/models/varnet.py``.  Module attribute
names follow the reference so its checkpoints load unchanged (including the
aliased ``cascades.N.model.*`` keys: one regulariser object is attached to
every cascade, varnet.py:138-140,171).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from . import centered_fft as cf
from . import complex_ops as co
from .regularisers import NormUnet, NormUnet3D


class SensitivityModel(nn.Module):
    """varnet.py:14-86."""

    def __init__(self, chans: int, num_pools: int, in_chans: int = 2, out_chans: int = 2,
                 drop_prob: float = 0.0):
        super().__init__()
        self.norm_unet = NormUnet(chans, num_pools, in_chans=in_chans,
                                  out_chans=out_chans, drop_prob=drop_prob)

    @staticmethod
    def acs_window(mask: torch.Tensor):
        """varnet.py:64-68.  Frame 0's mask only; rows are dim -3.  Returns
        (pad, num_low_freqs): keep rows [pad, pad + num_low_freqs)."""
        rows = mask[:, 0].squeeze()
        cent = mask.shape[-3] // 2
        left = int(torch.nonzero(rows[:cent] == 0)[-1])
        right = int(torch.nonzero(rows[cent:] == 0)[0]) + cent
        num_low = right - left
        pad = (mask.shape[-3] - num_low + 1) // 2
        return pad, num_low

    def forward(self, masked_kspace: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        pad, num_low = self.acs_window(mask)
        x = co.mask_center(masked_kspace.mean(dim=1), pad, pad + num_low)   # :71
        x = cf.ifft2c(x)                                                     # :74
        b, c, h, w, _ = x.shape
        x = self.norm_unet(x.reshape(b * c, 1, h, w, 2)).view(b, c, h, w, 2)  # :78-82
        x = x / co.rss_complex(x, dim=1).unsqueeze(-1).unsqueeze(1)          # :58-59
        return x.unsqueeze(1)                                                # :85


class VarNetBlock(nn.Module):
    """varnet.py:154-282."""

    def __init__(self, model: nn.Module, dynamic_type: str, weight_sharing: bool):
        super().__init__()
        self.model = model
        self.dynamic_type = dynamic_type
        self.weight_sharing = weight_sharing
        # varnet.py:176-179: softplus(lambda_0) = 1
        self.lambda_reg = nn.Parameter(torch.full((1,), math.log(math.e - 1.0)))

    @staticmethod
    def sens_expand(x, sens):
        """varnet.py:181-185."""
        return cf.fft2c(co.complex_mul(x, sens))

    @staticmethod
    def sens_reduce(k, sens):
        """varnet.py:187-194."""
        return co.complex_mul(cf.ifft2c(k), co.complex_conj(sens)).sum(dim=2, keepdim=True)

    def xfyf_transform(self, image: torch.Tensor) -> torch.Tensor:
        """varnet.py:196-241.  image: (b, t, h, w, 2)."""
        b, t, h, w, _ = image.shape
        mean = image.mean(dim=1, keepdim=True)                     # :205-206
        x = image - mean                                           # :207
        if self.dynamic_type == 'XF':                              # :209-213
            x = cf.fft1c(x.permute(0, 2, 3, 1, 4)).permute(0, 3, 1, 2, 4)
        xf = x.permute(0, 2, 3, 1, 4).reshape(b * h, 1, w, t, 2)   # :216
        yf = x.permute(0, 3, 2, 1, 4).reshape(b * w, 1, h, t, 2)   # :217
        if self.weight_sharing:
            xf, yf = self.model(xf), self.model(yf)                # :220-222
        else:
            xf, yf = self.model[0](xf), self.model[1](yf)          # :224-226
        xf = xf.view(b, h, 1, w, t, 2).permute(0, 4, 2, 1, 3, 5)   # :229
        yf = yf.view(b, w, 1, h, t, 2).permute(0, 4, 2, 3, 1, 5)   # :230
        out = 0.5 * (xf + yf)                                      # :232
        if self.dynamic_type == 'XF':                              # :234-238
            out = cf.ifft1c(out.permute(0, 2, 3, 4, 1, 5)).permute(0, 4, 1, 2, 3, 5)
        return out + mean.unsqueeze(2)                             # :241

    def regularise(self, image: torch.Tensor) -> torch.Tensor:
        """image (b, t, 1, h, w, 2) -> same; the dynamic-type switch of :255-278."""
        if self.dynamic_type in ('XF', 'XT'):
            return self.xfyf_transform(image.squeeze(2))
        if self.dynamic_type == '2D':
            return self.model(image.squeeze(0)).unsqueeze(0)       # :265-268 (b == 1)
        if self.dynamic_type == '3D':
            return self.model(image.permute(0, 2, 1, 3, 4, 5)).permute(0, 2, 1, 3, 4, 5)
        raise ValueError(self.dynamic_type)

    def forward(self, current_kspace, ref_kspace, mask, sens_maps):
        image = self.sens_reduce(current_kspace, sens_maps)        # :253
        model_term = self.sens_expand(self.regularise(image), sens_maps)
        v = F.softplus(self.lambda_reg, beta=1.0)                  # :281
        # :282 -- uint8 mask, (1 - mask) evaluated in uint8
        return (1 - mask) * model_term + mask * (model_term + v * ref_kspace) / (1 + v)


class VarNet(nn.Module):
    """varnet.py:91-151."""

    def __init__(self, num_cascades: int = 12, sens_chans: int = 8, sens_pools: int = 4,
                 chans: int = 18, pools: int = 4, dynamic_type: str = 'XF',
                 weight_sharing: bool = False):
        super().__init__()
        self.sens_net = SensitivityModel(sens_chans, sens_pools)
        if dynamic_type in ('XF', 'XT'):
            if weight_sharing:
                self.model = NormUnet(chans, pools)
            else:
                self.model = nn.ModuleList([NormUnet(chans, pools), NormUnet(chans, pools)])
        elif dynamic_type == '3D':
            self.model = NormUnet3D(chans, pools)
        else:
            self.model = NormUnet(chans, pools)
        self.cascades = nn.ModuleList(
            [VarNetBlock(self.model, dynamic_type, weight_sharing) for _ in range(num_cascades)])

    def forward(self, masked_kspace: torch.Tensor, mask: torch.Tensor,
                sens_maps: torch.Tensor = None) -> torch.Tensor:
        if sens_maps is None:
            sens_maps = self.sens_net(masked_kspace, mask)         # :144
        k = masked_kspace.clone()                                  # :145
        for cascade in self.cascades:                              # :147-148
            k = cascade(k, masked_kspace, mask, sens_maps)
        img = co.complex_mul(cf.ifft2c(k), co.complex_conj(sens_maps)).sum(dim=2)
        return co.complex_abs(img)                                 # :150-151
