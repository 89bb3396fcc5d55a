"""Oracle: CineNet (U-Net regulariser + conjugate-gradient data consistency) on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference ``reconstruction/models/cinenet.py``; attribute names follow the reference
(``model[.{0,1}]`` plain Unets, ``cascades.N.lambda_reg`` and the aliased ``cascades.N.model.*``).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from . import centered_fft as cf
from . import complex_ops as co
from .regularisers import Unet


class CineNetBlock(nn.Module):
    """cinenet.py:77-257."""

    def __init__(self, model: nn.Module, CG_iters: int, dynamic_type: str, weight_sharing: bool):
        super().__init__()
        self.model, self.CG_iters = model, CG_iters
        self.dynamic_type, self.weight_sharing = dynamic_type, weight_sharing
        self.lambda_reg = nn.Parameter(torch.full((1,), math.log(math.e - 1.0)))     # :103-106

    @staticmethod
    def sens_expand(x, sens):                                   # :109-113
        return cf.fft2c(co.complex_mul(x, sens))

    @staticmethod
    def sens_reduce(k, sens):                                   # :115-122
        return co.complex_mul(cf.ifft2c(k), co.complex_conj(sens)).sum(dim=2, keepdim=True)

    def HOperator(self, x, mask, sens):
        """:121-133: A^H M A x + softplus(lambda) x."""
        k = self.sens_expand(x, sens) * mask + 0.0
        return self.sens_reduce(k, sens) + F.softplus(self.lambda_reg) * x

    def ConjGrad(self, x, b, mask, sens, iters):
        """:136-171: exactly `iters` iterations, real inner products over interleaved (re, im),
        step sizes round-trip through Python floats."""
        r = b - self.HOperator(x, mask, sens)
        p = r.clone()
        rr_old = torch.dot(r.flatten(), r.flatten())
        for _ in range(iters):
            d = self.HOperator(p, mask, sens)
            alpha = rr_old / torch.dot(p.flatten(), d.flatten())
            x = torch.add(x, p, alpha=alpha.item())
            r = torch.add(r, d, alpha=-alpha.item())
            rr_new = torch.dot(r.flatten(), r.flatten())
            beta = rr_new / rr_old
            rr_old = rr_new
            p = torch.add(r, p, alpha=beta.item())
        return x

    def xfyf_transform(self, image):
        """:173-219; image (b, t, h, w, 2)."""
        b, t, h, w, _ = image.shape
        mean = image.mean(dim=1, keepdim=True)
        x = image - mean
        if self.dynamic_type == 'XF':
            x = cf.fft1c(x.permute(0, 2, 3, 1, 4)).permute(0, 3, 1, 2, 4)
        xf = x.permute(0, 2, 4, 3, 1).reshape(b * h, 2, w, t)           # :194
        yf = x.permute(0, 3, 4, 2, 1).reshape(b * w, 2, h, t)           # :195
        if self.weight_sharing:
            xf, yf = self.model(xf), self.model(yf)
        else:
            xf, yf = self.model[0](xf), self.model[1](yf)
        xf = xf.view(b, h, 1, 2, w, t).permute(0, 5, 2, 1, 4, 3)        # :206
        yf = yf.view(b, w, 1, 2, h, t).permute(0, 5, 2, 4, 1, 3)        # :207
        out = 0.5 * (xf + yf)
        if self.dynamic_type == 'XF':
            out = cf.ifft1c(out.permute(0, 2, 3, 4, 1, 5)).permute(0, 4, 1, 2, 3, 5)
        return out + mean.unsqueeze(2)

    def regularise(self, image_pred):
        b, t, c, h, w, ch = image_pred.shape
        if self.dynamic_type in ('XF', 'XT'):
            return self.xfyf_transform(image_pred.squeeze(2))                       # :234
        if self.dynamic_type == '2D':                                               # :242-244
            x = image_pred.permute(0, 1, 2, 5, 3, 4).reshape(b * t, c * ch, h, w)
            return self.model(x).reshape(b, t, c, ch, h, w).permute(0, 1, 2, 4, 5, 3)
        if self.dynamic_type == '3D':                                               # :251-253
            x = image_pred.permute(0, 5, 2, 1, 3, 4).reshape(b, ch * c, t, h, w)
            return self.model(x).reshape(b, ch, c, t, h, w).permute(0, 3, 2, 4, 5, 1)
        raise ValueError(self.dynamic_type)

    def forward(self, image_pred, image_ref, mask, sens):
        out = self.regularise(image_pred)
        rhs = image_ref + F.softplus(self.lambda_reg) * out                         # :255-257
        return self.ConjGrad(out, rhs, mask, sens, self.CG_iters)


class CineNet(nn.Module):
    """cinenet.py:14-73."""

    def __init__(self, num_cascades: int = 12, CG_iters: int = 4, chans: int = 18, pools: int = 4,
                 dynamic_type: str = 'XF', weight_sharing: bool = False):
        super().__init__()
        if dynamic_type in ('XF', 'XT'):
            self.model = Unet(chans, pools, dims=2) if weight_sharing else \
                nn.ModuleList([Unet(chans, pools, dims=2), Unet(chans, pools, dims=2)])
        elif dynamic_type == '3D':
            self.model = Unet(chans, pools, dims=3)
        else:
            self.model = Unet(chans, pools, dims=2)
        self.cascades = nn.ModuleList(
            [CineNetBlock(self.model, CG_iters, dynamic_type, weight_sharing) for _ in range(num_cascades)])

    def forward(self, masked_kspace, mask, sens_maps):
        image = CineNetBlock.sens_reduce(masked_kspace, sens_maps)                  # :64-66
        ref = image.clone()
        for cascade in self.cascades:
            image = cascade(image, ref, mask, sens_maps)
        return co.complex_abs(image.squeeze(2))                                     # :73
