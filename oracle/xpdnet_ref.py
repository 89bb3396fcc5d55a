"""Oracle: XPDNet with MWCNN image nets (CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference ``reconstruction/models/xpdnet.py``, ``denoisers/mwcnn.py``,
``denoisers/kspace_net.py`` and ``utils/padding.py``.  Module/attribute names follow the reference so
its state dicts load unchanged (including the ``cascades.M.image_net.*`` aliases, xpdnet.py:283-292).
"""
from typing import List

import torch
from torch import nn
from torch.nn import functional as F

from . import centered_fft as cf
from . import complex_ops as co
from .regularisers import Unet
from .varnet_ref import SensitivityModel as _VarnetSens


# ------------------------------------------------------------------ utils/padding.py
def pad_for_mwcnn(x: torch.Tensor, n_scales: int):
    """padding.py:7-49 -- multiple of 2^n_scales on the last two dims; odd sizes get the extra on the left."""
    if x.dim() < 2:
        raise ValueError("Number of dimensions cannot be less than 2")
    m = 2 ** n_scales
    pads = []
    for d in (x.shape[-1], x.shape[-2]):
        n_pad = 0 if d % m == 0 else (d // m + 1) * m - d
        left = n_pad // 2 if (d % 2 == 0 or n_pad == 0) else 1 + n_pad // 2
        pads += [left, n_pad // 2]
    return F.pad(x, pads), pads


def unpad_from_mwcnn(x: torch.Tensor, pad: List[int]) -> torch.Tensor:
    """padding.py:53-70."""
    h, w = x.shape[-2], x.shape[-1]
    return x[..., pad[2]:h - pad[3], pad[0]:w - pad[1]]


# ------------------------------------------------------------------ denoisers/mwcnn.py
class DWT(nn.Module):
    """mwcnn.py:216-236."""

    def forward(self, x):
        x01, x02 = x[:, :, 0::2] / 2, x[:, :, 1::2] / 2
        x1, x2, x3, x4 = x01[..., 0::2], x02[..., 0::2], x01[..., 1::2], x02[..., 1::2]
        return torch.cat([x1 + x2 + x3 + x4, -x1 - x2 + x3 + x4, -x1 + x2 - x3 + x4, x1 - x2 - x3 + x4], dim=1)


class IWT(nn.Module):
    """mwcnn.py:240-263 (device-agnostic)."""

    def forward(self, x):
        b, ch, h, w = x.shape
        c = ch // 4
        x1, x2, x3, x4 = (x[:, k * c:(k + 1) * c] / 2 for k in range(4))
        out = torch.zeros(b, c, 2 * h, 2 * w, dtype=x.dtype, device=x.device)
        out[:, :, 0::2, 0::2] = x1 - x2 - x3 + x4
        out[:, :, 1::2, 0::2] = x1 - x2 + x3 - x4
        out[:, :, 0::2, 1::2] = x1 + x2 - x3 - x4
        out[:, :, 1::2, 1::2] = x1 + x2 + x3 + x4
        return out


class WConvBlock(nn.Module):
    """mwcnn.py:183-212: conv3 'same' no bias, InstanceNorm, LeakyReLU(0.2)."""

    def __init__(self, in_chans: int, n_filters: int, dims: int = 2):
        super().__init__()
        conv, norm = (nn.Conv2d, nn.InstanceNorm2d) if dims == 2 else (nn.Conv3d, nn.InstanceNorm3d)
        self.layers = nn.Sequential(conv(in_chans, n_filters, kernel_size=3, padding='same', bias=False),
                                    norm(n_filters), nn.LeakyReLU(negative_slope=0.2, inplace=True))

    def forward(self, x):
        return self.layers(x)


class MWCNN(nn.Module):
    """mwcnn.py:8-179."""

    def __init__(self, in_chans, out_chans, dims=2, n_scales=3, n_filters_per_scale=(16, 32, 64),
                 n_convs_per_scale=(2, 2, 2), n_first_convs=1, first_conv_n_filters=16, res=False):
        super().__init__()
        self.in_chans, self.out_chans, self.dims, self.n_scales = in_chans, out_chans, dims, n_scales
        self.n_filters_per_scale, self.n_convs_per_scale = list(n_filters_per_scale), list(n_convs_per_scale)
        self.n_first_convs, self.first_conv_n_filters, self.res = n_first_convs, first_conv_n_filters, res
        conv = nn.Conv2d if dims == 2 else nn.Conv3d
        if n_first_convs > 0:                                                    # :64-83
            self.first_convs = nn.ModuleList([WConvBlock(in_chans, first_conv_n_filters, dims)])
            for _ in range(1, 2 * n_first_convs - 1):
                self.first_convs.append(WConvBlock(first_conv_n_filters, first_conv_n_filters, dims))
            self.first_convs.append(conv(first_conv_n_filters, out_chans, kernel_size=3, padding='same', bias=True))
        self.conv_blocks_per_scale = nn.ModuleList([
            nn.ModuleList([WConvBlock(*self.chans_for_conv_for_scale(s, i), dims)
                           for i in range(self.n_convs_per_scale[s] * 2)]) for s in range(n_scales)])
        if n_first_convs < 1:                                                    # :96-104
            self.conv_blocks_per_scale[0][-1] = conv(self.n_filters_per_scale[0], 4 * out_chans, kernel_size=3,
                                                     padding='same', bias=True)
        self.pooling, self.unpooling = DWT(), IWT()

    def chans_for_conv_for_scale(self, s, i):
        """:110-132."""
        cin = cout = self.n_filters_per_scale[s]
        if i == 0:
            cin = 4 * (self.first_conv_n_filters if s == 0 else self.n_filters_per_scale[s - 1])
        if i == self.n_convs_per_scale[s] * 2 - 1:
            cout = max(4 * self.first_conv_n_filters, 4 * self.out_chans) if s == 0 else 4 * self.n_filters_per_scale[s - 1]
        return cin, cout

    def forward(self, x):
        """:135-179."""
        feats, cur = [], x
        if self.n_first_convs > 0:
            for c in self.first_convs[:self.n_first_convs]:
                cur = c(cur)
            first = cur
        for s in range(self.n_scales):
            cur = self.pooling(cur)
            for c in self.conv_blocks_per_scale[s][:self.n_convs_per_scale[s]]:
                cur = c(cur)
            feats.append(cur)
        for s in range(self.n_scales - 1, -1, -1):
            if s != self.n_scales - 1:
                cur = self.unpooling(cur) + feats[s]
            for c in self.conv_blocks_per_scale[s][self.n_convs_per_scale[s]:]:
                cur = c(cur)
        cur = self.unpooling(cur)
        if self.n_first_convs > 0:
            cur = cur + first
            for c in self.first_convs[self.n_first_convs:]:
                cur = c(cur)
        return x + cur if self.res else cur


class KSpaceCNN(nn.Module):
    """kspace_net.py:6-60: Conv3d(3^3, 'same', bias) + ReLU stack over (t, h, w), coils as batch."""

    def __init__(self, in_chans, out_chans, n_convs=3, n_filters=16):
        super().__init__()
        self.out_chans = out_chans
        convs = [nn.Conv3d(in_chans, n_filters, 3, padding='same'), nn.ReLU(inplace=True)]
        for _ in range(1, n_convs - 1):
            convs += [nn.Conv3d(n_filters, n_filters, 3, padding='same'), nn.ReLU(inplace=True)]
        convs.append(nn.Conv3d(n_filters, out_chans, 3, padding='same'))
        self.layers = nn.Sequential(*convs)

    def forward(self, x):
        b, t, c, h, w, ch = x.shape
        y = self.layers(x.permute(0, 2, 5, 1, 3, 4).reshape(b * c, ch, t, h, w))
        return y.reshape(b, c, self.out_chans, t, h, w).permute(0, 3, 1, 4, 5, 2)


# ------------------------------------------------------------------ models/xpdnet.py
class SensitivityModel(nn.Module):
    """xpdnet.py:17-100: plain Unet on (b*c, 2, h, w) + input residual, then / RSS."""

    def __init__(self, chans, num_pools, in_chans=2, out_chans=2, drop_prob=0.0, res_connection=True):
        super().__init__()
        self.res_connection = res_connection
        self.unet_model = Unet(chans, num_pools, in_chans=in_chans, out_chans=out_chans, drop_prob=drop_prob)

    def forward(self, masked_kspace, mask):
        pad, num_low = _VarnetSens.acs_window(mask)                               # :75-79
        x = cf.ifft2c(co.mask_center(masked_kspace.mean(dim=1), pad, pad + num_low))
        b, c, h, w, _ = x.shape
        x = x.view(b * c, h, w, 2).permute(0, 3, 1, 2)                            # :55-59
        y = self.unet_model(x)
        if self.res_connection:
            y = y + x                                                             # :92-95
        y = y.view(b, c, 2, h, w).permute(0, 1, 3, 4, 2)
        y = y / co.rss_complex(y, dim=1).unsqueeze(-1).unsqueeze(1)
        return y.unsqueeze(1)


def forward_operator(image, mask, sens, buffer_size, masked):
    """xpdnet.py:104-133."""
    img = torch.stack([image[..., 0], image[..., buffer_size]], dim=-1)
    k = cf.fft2c(co.complex_mul(img, sens))
    return k * mask + 0.0 if masked else k


def backward_operator(kspace, mask, sens, buffer_size, masked):
    """xpdnet.py:137-167."""
    k = torch.stack([kspace[..., 0], kspace[..., buffer_size]], dim=-1)
    if masked:
        k = k * mask + 0.0
    return co.complex_mul(cf.ifft2c(k), co.complex_conj(sens)).sum(dim=2, keepdim=True)


class XPDNetBlock(nn.Module):
    """xpdnet.py:330-542."""

    def __init__(self, kspace_net, image_net, n_scales, dynamic_type, weight_sharing, buffer_kwargs):
        super().__init__()
        self.kspace_net, self.image_net = kspace_net, image_net
        self.n_scales, self.dynamic_type, self.weight_sharing = n_scales, dynamic_type, weight_sharing
        self.i_buffer_mode, self.k_buffer_mode = buffer_kwargs['i_buffer_mode'], buffer_kwargs['k_buffer_mode']
        self.i_buffer_size, self.k_buffer_size = buffer_kwargs['i_buffer_size'], buffer_kwargs['k_buffer_size']

    def k_domain_correction(self, i_domain, image_buffer, kspace_buffer, mask, sens, ref_kspace):
        """:372-403."""
        fwd = co.real_to_complex_multi_ch(forward_operator(image_buffer, mask, sens, self.i_buffer_size, True), 1)
        if self.k_buffer_mode:
            kb = torch.cat([co.real_to_complex_multi_ch(kspace_buffer, self.k_buffer_size), fwd], dim=-1)
        else:
            kb = fwd
        kb = torch.cat([kb, co.real_to_complex_multi_ch(ref_kspace, 1)], dim=-1)
        return self.kspace_net[i_domain // 2](co.complex_to_real_multi_ch(kb))

    def i_domain_correction(self, i_domain, image_buffer, kspace_buffer, mask, sens):
        """:406-446."""
        bwd = co.real_to_complex_multi_ch(backward_operator(kspace_buffer, mask, sens, self.k_buffer_size, True), 1)
        ib = torch.cat([co.real_to_complex_multi_ch(image_buffer, self.i_buffer_size), bwd], dim=-1)
        ib = co.complex_to_real_multi_ch(ib)
        b, t, c, h, w, ch = ib.shape
        ch_out = 2 * self.i_buffer_size
        if self.dynamic_type in ('XF', 'XT'):
            return self.xfyf_transform(ib.squeeze(2), i_domain)
        x = ib.permute(0, 1, 2, 5, 3, 4).reshape(b * t, c * ch, h, w)            # :442-444 (no padding in 2D)
        return self.image_net[i_domain // 2](x).reshape(b, t, c, ch_out, h, w).permute(0, 1, 2, 4, 5, 3)

    def xfyf_transform(self, ib, i_domain):
        """:449-509."""
        b, t, h, w, ch = ib.shape
        n = self.i_buffer_size
        mean = ib.mean(dim=1, keepdim=True)
        x = ib - mean
        if self.dynamic_type == 'XF':
            x = co.complex_to_real_multi_ch(cf.xpd_temporal_fft(co.real_to_complex_multi_ch(x, n + 1), dim=1))
        xf = x.permute(0, 2, 4, 3, 1).reshape(b * h, ch, w, t)
        yf = x.permute(0, 3, 4, 2, 1).reshape(b * w, ch, h, t)
        xf, pxf = pad_for_mwcnn(xf, self.n_scales)
        yf, pyf = pad_for_mwcnn(yf, self.n_scales)
        nets = self.image_net[i_domain // 2]
        xf, yf = (nets(xf), nets(yf)) if self.weight_sharing else (nets[0](xf), nets[1](yf))
        xf, yf = unpad_from_mwcnn(xf, pxf), unpad_from_mwcnn(yf, pyf)
        xf = xf.reshape(b, h, 1, 2 * n, w, t).permute(0, 5, 2, 1, 4, 3)
        yf = yf.reshape(b, w, 1, 2 * n, h, t).permute(0, 5, 2, 4, 1, 3)
        out = 0.5 * (xf + yf)
        if self.dynamic_type == 'XF':
            out = co.complex_to_real_multi_ch(cf.xpd_temporal_ifft(co.real_to_complex_multi_ch(out, n), dim=1))
        m = mean.unsqueeze(2)
        return out + torch.cat([m[..., :n], m[..., n + 1:-1]], dim=-1)            # :504-509

    def forward(self, domain, i_domain, image_buffer, kspace_buffer, ref_kspace, mask, sens):
        if domain == 'K':
            kspace_buffer = self.k_domain_correction(i_domain, image_buffer, kspace_buffer, mask, sens, ref_kspace)
        if domain == 'I':
            image_buffer = self.i_domain_correction(i_domain, image_buffer, kspace_buffer, mask, sens)
        return image_buffer, kspace_buffer


class XPDNet(nn.Module):
    """xpdnet.py:171-326."""

    def __init__(self, num_cascades=12, sens_chans=8, sens_pools=4, n_scales=3, n_filters_per_scale=(16, 32, 64),
                 n_convs_per_scale=(2, 2, 2), n_first_convs=1, first_conv_n_filters=16, res=False, primal_only=True,
                 n_primal=5, n_dual=1, dynamic_type='XF', weight_sharing=False):
        super().__init__()
        self.domain_sequence = 'KI' * num_cascades
        self.i_buffer_size = n_primal
        self.k_buffer_mode = not primal_only
        self.k_buffer_size = 1 if primal_only else n_dual
        self.n_scales, self.dynamic_type, self.weight_sharing = n_scales, dynamic_type, weight_sharing
        self.sens_net = SensitivityModel(sens_chans, sens_pools)
        if not primal_only:
            self.kspace_net = nn.ModuleList([KSpaceCNN(2 * (n_dual + 2), 2 * n_dual, 3, 16) for _ in range(num_cascades)])
        else:
            self.kspace_net = [self.measurements_residual for _ in range(num_cascades)]
        kw = dict(in_chans=2 * (n_primal + 1), out_chans=2 * n_primal, dims=2, n_scales=n_scales,
                  n_filters_per_scale=n_filters_per_scale, n_convs_per_scale=n_convs_per_scale,
                  n_first_convs=n_first_convs, first_conv_n_filters=first_conv_n_filters, res=res)
        if dynamic_type in ('XF', 'XT') and not weight_sharing:
            self.image_net = nn.ModuleList([nn.ModuleList([MWCNN(**kw), MWCNN(**kw)]) for _ in range(num_cascades)])
        else:
            self.image_net = nn.ModuleList([MWCNN(**kw) for _ in range(num_cascades)])
        bk = dict(i_buffer_mode=True, k_buffer_mode=self.k_buffer_mode, i_buffer_size=n_primal, k_buffer_size=self.k_buffer_size)
        self.cascades = nn.ModuleList([XPDNetBlock(self.kspace_net, self.image_net, n_scales, dynamic_type, weight_sharing, bk)
                                       for _ in range(len(self.domain_sequence))])

    @staticmethod
    def measurements_residual(concat_kspace):
        """:295-298."""
        cur = torch.stack([concat_kspace[..., 0], concat_kspace[..., 2]], dim=-1)
        ref = torch.stack([concat_kspace[..., 1], concat_kspace[..., 3]], dim=-1)
        return cur - ref

    def forward(self, masked_kspace, mask):
        sens = self.sens_net(masked_kspace, mask)
        image = backward_operator(masked_kspace, mask, sens, 1, False)
        kb = torch.repeat_interleave(masked_kspace, self.k_buffer_size, dim=-1)
        ib = torch.repeat_interleave(image, self.i_buffer_size, dim=-1)
        for i, dom in enumerate(self.domain_sequence):
            ib, kb = self.cascades[i](dom, i, ib, kb, masked_kspace, mask, sens)
        out = torch.stack([ib[..., 0], ib[..., self.i_buffer_size]], dim=-1)
        return co.complex_abs(out.squeeze(2))
