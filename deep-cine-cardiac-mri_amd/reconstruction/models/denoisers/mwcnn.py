"""Multi-level wavelet CNN whose forward pass runs on the gfx950 conv kernels.

Same constructor, attribute tree and state-dict keys as the reference's denoisers/mwcnn.py (MWCNN :8,
ConvBlock :183, DWT :216, IWT :240).  ``forward`` hands packed weights to ``cine_mwcnn_forward``: the
Haar analysis / synthesis steps and the additive skips are folded into the operand staging of the
neighbouring 3x3 convolutions, so no wavelet-domain tensor is materialised.
"""
from typing import List, Tuple

import torch
from torch import nn

from cine_hip import ops


class ConvBlock(nn.Module):
    """Parameter holder: conv3 'same' no bias + InstanceNorm + LeakyReLU(0.2)."""

    def __init__(self, in_chans: int, n_filters: int, dims: int):
        super().__init__()
        conv, norm = (nn.Conv2d, nn.InstanceNorm2d) if dims == 2 else (nn.Conv3d, nn.InstanceNorm3d)
        self.layers = nn.Sequential(conv(in_chans, n_filters, kernel_size=3, padding='same', bias=False),
                                    norm(n_filters), nn.LeakyReLU(negative_slope=0.2, inplace=True))

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        conv = self.layers[0]
        y, part = ops.conv3x3_in([(inputs, None, 0)], ops.pack_conv3x3(conv.weight), conv.out_channels,
                                 inputs.shape[2], inputs.shape[3])
        return ops.instnorm_lrelu_apply(y, part)


class DWT(nn.Module):
    """Haar analysis (reference mwcnn.py:216-236); a pure re-indexing, device-agnostic."""

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        x01, x02 = inputs[:, :, 0::2] / 2, inputs[:, :, 1::2] / 2
        x1, x2, x3, x4 = x01[..., 0::2], x02[..., 0::2], x01[..., 1::2], x02[..., 1::2]
        return torch.cat([x1 + x2 + x3 + x4, -x1 - x2 + x3 + x4, -x1 + x2 - x3 + x4, x1 - x2 - x3 + x4], dim=1)


class IWT(nn.Module):
    """Haar synthesis (reference mwcnn.py:240-263) on the input's own device."""

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        b, ch, h, w = inputs.shape
        c = ch // 4
        x1, x2, x3, x4 = (inputs[:, k * c:(k + 1) * c] / 2 for k in range(4))
        out = torch.zeros([b, c, 2 * h, 2 * w], dtype=inputs.dtype, device=inputs.device)
        out[:, :, 0::2, 0::2] = x1 - x2 - x3 + x4
        out[:, :, 1::2, 0::2] = x1 - x2 + x3 - x4
        out[:, :, 0::2, 1::2] = x1 + x2 - x3 - x4
        out[:, :, 1::2, 1::2] = x1 + x2 + x3 + x4
        return out


class MWCNN(nn.Module):
    def __init__(self, in_chans: int, out_chans: int, dims: int = 2, n_scales: int = 3,
                 n_filters_per_scale: List[int] = [16, 32, 64], n_convs_per_scale: List[int] = [2, 2, 2],
                 n_first_convs: int = 1, first_conv_n_filters: int = 16, res: bool = False):
        super().__init__()
        self.in_chans, self.out_chans, self.dims, self.n_scales = in_chans, out_chans, dims, n_scales
        self.n_filters_per_scale, self.n_convs_per_scale = list(n_filters_per_scale), list(n_convs_per_scale)
        self.n_first_convs, self.first_conv_n_filters, self.res = n_first_convs, first_conv_n_filters, res
        assert self.dims in [2, 3], "Dimensions must be either 2 or 3"
        conv = nn.Conv2d if dims == 2 else nn.Conv3d
        if n_first_convs > 0:
            self.first_convs = nn.ModuleList([ConvBlock(in_chans, first_conv_n_filters, dims)])
            for _ in range(1, 2 * n_first_convs - 1):
                self.first_convs.append(ConvBlock(first_conv_n_filters, first_conv_n_filters, dims))
            self.first_convs.append(conv(first_conv_n_filters, out_chans, kernel_size=3, padding='same', bias=True))
        self.conv_blocks_per_scale = nn.ModuleList([
            nn.ModuleList([ConvBlock(*self.chans_for_conv_for_scale(s, i), dims)
                           for i in range(self.n_convs_per_scale[s] * 2)]) for s in range(n_scales)])
        if n_first_convs < 1:
            self.conv_blocks_per_scale[0][-1] = conv(self.n_filters_per_scale[0], 4 * out_chans, kernel_size=3,
                                                     padding='same', bias=True)
        self.pooling, self.unpooling = DWT(), IWT()
        self._hip = None

    def chans_for_conv_for_scale(self, i_scale: int, i_conv: int) -> Tuple[int, int]:
        in_chans = n_filters = self.n_filters_per_scale[i_scale]
        if i_conv == 0:
            in_chans = 4 * (self.first_conv_n_filters if i_scale == 0 else self.n_filters_per_scale[i_scale - 1])
        if i_conv == self.n_convs_per_scale[i_scale] * 2 - 1:
            n_filters = max(4 * self.first_conv_n_filters, 4 * self.out_chans) if i_scale == 0 \
                else 4 * self.n_filters_per_scale[i_scale - 1]
        return in_chans, n_filters

    def hip_weights(self):
        if self._hip is None:
            self._hip = ops.MwcnnWeights(self)
        return self._hip

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        if self.dims != 2:
            # constructible like the reference's (Conv3d parameter holders, same state-dict keys) -- but the reference's own forward cannot run with
            # dims = 3: its IWT unpacks FOUR dimensions (``b, ch, h, w = inputs.shape``, mwcnn.py:252), so a 5-D volume raises ValueError at the first
            # up-sampling step, and a 4-D (unbatched) input fails at the first conv behind the DWT (the sub-bands are concatenated along dim 1,
            # which is then the depth axis).  There is no reference behaviour to reproduce; XPDNet never builds it (xpdnet.py:251-262).
            raise NotImplementedError("MWCNN(dims=3).forward: the reference's forward raises for every input (its IWT unpacks four dimensions, "
                                      "mwcnn.py:252); only dims = 2 has a defined result")
        if torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters())):
            from cine_hip import autograd as ag          # training: forward keeps every feature map, backward = cine_mwcnn_backward
            return ag.mwcnn(inputs, self.hip_weights())
        return ops.mwcnn_forward(inputs, self.hip_weights())
