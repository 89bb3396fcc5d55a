"""Oracle: the three convolutional-RNN hybrids (VarNet_RNN, CineNet_RNN, XPDNet_RNN) on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference ``reconstruction/models/recurrent_varnet.py``, ``recurrent_cinenet.py`` and
``recurrent_xpdnet.py``.  ``CRNNcell`` / ``BCRNNlayer`` are verbatim-identical in the three reference files
(recurrent_varnet.py:153-259) and are stated once here.  Attribute names follow the reference
(``bcrnn.CRNN_model.{i2h,h2h,ih2ih}``, ``conv{1,2,3}_{x,h}``, ``conv4_x``, ``lambda_reg``, ``sens_net``).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from . import centered_fft as cf
from . import complex_ops as co
from .varnet_ref import SensitivityModel as VarnetSens
from .xpdnet_ref import SensitivityModel as XpdSens, KSpaceCNN, forward_operator, backward_operator


class CRNNcell(nn.Module):
    """recurrent_varnet.py:153-200."""

    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        self.i2h = nn.Conv2d(input_size, hidden_size, kernel_size, padding=kernel_size // 2)
        self.h2h = nn.Conv2d(hidden_size, hidden_size, kernel_size, padding=kernel_size // 2)
        self.ih2ih = nn.Conv2d(hidden_size, hidden_size, kernel_size, padding=kernel_size // 2)

    def forward(self, x, hidden_iteration, hidden):
        return F.relu(self.i2h(x) + self.h2h(hidden) + self.ih2ih(hidden_iteration))


class BCRNNlayer(nn.Module):
    """recurrent_varnet.py:203-259: forward and backward passes over t with the SAME cell, summed."""

    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        self.hidden_size = hidden_size
        self.CRNN_model = CRNNcell(input_size, hidden_size, kernel_size)

    def forward(self, x, hidden_iteration):
        t, b, _, h, w = x.shape
        zero = x.new_zeros(b, self.hidden_size, h, w)
        fwd, bwd, hid = [], [], zero
        for i in range(t):
            hid = self.CRNN_model(x[i], hidden_iteration[i], hid)
            fwd.append(hid)
        hid = zero
        for i in range(t - 1, -1, -1):
            hid = self.CRNN_model(x[i], hidden_iteration[i], hid)
            bwd.append(hid)
        out = torch.cat(fwd) + torch.cat(bwd[::-1])
        return out.view(t, b, self.hidden_size, h, w)


class _CRNNBody(nn.Module):
    """The BCRNN + three (conv_x + conv_h, ReLU) layers + conv4 shared by all three models
    (recurrent_varnet.py:48-63, 116-136)."""

    def _make_body(self, in_ch, chans, out_ch):
        self.bcrnn = BCRNNlayer(input_size=in_ch, hidden_size=chans, kernel_size=3)
        for k in (1, 2, 3):
            setattr(self, f"conv{k}_x", nn.Conv2d(chans, chans, 3, padding=1))
            setattr(self, f"conv{k}_h", nn.Conv2d(chans, chans, 3, padding=1))
        self.conv4_x = nn.Conv2d(chans, out_ch, 3, padding=1)

    def _body(self, x, state):
        """x (t, b, ch, h, w); state = [x0, x1, x2, x3] of the previous cascade, each (t*b, chans, h, w)."""
        t, b, _, h, w = x.shape
        x0 = self.bcrnn(x, state[0].view(t, b, self.chans, h, w)).view(-1, self.chans, h, w)
        x1 = F.relu(self.conv1_h(state[1]) + self.conv1_x(x0))
        x2 = F.relu(self.conv2_h(state[2]) + self.conv2_x(x1))
        x3 = F.relu(self.conv3_h(state[3]) + self.conv3_x(x2))
        return self.conv4_x(x3), [x0, x1, x2, x3]

    def _zero_state(self, t, b, h, w, like):
        return [like.new_zeros(t * b, self.chans, h, w) for _ in range(4)]


def _soft_dc(k, kref, mask, v):
    return (1 - mask) * k + mask * (k + v * kref) / (1 + v)


class VarNet_RNN(_CRNNBody):
    """recurrent_varnet.py:13-150."""

    def __init__(self, num_cascades=12, sens_chans=8, sens_pools=4, chans=18):
        super().__init__()
        self.num_cascades, self.chans = num_cascades, chans
        self.sens_net = VarnetSens(sens_chans, sens_pools)
        self._make_body(2, chans, 2)
        self.lambda_reg = nn.Parameter(torch.full((1,), math.log(math.e - 1.0)))

    def forward(self, ref_kspace, mask):
        sens = self.sens_net(ref_kspace, mask)
        img = co.complex_mul(cf.ifft2c(ref_kspace), co.complex_conj(sens)).sum(dim=2)      # (b, t, h, w, 2)
        b, t, h, w, _ = img.shape
        state = self._zero_state(t, b, h, w, img)
        for _ in range(self.num_cascades):
            x = img.permute(1, 0, 4, 2, 3).contiguous()                                     # (t, b, 2, h, w)
            x4, state = self._body(x, state)
            out = (x.view(-1, 2, h, w) + x4).view(t, b, 2, h, w).permute(1, 0, 3, 4, 2)    # (b, t, h, w, 2)
            k = cf.fft2c(co.complex_mul(out.unsqueeze(2), sens))                            # :65-69
            dc = _soft_dc(k, ref_kspace, mask, F.softplus(self.lambda_reg))                 # :80-90
            img = co.complex_mul(cf.ifft2c(dc), co.complex_conj(sens)).sum(dim=2)
        return co.complex_abs(img)


class CineNet_RNN(_CRNNBody):
    """recurrent_cinenet.py:12-187."""

    def __init__(self, num_cascades=10, CG_iters=4, chans=64):
        super().__init__()
        self.num_cascades, self.CG_iters, self.chans = num_cascades, CG_iters, chans
        self._make_body(2, chans, 2)
        self.lambda_reg = nn.Parameter(torch.full((1,), math.log(math.e - 1.0)))

    def HOperator(self, x, mask, sens):
        k = cf.fft2c(co.complex_mul(x, sens)) * mask + 0.0
        return co.complex_mul(cf.ifft2c(k), co.complex_conj(sens)).sum(dim=2, keepdim=True) + F.softplus(self.lambda_reg) * x

    def ConjGrad(self, x, b, mask, sens, iters):
        r = b - self.HOperator(x, mask, sens)
        p = r.clone()
        rr = torch.dot(r.flatten(), r.flatten())
        for _ in range(iters):
            d = self.HOperator(p, mask, sens)
            alpha = rr / torch.dot(p.flatten(), d.flatten())
            x = torch.add(x, p, alpha=alpha.item())
            r = torch.add(r, d, alpha=-alpha.item())
            rr_new = torch.dot(r.flatten(), r.flatten())
            beta = rr_new / rr
            rr = rr_new
            p = torch.add(r, p, alpha=beta.item())
        return x

    def forward(self, ref_kspace, mask, sens_maps):
        x_ref = co.complex_mul(cf.ifft2c(ref_kspace), co.complex_conj(sens_maps)).sum(dim=2, keepdim=True)
        img = x_ref.squeeze(2)
        b, t, h, w, _ = img.shape
        state = self._zero_state(t, b, h, w, img)
        for _ in range(self.num_cascades):
            x = img.permute(1, 0, 4, 2, 3).contiguous()
            x4, state = self._body(x, state)
            out = (x.view(-1, 2, h, w) + x4).view(t, b, 2, h, w).permute(1, 0, 3, 4, 2).unsqueeze(2)
            out = self.ConjGrad(out, x_ref + F.softplus(self.lambda_reg) * out, mask, sens_maps, self.CG_iters)
            img = out.squeeze(2)
        return co.complex_abs(img)


class XPDNet_RNN(_CRNNBody):
    """recurrent_xpdnet.py:14-240."""

    def __init__(self, num_cascades=12, sens_chans=8, sens_pools=4, chans=18, primal_only=True, n_primal=5, n_dual=1):
        super().__init__()
        self.num_cascades, self.chans = num_cascades, chans
        self.i_buffer_size = n_primal
        self.k_buffer_mode = not primal_only
        self.k_buffer_size = 1 if primal_only else n_dual
        self.sens_net = XpdSens(sens_chans, sens_pools)
        self._make_body(2 * (n_primal + 1), chans, 2 * n_primal)
        if not primal_only:
            self.kspace_net = nn.ModuleList([KSpaceCNN(2 * (n_dual + 2), 2 * n_dual, 3, 16) for _ in range(num_cascades)])
        else:
            self.kspace_net = [self.measurements_residual for _ in range(num_cascades)]

    @staticmethod
    def measurements_residual(k):
        return torch.stack([k[..., 0], k[..., 2]], -1) - torch.stack([k[..., 1], k[..., 3]], -1)

    def forward(self, ref_kspace, mask):
        n = self.i_buffer_size
        sens = self.sens_net(ref_kspace, mask)
        image = backward_operator(ref_kspace, mask, sens, 1, False)
        kb = torch.repeat_interleave(ref_kspace, self.k_buffer_size, dim=-1)
        ib = torch.repeat_interleave(image, n, dim=-1)
        b, t, _, h, w, _ = ib.shape
        ch = 2 * (n + 1)
        state = self._zero_state(t, b, h, w, ib)
        for i in range(self.num_cascades):
            fwd = co.real_to_complex_multi_ch(forward_operator(ib, mask, sens, n, True), 1)
            cat = torch.cat([co.real_to_complex_multi_ch(kb, self.k_buffer_size), fwd], -1) if self.k_buffer_mode else fwd
            cat = torch.cat([cat, co.real_to_complex_multi_ch(ref_kspace, 1)], -1)
            kb = self.kspace_net[i](co.complex_to_real_multi_ch(cat))
            bwd = co.real_to_complex_multi_ch(backward_operator(kb, mask, sens, self.k_buffer_size, True), 1)
            ibx = co.complex_to_real_multi_ch(torch.cat([co.real_to_complex_multi_ch(ib, n), bwd], -1))
            x = ibx.squeeze(2).permute(1, 0, 4, 2, 3).contiguous()                          # (t, b, ch, h, w)
            x4, state = self._body(x, state)
            xv = x.view(-1, ch, h, w)
            out = torch.cat([xv[:, :n], xv[:, n + 1:-1]], dim=1) + x4                       # :226-231
            ib = out.view(t, b, 1, 2 * n, h, w).permute(1, 0, 2, 4, 5, 3)
        return co.complex_abs(torch.stack([ib[..., 0], ib[..., n]], -1).squeeze(2))
