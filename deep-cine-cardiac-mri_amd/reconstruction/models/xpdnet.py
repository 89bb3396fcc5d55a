"""XPDNet (primal-only, MWCNN image nets) on the MI355X kernels.

Drop-in for the reference's models/xpdnet.py (SensitivityModel :17, ForwardOperator :104,
BackwardOperator :137, XPDNet :171, XPDNetBlock :330): same constructors, ``forward(masked_kspace, mask)``
and state-dict keys (``sens_net.unet_model.*``, ``image_net.N[.{0,1}].*`` and the aliased
``cascades.M.image_net.*``).  Per cascade the HIP path issues
  K step : S x0 -> row FFT; column FFT -> hard mask -> minus k_ref -> column IFFT (one kernel, hybrid space)
  I step : row IFFT + conj(S) + coil sum; buffer pack (temporal mean / XPDNet's own temporal transform /
           x-f, y-f rotation / left-heavy zero pad); two MWCNNs; unpack
and never materialises the k-space buffer.  ``primal_only=False`` adds the KSpaceCNN dual update (Conv3d on the MFMA kernel) and materialises
the k-space buffer.  GPU tensors only.  With gradients enabled (XT / XF / 2D, row masks; with or without the dual net) the forward builds an autograd graph of
``cine_hip.autograd`` Functions: the sensitivity network, the K step + masked backward operator (image-space, with respect to image and maps),
the I-step network (buffer pack / both MWCNNs / unpack) -- all with hand-written HIP backward kernels.
"""
from typing import Dict, List, Union

import torch
from torch import nn

from cine_hip import autograd as ag
from cine_hip import ops
from .denoisers.unet import Unet
from .denoisers.mwcnn import MWCNN
from .denoisers.kspace_net import KSpaceCNN
from .varnet import SensitivityModel as _VarnetSens


class SensitivityModel(nn.Module):
    def __init__(self, chans: int, num_pools: int, in_chans: int = 2, out_chans: int = 2, drop_prob: float = 0.0,
                 res_connection: bool = True):
        super().__init__()
        self.res_connection = res_connection
        self.unet_model = Unet(chans, num_pools, in_chans=in_chans, out_chans=out_chans, drop_prob=drop_prob)

    def forward(self, masked_kspace: torch.Tensor, mask: torch.Tensor, acs=None) -> torch.Tensor:
        if acs is None:      # the window is found on the device: no host read-back between the caller and the launches (acs_window is the host form)
            x = ops.sens_prologue(masked_kspace, ops.acs_window_dev(mask))      # (b, c, h, w, 2)
        else:
            pad, n_low = acs
            x = ops.sens_prologue(masked_kspace, pad, pad + n_low)
        b, c, h, w, _ = x.shape
        if ag.grad_mode(self):
            # unpack(unet(planes) + planes) = unpack(unet(planes)) + x: the repacks are linear and x is data (no gradient)
            y = ag.norm_unet(x.view(b * c, h, w, 2), self.unet_model.hip_weights(), norm=False)
            if self.res_connection:
                y = y + x.view(b * c, h, w, 2)
            return ag.RssNormFn.apply(y.view(b, c, h, w, 2)).unsqueeze(1)
        planes, _ = ops.normunet_pack(x.view(b * c, h, w, 2), norm=False)      # xpdnet.py:55-59
        y = self.unet_model(planes)
        if self.res_connection:
            y = ops.axpby_dev(y, planes, num=_one(y))                          # x += x_temp (:92-95)
        return ops.rss_normalise_(ops.normunet_unpack(y, None, h, w).view(b, c, h, w, 2)).unsqueeze(1)


class ForwardOperator(nn.Module):
    def __init__(self, masked: bool = False):
        super().__init__()
        self.masked = masked

    def forward(self, image, mask, sens_maps, buffer_size: int):
        img = ops.extract_complex(image, 0, buffer_size)
        return ops.sens_expand_dc(img, sens_maps, None, mask if self.masked else None, None, hard_mask=self.masked)


class BackwardOperator(nn.Module):
    def __init__(self, masked: bool = False):
        super().__init__()
        self.masked = masked

    def forward(self, kspace, mask, sens_maps, buffer_size: int):
        k = ops.extract_complex(kspace, 0, buffer_size)
        if self.masked:
            k = k * mask + 0.0          # tiny element-wise op; the fused path masks inside the column kernel
        return ops.sens_reduce(k, sens_maps)


class XPDNetBlock(nn.Module):
    def __init__(self, kspace_net, image_net, n_scales: int, dynamic_type: str, weight_sharing: bool,
                 buffer_kwargs: Dict[str, Union[bool, int]]):
        super().__init__()
        self.kspace_net, self.image_net = kspace_net, image_net
        self.n_scales, self.dynamic_type, self.weight_sharing = n_scales, dynamic_type, weight_sharing
        self.i_buffer_mode, self.k_buffer_mode = buffer_kwargs['i_buffer_mode'], buffer_kwargs['k_buffer_mode']
        self.i_buffer_size, self.k_buffer_size = buffer_kwargs['i_buffer_size'], buffer_kwargs['k_buffer_size']
        self.forward_op = ForwardOperator(masked=True)
        self.backward_op = BackwardOperator(masked=True)

    def regularise(self, i_domain: int, image_buffer: torch.Tensor, backward_img: torch.Tensor) -> torch.Tensor:
        """I-step network (reference xpdnet.py:424-446): buffer (b,t,1,h,w,2n) + backward image -> new buffer."""
        b, t, _, h, w, _ = image_buffer.shape
        n = self.i_buffer_size
        nets = self.image_net[i_domain // 2]
        if self.dynamic_type in ['XF', 'XT']:
            xf = self.dynamic_type == 'XF'
            net_x, net_y = (nets, nets) if self.weight_sharing else (nets[0], nets[1])
            if ag.grad_mode(self):
                if getattr(net_x, "dims", 2) != 2:
                    raise NotImplementedError("training through the 3-D MWCNN is not on the HIP path")
                return ag.xpd_regularise(image_buffer, backward_img, n, self.n_scales, xf, net_x.hip_weights(), net_y.hip_weights())
            pxf, pyf, mean = ops.xpd_pack(image_buffer, backward_img, n, self.n_scales, xf)
            if pxf.shape[1:] == pyf.shape[1:] and pxf.data_ptr() + pxf.numel() * 4 == pyf.data_ptr() and getattr(net_x, "dims", 2) == 2:
                # x-t and y-t planes of one shape, adjacent in memory: both networks in the same launches (two weight sets)
                joint = torch.as_strided(pxf, (pxf.shape[0] + pyf.shape[0],) + tuple(pxf.shape[1:]), pxf.stride())
                out = ops.mwcnn_forward(joint, net_x.hip_weights(), net_y.hip_weights(), pxf.shape[0])
                return ops.xpd_unpack(out[:pxf.shape[0]], out[pxf.shape[0]:], mean, b, t, h, w, n, self.n_scales, xf)
            return ops.xpd_unpack(net_x(pxf), net_y(pyf), mean, b, t, h, w, n, self.n_scales, xf)
        if self.dynamic_type == '2D':
            # (b, t, 1, h, w, 2(n+1)) channel-last -> (b*t, 2(n+1), h, w); no padding in 2-D mode (:442-444)
            cat = torch.cat([image_buffer[..., :n], backward_img[..., :1], image_buffer[..., n:], backward_img[..., 1:]], dim=-1)
            if ag.grad_mode(self):      # the frames are the MWCNN's batch; the module's forward dispatches to the HIP backward kernels
                planes = cat.reshape(b * t, h, w, 2 * (n + 1)).permute(0, 3, 1, 2).contiguous()
                return nets(planes).permute(0, 2, 3, 1).reshape(b, t, 1, h, w, 2 * n)
            planes = ops.chanlast_to_planes(cat.reshape(b * t, h, w, 2 * (n + 1)))
            return ops.planes_to_chanlast(nets(planes), h, w).view(b, t, 1, h, w, 2 * n)
        raise ValueError(f"unknown dynamic_type {self.dynamic_type!r}")


_ones = {}


def _one(like: torch.Tensor) -> torch.Tensor:
    t = _ones.get(like.device)
    if t is None:
        t = _ones[like.device] = torch.ones(1, device=like.device, dtype=torch.float32)
    return t


class XPDNet(nn.Module):
    def __init__(self, num_cascades: int = 12, sens_chans: int = 8, sens_pools: int = 4, n_scales: int = 3,
                 n_filters_per_scale: List[int] = [16, 32, 64], n_convs_per_scale: List[int] = [2, 2, 2],
                 n_first_convs: int = 1, first_conv_n_filters: int = 16, res: bool = False, primal_only: bool = True,
                 n_primal: int = 5, n_dual: int = 1, dynamic_type: str = 'XF', weight_sharing: bool = False):
        super().__init__()
        self.domain_sequence = 'KI' * num_cascades
        self.i_buffer_mode = True
        self.k_buffer_mode = not primal_only
        self.i_buffer_size = n_primal
        self.k_buffer_size = 1 if primal_only else n_dual
        self.n_scales, self.dynamic_type, self.weight_sharing = n_scales, dynamic_type, weight_sharing
        self.sens_net = SensitivityModel(sens_chans, sens_pools)
        self.backward_op = BackwardOperator(masked=False)
        if not primal_only:
            self.kspace_net = nn.ModuleList([KSpaceCNN(in_chans=2 * (n_dual + 2), out_chans=2 * n_dual, n_convs=3,
                                                       n_filters=16) for _ in range(num_cascades)])
        else:
            self.kspace_net = [self.measurements_residual for _ in range(num_cascades)]
        kw = dict(in_chans=2 * (n_primal + 1), out_chans=2 * n_primal, dims=2, n_scales=n_scales,
                  n_filters_per_scale=n_filters_per_scale, n_convs_per_scale=n_convs_per_scale,
                  n_first_convs=n_first_convs, first_conv_n_filters=first_conv_n_filters, res=res)
        if dynamic_type in ['XF', 'XT'] and not weight_sharing:
            self.image_net = nn.ModuleList([nn.ModuleList([MWCNN(**kw), MWCNN(**kw)]) for _ in range(num_cascades)])
        else:
            self.image_net = nn.ModuleList([MWCNN(**kw) for _ in range(num_cascades)])
        bk = {'i_buffer_mode': True, 'k_buffer_mode': self.k_buffer_mode, 'i_buffer_size': n_primal,
              'k_buffer_size': self.k_buffer_size}
        self.cascades = nn.ModuleList([XPDNetBlock(self.kspace_net, self.image_net, n_scales, dynamic_type,
                                                   weight_sharing, bk) for _ in range(len(self.domain_sequence))])

    @staticmethod
    def measurements_residual(concat_kspace: torch.Tensor) -> torch.Tensor:
        return concat_kspace[..., [0, 2]] - concat_kspace[..., [1, 3]]

    def forward(self, masked_kspace: torch.Tensor, mask: torch.Tensor, acs=None) -> torch.Tensor:
        mask = ops.as_mask_u8(mask, masked_kspace)          # any numeric 0 / 1 mask; broadcast along batch / time like the reference
        if ag.grad_mode(self):
            return self._forward_train(masked_kspace, mask, acs)
        with torch.no_grad():
            return self._forward_infer(masked_kspace, mask, acs)

    def _forward_train(self, masked_kspace, mask, acs):
        """The chain of ``_forward_infer`` (reference xpdnet.py:301-326) as an autograd graph.  Primal-only: the K step + masked backward
        operator is one image-space Function; with the KSpaceCNN dual net the k-space buffer is a learned quantity and the forward / backward
        operators are Functions with k-space gradients (FFT2 is unitary: each adjoint is the other operator)."""
        n, nd = self.i_buffer_size, self.k_buffer_size
        general = ops.is_general_mask(mask, masked_kspace)      # varies along w: the literal k-space chain (reference xpdnet.py:128-131 multiplies by any mask)
        if self.dynamic_type not in ['XF', 'XT', '2D'] or not (general or ops.is_row_mask(mask, masked_kspace)):
            raise NotImplementedError("training through the HIP path: dynamic_type XF / XT / 2D")
        mf = mask.to(masked_kspace.dtype) if general else None
        pick = lambda buf, k: torch.stack((buf[..., 0], buf[..., k]), dim=-1)
        sens_maps = self.sens_net(masked_kspace, mask, acs)
        image = ag.CoilReduceFn.apply(masked_kspace, sens_maps, None)           # unmasked backward op (:303)
        image_buffer = image.repeat_interleave(n, dim=-1)                        # (:307): [re x n, im x n]
        if self.k_buffer_mode:
            kbuf = masked_kspace.repeat_interleave(nd, dim=-1)                   # (:306)
        elif not general:
            zf = ag.CoilReduceFn.apply(masked_kspace, sens_maps, mask)          # A^H M k_ref
        for i_domain in range(1, len(self.domain_sequence), 2):
            x0 = pick(image_buffer, n)                                           # channel 0 of the buffer (:128)
            if self.k_buffer_mode:
                fwd = ag.SensExpandFn.apply(x0, sens_maps, None) * mf if general else ag.SensExpandFn.apply(x0, sens_maps, mask)      # M A x0 (:385-403)
                cat = torch.cat([kbuf[..., :nd], fwd[..., :1], masked_kspace[..., :1], kbuf[..., nd:], fwd[..., 1:], masked_kspace[..., 1:]], dim=-1)
                kbuf = self.kspace_net[i_domain // 2](cat)
                if general:
                    backward_img = ag.SensReduceFn.apply((pick(kbuf, nd) * mf).contiguous(), sens_maps, None)
                else:
                    backward_img = ag.SensReduceFn.apply(pick(kbuf, nd).contiguous(), sens_maps, mask)  # masked backward op (:161-167)
            elif general:
                backward_img = ag.masked_residual_backward(x0, sens_maps, masked_kspace, mask)          # A^H m (m A x0 - k_ref), literally
            else:
                backward_img = ag.ImageDcFixedFn.apply(x0, sens_maps, zf, mask, 1.0, 0.0, -1.0)         # A^H M (A x0 - k_ref)
            image_buffer = self.cascades[i_domain].regularise(i_domain, image_buffer, backward_img)
        return ag.AbsFn.apply(pick(image_buffer, n).squeeze(2))                   # (:321-326)

    def _forward_infer(self, masked_kspace, mask, acs):
        n = self.i_buffer_size
        sens_maps = self.sens_net(masked_kspace, mask, acs)
        image = ops.sens_reduce(masked_kspace, sens_maps)                       # unmasked backward op (:303)
        image_buffer = ops.repeat_complex(image, n)                             # (:307)
        rowmask = ops.is_row_mask(mask, masked_kspace) and not self.k_buffer_mode
        general = ops.is_general_mask(mask, masked_kspace)      # varies along w (reference xpdnet.py:128-131 multiplies by any broadcastable mask)
        hyb = None if (rowmask or general) else torch.empty_like(masked_kspace)
        if rowmask:     # A^H M k_ref, constant over the cascades: the K + backward step becomes A^H M A x0 - zf in one kernel
            zf = ops.hybrid_reduce(ops.kspace_to_hybrid(masked_kspace, mask=mask), sens_maps)
        nd = self.k_buffer_size
        kbuf = ops.repeat_complex(masked_kspace, nd) if self.k_buffer_mode else None          # (:306)
        tiled = ops.sens_tile_pack(sens_maps) if rowmask else None              # the maps as the DC kernel reads them fastest, once per forward
        for i_domain in range(1, len(self.domain_sequence), 2):                 # each 'K' then 'I' pair (:310-319)
            x0 = ops.extract_complex(image_buffer, 0, n)                        # channel 0 of the buffer (:128)
            if self.k_buffer_mode:
                # dual buffer: the k-space net needs the whole k-space, so it is materialised (:385-403)
                fwd = ops.sens_expand_dc(x0, sens_maps) * mask + 0.0 if general else ops.sens_expand_dc(x0, sens_maps, None, mask, None, hard_mask=True)
                cat = torch.cat([kbuf[..., :nd], fwd[..., :1], masked_kspace[..., :1],
                                 kbuf[..., nd:], fwd[..., 1:], masked_kspace[..., 1:]], dim=-1)
                kbuf = self.kspace_net[i_domain // 2](cat).contiguous()
                k0 = ops.extract_complex(kbuf, 0, nd) * mask + 0.0              # masked backward op (:161-167)
                backward_img = ops.sens_reduce(k0, sens_maps)
            elif rowmask:
                backward_img = ops.image_dc(x0, sens_maps, zf, mask, weights=(1.0, 0.0, -1.0), sens_tiled=tiled)   # A^H M (A x0 - k_ref)
            elif general:
                backward_img = ops.masked_residual_backward(x0, sens_maps, masked_kspace, mask)
            else:
                ops.expand_resid_hybrid(x0, sens_maps, masked_kspace, mask, out=hyb)    # K: M A x0 - k_ref
                backward_img = ops.hybrid_reduce(hyb, sens_maps)                # I: masked backward op
            image_buffer = self.cascades[i_domain].regularise(i_domain, image_buffer, backward_img)
        return ops.complex_abs(ops.extract_complex(image_buffer, 0, n).squeeze(2))      # (:321-326)
