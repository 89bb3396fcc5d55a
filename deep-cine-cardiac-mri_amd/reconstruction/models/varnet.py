"""Dynamic end-to-end VarNet on the MI355X kernels.

Drop-in for the reference's models/varnet.py (SensitivityModel :14, VarNet :91,
VarNetBlock :154): same constructors, ``forward`` signatures, tensor layouts
(k-space (b,t,c,h,w,2) f32, mask (b,t,1,h,1,1) uint8) and state-dict keys,
including the aliased ``cascades.N.model.*`` entries.

What differs is where the arithmetic runs.  Per cascade the HIP path issues
  sens_reduce   : column IFFT pass + row IFFT pass fused with conj(S) x, coil sum
  regulariser   : temporal mean/DFT + x-f / y-f rotation + group norm (pack),
                  both U-Nets in one launch sequence, inverse (unpack)
  sens_expand+DC: row FFT pass fused with S x, column FFT pass fused with the soft DC
and never materialises shifted copies, stacked complex temporaries or normalised
activations.  GPU tensors only.

Training: when gradients are enabled and a parameter requires them, ``forward``
builds an autograd graph out of ``cine_hip.autograd`` Functions whose backward
passes are hand-written HIP kernels (input / weight gradients on the MFMA conv
kernels, InstanceNorm + LeakyReLU backward, the adjoint rotations and DFTs, the
self-adjoint image-space DC) -- what ``pl_modules/varnet_module.py:97-113``
(``training_step``: forward + SSIMLoss, then ``loss.backward()``) needs.  With
gradients disabled the inference path below runs unchanged.
"""
import math
from typing import Optional

import torch
from torch import nn

from cine_hip import autograd as ag
from cine_hip import ops
from .denoisers.norm_unet import NormUnet, NormUnet3D


class SensitivityModel(nn.Module):
    def __init__(self, chans: int, num_pools: int, in_chans: int = 2, out_chans: int = 2, drop_prob: float = 0.0):
        super().__init__()
        self.norm_unet = NormUnet(chans, num_pools, in_chans=in_chans, out_chans=out_chans, drop_prob=drop_prob)

    @staticmethod
    def acs_window(mask: torch.Tensor):
        """Rows [pad, pad + n_low) to keep (reference varnet.py:64-68: frame 0's mask only) as host integers: it reads the 1-D mask back to
        the host and so waits for the GPU.  ``forward`` without ``acs`` uses the device form (``ops.acs_window_dev``: same arithmetic, no wait)."""
        if mask.shape[-2] != 1:
            raise ValueError("the ACS window is read from a row mask (reference varnet.py:64-68 indexes the mask's h axis); with a mask that "
                             "varies along w pass acs=(pad, n_low) or sens_maps")
        rows = mask[:, 0].reshape(-1).cpu()
        cent = mask.shape[-3] // 2
        left = int(torch.nonzero(rows[:cent] == 0)[-1])
        right = int(torch.nonzero(rows[cent:] == 0)[0]) + cent
        n_low = right - left
        return (mask.shape[-3] - n_low + 1) // 2, n_low

    def forward(self, masked_kspace: torch.Tensor, mask: torch.Tensor, acs=None) -> torch.Tensor:
        if acs is None:      # the window is found on the device: no host read-back between the caller and the launches (acs_window is the host form)
            x = ops.sens_prologue(masked_kspace, ops.acs_window_dev(mask))      # (b, c, h, w, 2)
        else:
            pad, n_low = acs
            x = ops.sens_prologue(masked_kspace, pad, pad + n_low)
        b, c, h, w, _ = x.shape
        x = self.norm_unet(x.view(b * c, 1, h, w, 2)).view(b, c, h, w, 2)
        if x.requires_grad:
            return ag.RssNormFn.apply(x).unsqueeze(1)
        return ops.rss_normalise_(x).unsqueeze(1)


class VarNetBlock(nn.Module):
    def __init__(self, model: nn.Module, dynamic_type: str, weight_sharing: bool):
        super().__init__()
        self.model = model
        self.dynamic_type = dynamic_type
        self.weight_sharing = weight_sharing
        self.Softplus = nn.Softplus(1.)
        self.lambda_reg = nn.Parameter(torch.full((1,), math.log(math.e - 1.0)))   # softplus -> 1
        self._uw = None

    # -- pieces, same names as the reference ------------------------------------------------
    def sens_expand(self, x: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        return ops.sens_expand_dc(x, sens_maps)

    def sens_reduce(self, x: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        return ops.sens_reduce(x, sens_maps)

    def _xfyf_weights(self):
        # kept on the shared model object: every cascade holds the same networks (varnet.py:138-140), one set of packed weights serves all
        uw = self.model.__dict__.get("_hip_uw")
        if uw is None:
            nets = [self.model, self.model] if self.weight_sharing else [self.model[0], self.model[1]]
            uw = self.model.__dict__["_hip_uw"] = (ops.UnetWeights([nets[0].unet, nets[1].unet]),
                                                    ops.UnetWeights([nets[0].unet]), ops.UnetWeights([nets[1].unet]))
        return uw

    def xfyf_transform(self, image_combined: torch.Tensor) -> torch.Tensor:
        """(b, t, h, w, 2) -> (b, t, 1, h, w, 2)."""
        b, t, h, w, _ = image_combined.shape
        xf = self.dynamic_type == 'XF'
        both, wx, wy = self._xfyf_weights()
        if ag.grad_mode(self) or both.drops():       # (dropout is active in training mode even without autograd)
            return ag.xfyf(image_combined, xf, both, wx, wy)
        pxf, pyf, sxf, syf, mean = ops.xfyf_pack(image_combined, xf)
        if pxf.shape == pyf.shape and pxf.data_ptr() + pxf.numel() * 4 == pyf.data_ptr():
            joint = torch.as_strided(pxf, (2 * pxf.shape[0],) + tuple(pxf.shape[1:]), pxf.stride())
            out = ops.unet2d_forward(joint, both)
            oxf, oyf = out[:pxf.shape[0]], out[pxf.shape[0]:]
        else:
            oxf, oyf = ops.unet2d_forward(pxf, wx), ops.unet2d_forward(pyf, wy)
        return ops.xfyf_unpack(oxf, oyf, sxf, syf, mean, b, t, h, w, xf)

    def regularise(self, image_combined: torch.Tensor) -> torch.Tensor:
        """(b, t, 1, h, w, 2) -> same; the dynamic-type switch of reference varnet.py:255-278."""
        if self.dynamic_type in ['XF', 'XT']:
            return self.xfyf_transform(image_combined.squeeze(2))
        if self.dynamic_type == '2D':
            return self.model(image_combined.squeeze(0)).unsqueeze(0)
        if self.dynamic_type == '3D':
            return self.model(image_combined.permute(0, 2, 1, 3, 4, 5)).permute(0, 2, 1, 3, 4, 5).contiguous()
        raise ValueError(f"unknown dynamic_type {self.dynamic_type!r}")

    def forward(self, current_kspace, ref_kspace, mask, sens_maps, _destroy_current: bool = False):
        image = ops.sens_reduce(current_kspace, sens_maps, destroy_input=_destroy_current)
        model_out = self.regularise(image)
        out = current_kspace if _destroy_current else None
        if ops.is_general_mask(mask, ref_kspace):          # varies along w: the DC line of reference varnet.py:281-282 term by term
            return ops.soft_dc_blend(ops.sens_expand_dc(model_out, sens_maps, out=out), ref_kspace, mask, self.lambda_reg.detach())
        return ops.sens_expand_dc(model_out, sens_maps, ref_kspace, mask, self.lambda_reg, out=out)


class VarNet(nn.Module):
    def __init__(self, num_cascades: int = 12, sens_chans: int = 8, sens_pools: int = 4, chans: int = 18,
                 pools: int = 4, dynamic_type: str = 'XF', weight_sharing: bool = False):
        super().__init__()
        self.sens_net = SensitivityModel(sens_chans, sens_pools)
        if dynamic_type in ['XF', 'XT']:
            self.model = NormUnet(chans, pools) if weight_sharing else \
                nn.ModuleList([NormUnet(chans, pools), NormUnet(chans, pools)])
        elif dynamic_type == '3D':
            self.model = NormUnet3D(chans, pools)
        else:
            self.model = NormUnet(chans, pools)
        self.cascades = nn.ModuleList(
            [VarNetBlock(self.model, dynamic_type, weight_sharing) for _ in range(num_cascades)])

    def forward(self, masked_kspace: torch.Tensor, mask: torch.Tensor,
                sens_maps: Optional[torch.Tensor] = None, acs=None) -> torch.Tensor:
        """(b,t,c,h,w,2), (b,t,1,h,1,1) uint8 -> (b,t,h,w) magnitude.  ``sens_maps`` (optional,
        (b,1,c,h,w,2)) bypasses the sens-map network; ``acs`` = (pad, n_low) skips the host
        read-back of the mask (needed inside hipGraph capture).  Masks of another dtype (the reference's
        apply_mask returns a float mask) are converted once."""
        mask = ops.as_mask_u8(mask, masked_kspace)         # row mask (b,t,1,h,1,1), or general mask (b,t,1,h,w,1) when it varies along w
        if ag.grad_mode(self):
            return self._forward_train(masked_kspace, mask, sens_maps, acs)
        with torch.no_grad():
            return self._forward_infer(masked_kspace, mask, sens_maps, acs)

    def _forward_train(self, masked_kspace, mask, sens_maps, acs):
        """The image-space cascade chain of ``_forward_infer`` as an autograd graph (reference varnet.py:143-151)."""
        if sens_maps is None:
            sens_maps = self.sens_net(masked_kspace, mask, acs)
        if not ops.is_row_mask(mask, masked_kspace):
            # a mask that varies along w: the literal k-space chain of reference varnet.py:145-151 as an autograd graph -- coil
            # operators through their HIP kernels and adjoints (SensReduceFn / SensExpandFn), the DC line in torch elementwise ops
            kspace = masked_kspace
            for cascade in self.cascades:
                image = ag.SensReduceFn.apply(kspace, sens_maps, None)
                model_term = ag.SensExpandFn.apply(cascade.regularise(image), sens_maps, None)
                kspace = ops.soft_dc_blend(model_term, masked_kspace, mask, cascade.lambda_reg)
            return ag.AbsFn.apply(ag.SensReduceFn.apply(kspace, sens_maps, None).squeeze(2))
        image = ag.CoilReduceFn.apply(masked_kspace, sens_maps, None)          # first cascade's sens_reduce(masked_kspace)
        if len(self.cascades) == 0:
            return ag.AbsFn.apply(image.squeeze(2))
        zf = ag.CoilReduceFn.apply(masked_kspace, sens_maps, mask)             # sens_reduce(mask * k_ref)
        for cascade in self.cascades:
            image = ag.ImageDcFn.apply(cascade.regularise(image), sens_maps, zf, mask, cascade.lambda_reg)
        return ag.AbsFn.apply(image.squeeze(2))

    def _forward_infer(self, masked_kspace, mask, sens_maps, acs):
        if sens_maps is None:
            sens_maps = self.sens_net(masked_kspace, mask, acs)
        if not ops.is_row_mask(mask, masked_kspace):
            # not the reference's (b, t, 1, h, 1, 1) row mask: the literal k-space chain of reference varnet.py:145-151
            kspace = masked_kspace.clone()
            for cascade in self.cascades:
                kspace = cascade(kspace, masked_kspace, mask, sens_maps, _destroy_current=True)
            return ops.sens_reduce(kspace, sens_maps, magnitude=True, destroy_input=True)
        # Cascade chain on the coil-combined image.  The k-space between two cascades (reference varnet.py:147-148) is
        # consumed only by the next sens_reduce, and with a row mask the DC commutes with the transform along w, so
        #   reduce(DC(expand(x))) = sum_c conj(S_c) IFFT_h[(m ? 1/(1+v) : 1) FFT_h(S_c x)] + v/(1+v) reduce(m k_ref)
        # (cine_image_dc): per cascade the FFT+DC step reads x, S and the constant zero-filled term instead of making
        # three passes over the 72 MB coil-wise k-space.
        hyb = ops.kspace_to_hybrid(masked_kspace)
        image = ops.hybrid_reduce(hyb, sens_maps)                      # first cascade's sens_reduce(masked_kspace)
        if len(self.cascades) == 0:
            return ops.complex_abs(image.squeeze(2))
        ops.kspace_to_hybrid(masked_kspace, out=hyb, mask=mask)
        zf = ops.hybrid_reduce(hyb, sens_maps)                         # sens_reduce(mask * k_ref)
        last = len(self.cascades) - 1
        tiled = ops.sens_tile_pack(sens_maps)          # the maps in the order the DC kernel reads fastest: once per forward, for every cascade
        for i, cascade in enumerate(self.cascades):
            image = ops.image_dc(cascade.regularise(image), sens_maps, zf, mask, cascade.lambda_reg, magnitude=(i == last), sens_tiled=tiled)
        return image
