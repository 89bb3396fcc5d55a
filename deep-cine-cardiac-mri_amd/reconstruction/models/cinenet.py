"""CineNet on the MI355X kernels: U-Net regulariser + conjugate-gradient data consistency.

Drop-in for the reference's models/cinenet.py (CineNet :14, CineNetBlock :77): same constructor,
``forward(masked_kspace, mask, sens_maps)`` signature and state-dict keys.  The normal operator
A^H M A (HOperator :121-133) is one row-FFT kernel, one fused column FFT -> hard mask -> column IFFT
kernel and one row-IFFT + coil-sum kernel; the CG scalars (alpha, beta, :159-169) stay in device
memory, so the solve has no host synchronisation and can be captured in a hipGraph.
GPU tensors only.  With gradients enabled the forward builds an autograd graph of ``cine_hip.autograd``
Functions: the bare U-Nets through the HIP backward kernels, the conjugate-gradient solve through its adjoint recurrence
(the reference detaches the step sizes, cinenet.py:159-169), lambda_reg through both.
"""
import math

import torch
from torch import nn

from cine_hip import autograd as ag
from cine_hip import ops
from .denoisers.unet import Unet


class CineNetBlock(nn.Module):
    def __init__(self, model: nn.Module, CG_iters: int, dynamic_type: str, weight_sharing: bool):
        super().__init__()
        self.model = model
        self.CG_iters = CG_iters
        self.dynamic_type = dynamic_type
        self.weight_sharing = weight_sharing
        self.Softplus = nn.Softplus(1.)
        self.lambda_reg = nn.Parameter(torch.full((1,), math.log(math.e - 1.0)))
        self._uw = None

    def sens_expand(self, x, sens_maps):
        return ops.sens_expand_dc(x, sens_maps)

    def sens_reduce(self, x, sens_maps):
        return ops.sens_reduce(x, sens_maps)

    def HOperator(self, x, mask, sens_maps, _hyb=None, _tiled=None):
        """A^H M A x + softplus(lambda) x  (reference cinenet.py:121-133).  With the reference's row mask the normal
        operator is one image-space kernel (cine_image_dc with weights (1, 0, 0)): the mask commutes with the row FFT."""
        return ops.h_operator(x, sens_maps, mask, self.lambda_reg, _hyb, _tiled)

    def ConjGrad(self, x, b, mask, sens_maps, CG_iters: int, _tiled=None, _inplace=False):
        """Hx = b with exactly CG_iters iterations (reference cinenet.py:136-171)."""
        bsz, t, _, h, w, _ = x.shape
        rowmask = ops.is_row_mask(mask, sens_maps.expand(-1, t, -1, -1, -1, -1))
        if rowmask and ops.FUSED_CG and ops.CG_SOLVER:
            # the whole solve in 2 + 2 * CG_iters launches (cine_conj_grad): the direction update rides in the next operator's loads
            out = ops.conj_grad(x if _inplace else x.clone(), b, sens_maps, mask, self.lambda_reg, CG_iters, _tiled)
            if out is not None:
                return out
        hyb = None if rowmask else torch.empty((bsz, t, sens_maps.shape[2], h, w, 2), device=x.device, dtype=x.dtype)
        r = ops.axpby_dev(b, self.HOperator(x, mask, sens_maps, hyb, _tiled), num=_one(x), sign=-1.0)
        p = r.clone()
        rr_old = ops.dot(r, r)
        x = x.clone()
        rr_new = torch.empty_like(rr_old)
        for _ in range(CG_iters):
            # d = H p; alpha = rr / p.d; x += alpha p; r -= alpha d; rr' = r.r; p = r + (rr' / rr) p   (:153-169), scalars on the device
            if rowmask:     # the p.d partial sums come out of the operator's own last kernel: four launches per iteration
                ops.normal_op_cg_step(x, r, p, sens_maps, mask, self.lambda_reg, rr_old, rr_new, sens_tiled=_tiled)
            else:
                ops.cg_step(x, r, p, self.HOperator(p, mask, sens_maps, hyb), rr_old, rr_new)
            rr_old, rr_new = rr_new, rr_old
        return x

    def _xfyf_weights(self):
        # kept on the shared model object: every cascade holds the same networks, one set of packed weights serves all
        uw = self.model.__dict__.get("_hip_uw")
        if uw is None:
            nets = [self.model, self.model] if self.weight_sharing else [self.model[0], self.model[1]]
            uw = self.model.__dict__["_hip_uw"] = (ops.UnetWeights(nets), ops.UnetWeights([nets[0]]), ops.UnetWeights([nets[1]]))
        return uw

    def xfyf_transform(self, image_combined):
        """(b, t, h, w, 2) -> (b, t, 1, h, w, 2); planes go to the bare U-Nets unnormalised and unpadded."""
        b, t, h, w, _ = image_combined.shape
        xf = self.dynamic_type == 'XF'
        both, wx, wy = self._xfyf_weights()
        if ag.grad_mode(self):
            return ag.xfyf(image_combined, xf, both, wx, wy, norm=False)
        pxf, pyf, _, _, mean = ops.xfyf_pack(image_combined, xf, norm=False)
        if pxf.shape == pyf.shape and pxf.data_ptr() + pxf.numel() * 4 == pyf.data_ptr():
            joint = torch.as_strided(pxf, (2 * pxf.shape[0],) + tuple(pxf.shape[1:]), pxf.stride())
            out = ops.unet2d_forward(joint, both)
            oxf, oyf = out[:pxf.shape[0]], out[pxf.shape[0]:]
        else:
            oxf, oyf = ops.unet2d_forward(pxf, wx), ops.unet2d_forward(pyf, wy)
        return ops.xfyf_unpack(oxf, oyf, None, None, mean, b, t, h, w, xf)

    def regularise(self, image_pred):
        b, t, c, h, w, ch = image_pred.shape
        if self.dynamic_type in ['XF', 'XT']:
            return self.xfyf_transform(image_pred.squeeze(2))
        if self.dynamic_type == '2D':
            if ag.grad_mode(self):
                return ag.norm_unet(image_pred.reshape(b * t, h, w, 2), self.model.hip_weights(), norm=False).view(b, t, 1, h, w, 2)
            planes, _ = ops.normunet_pack(image_pred.reshape(b * t, h, w, 2), norm=False)
            return ops.normunet_unpack(self.model(planes), None, h, w).view(b, t, 1, h, w, 2)
        if self.dynamic_type == '3D':
            # (b, t, 1, h, w, 2) -> (b, 2, t, h, w) volumes for the bare 3-D Unet and back (reference cinenet.py:251-253)
            if ag.grad_mode(self):
                vol = image_pred.reshape(b, t, h, w, 2).permute(0, 4, 1, 2, 3).contiguous()
                return ag.unet3d(vol, self.model.hip_weights()).permute(0, 2, 3, 4, 1).reshape(b, t, 1, h, w, 2)
            planes, _ = ops.normunet3d_pack(image_pred.reshape(b, t, h, w, 2), norm=False)
            return ops.normunet3d_unpack(self.model(planes), None, t, h, w).view(b, t, 1, h, w, 2)
        raise ValueError(f"unknown dynamic_type {self.dynamic_type!r}")

    def forward(self, image_pred, image_ref, mask, sens_maps, _tiled=None):
        if ag.grad_mode(self):
            model_out = self.regularise(image_pred)
            rhs = ag.AxpbyLamFn.apply(image_ref, model_out, self.lambda_reg)
            return ag.ConjGradFn.apply(model_out, rhs, self.lambda_reg, mask, sens_maps, self.CG_iters)
        model_out = self.regularise(image_pred)
        if ops.FUSED_CG and ops.CG_SOLVER and ops.is_row_mask(mask, sens_maps.expand(-1, image_pred.shape[1], -1, -1, -1, -1)):
            # one call = the whole DC block: rhs = x_ref + v x_reg formed in the solver's set-up kernel, 2 + 2 * CG_iters launches
            out = ops.conj_grad(model_out, image_ref, sens_maps, mask, self.lambda_reg, self.CG_iters, _tiled, rhs_is_ref=True)
            if out is not None:
                return out
        rhs = ops.axpby_dev(image_ref, model_out, lambda_reg=self.lambda_reg)       # x_ref + v x_reg
        return self.ConjGrad(model_out, rhs, mask, sens_maps, self.CG_iters, _tiled, _inplace=True)       # (model_out is this call's own tensor)


_ones = {}


def _one(like: torch.Tensor) -> torch.Tensor:
    t = _ones.get(like.device)
    if t is None:
        t = _ones[like.device] = torch.ones(1, device=like.device, dtype=torch.float32)
    return t


class CineNet(nn.Module):
    def __init__(self, num_cascades: int = 12, CG_iters: int = 4, chans: int = 18, pools: int = 4,
                 dynamic_type: str = 'XF', weight_sharing: bool = False):
        super().__init__()
        if dynamic_type in ['XF', 'XT']:
            self.model = Unet(chans, pools, dims=2) if weight_sharing else \
                nn.ModuleList([Unet(chans, pools, dims=2), Unet(chans, pools, dims=2)])
        elif dynamic_type == '3D':
            self.model = Unet(chans, pools, dims=3)
        else:
            self.model = Unet(chans, pools, dims=2)
        self.cascades = nn.ModuleList(
            [CineNetBlock(self.model, CG_iters, dynamic_type, weight_sharing) for _ in range(num_cascades)])

    def forward(self, masked_kspace: torch.Tensor, mask: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        mask = ops.as_mask_u8(mask, masked_kspace)          # any numeric 0 / 1 mask that broadcasts, like the reference
        if ag.grad_mode(self):               # k-space and maps are data: the graph starts at the first regulariser
            with torch.no_grad():
                image_pred = ops.sens_reduce(masked_kspace, sens_maps)
                image_ref = image_pred.clone()
            for cascade in self.cascades:
                image_pred = cascade(image_pred, image_ref, mask, sens_maps)
            return ag.AbsFn.apply(image_pred.squeeze(2))
        with torch.no_grad():
            return self._forward_infer(masked_kspace, mask, sens_maps)

    def _forward_infer(self, masked_kspace, mask, sens_maps):
        image_pred = ops.sens_reduce(masked_kspace, sens_maps)
        image_ref = image_pred.clone()
        tiled = ops.sens_tile_pack(sens_maps)          # once per forward: 42 applications of the normal operator read it (6 cascades x (1 + 6 CG iterations))
        for cascade in self.cascades:
            image_pred = cascade(image_pred, image_ref, mask, sens_maps, tiled)
        return ops.complex_abs(image_pred.squeeze(2))
