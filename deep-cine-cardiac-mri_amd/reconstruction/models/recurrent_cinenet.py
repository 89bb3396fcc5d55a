"""CineNet_RNN on the MI355X kernels (drop-in for the reference's models/recurrent_cinenet.py:12-187)."""
import math

import torch
from torch import nn

from cine_hip import autograd as ag
from cine_hip import ops
from .cinenet import CineNetBlock
from .recurrent_common import BCRNNlayer, CRNNBody, CRNNcell  # noqa: F401


class CineNet_RNN(CRNNBody):
    def __init__(self, num_cascades: int = 10, CG_iters: int = 4, chans: int = 64):
        super().__init__()
        self.num_cascades, self.CG_iters, self.chans = num_cascades, CG_iters, chans
        self._make_body(2, chans, 2)
        self.Softplus = nn.Softplus(1.)
        self.lambda_reg = nn.Parameter(torch.full((1,), math.log(math.e - 1.0)))

    # the normal operator and CG are CineNet's (reference recurrent_cinenet.py:75-125 == cinenet.py:121-171)
    HOperator = CineNetBlock.HOperator
    ConjGrad = CineNetBlock.ConjGrad

    def forward(self, ref_kspace: torch.Tensor, mask: torch.Tensor, sens_maps: torch.Tensor) -> torch.Tensor:
        mask = ops.as_mask_u8(mask, ref_kspace)          # any numeric 0 / 1 mask; broadcast along batch / time like the reference
        if ag.grad_mode(self):
            return self._forward_train(ref_kspace, mask, sens_maps)
        with torch.no_grad():
            return self._forward_infer(ref_kspace, mask, sens_maps)

    def _forward_train(self, ref_kspace, mask, sens_maps):
        """``_forward_infer`` as an autograd graph (k-space and maps are data): CRNN body through the HIP backward kernels, the
        conjugate-gradient solve through its adjoint recurrence (detached step sizes, recurrent_cinenet.py:113-123), lambda through both."""
        b, t, _, h, w, _ = ref_kspace.shape
        if b != 1 or not (ops.is_row_mask(mask, ref_kspace) or ops.is_general_mask(mask, ref_kspace)):      # (ConjGradFn's operator serves both layouts)
            raise NotImplementedError("training through the HIP path: batch 1")
        with torch.no_grad():
            x_ref = ops.sens_reduce(ref_kspace, sens_maps)
        img = x_ref
        state = self.zero_state(t, b, h, w, img)
        for _ in range(self.num_cascades):
            planes = img.view(t, h, w, 2).permute(0, 3, 1, 2).contiguous()
            out, state = self.body_train(planes.view(t, 1, 2, h, w), state, planes)
            x = out.permute(0, 2, 3, 1).reshape(1, t, 1, h, w, 2)
            rhs = ag.AxpbyLamFn.apply(x_ref, x, self.lambda_reg)
            img = ag.ConjGradFn.apply(x, rhs, self.lambda_reg, mask, sens_maps, self.CG_iters)
        return ag.AbsFn.apply(img.squeeze(2))

    def _forward_infer(self, ref_kspace, mask, sens_maps):
        b, t, _, h, w, _ = ref_kspace.shape
        if b != 1:
            raise NotImplementedError("the CRNN models assume batch 1, like the reference")
        x_ref = ops.sens_reduce(ref_kspace, sens_maps)
        img = x_ref
        state = self.zero_state(t, b, h, w, img)
        tiled = ops.sens_tile_pack(sens_maps)             # the maps as the normal operator reads them fastest, once per forward
        for _ in range(self.num_cascades):
            planes, _ = ops.normunet_pack(img.view(t, h, w, 2), norm=False)
            out, state = self.body(planes.view(t, 1, 2, h, w), state, planes)
            x = ops.normunet_unpack(out, None, h, w).view(1, t, 1, h, w, 2)
            rhs = ops.axpby_dev(x_ref, x, lambda_reg=self.lambda_reg)
            img = self.ConjGrad(x, rhs, mask, sens_maps, self.CG_iters, tiled)
        return ops.complex_abs(img.squeeze(2))
