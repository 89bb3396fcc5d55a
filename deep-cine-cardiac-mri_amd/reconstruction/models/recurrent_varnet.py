"""VarNet_RNN on the MI355X kernels (drop-in for the reference's models/recurrent_varnet.py:13-150)."""
import math

import torch
from torch import nn

from cine_hip import autograd as ag
from cine_hip import ops
from .recurrent_common import BCRNNlayer, CRNNBody, CRNNcell  # noqa: F401  (re-exported like the reference)
from .varnet import SensitivityModel


class VarNet_RNN(CRNNBody):
    def __init__(self, num_cascades: int = 12, sens_chans: int = 8, sens_pools: int = 4, chans: int = 18):
        super().__init__()
        self.num_cascades, self.chans = num_cascades, chans
        self.sens_net = SensitivityModel(sens_chans, sens_pools)
        self._make_body(2, chans, 2)
        self.Softplus = nn.Softplus(1.)
        self.lambda_reg = nn.Parameter(torch.full((1,), math.log(math.e - 1.0)))

    def forward(self, ref_kspace: torch.Tensor, mask: torch.Tensor, acs=None) -> torch.Tensor:
        mask = ops.as_mask_u8(mask, ref_kspace)          # any numeric 0 / 1 mask; broadcast along batch / time like the reference
        if ag.grad_mode(self):
            return self._forward_train(ref_kspace, mask, acs)
        with torch.no_grad():
            return self._forward_infer(ref_kspace, mask, acs)

    def _forward_train(self, ref_kspace, mask, acs):
        """The chain of ``_forward_infer`` (reference recurrent_varnet.py:92-150) as an autograd graph: sensitivity network, BCRNN +
        conv pairs through the HIP backward kernels (hidden states flow across time AND cascades), image-space soft DC."""
        b, t, _, h, w, _ = ref_kspace.shape
        general = ops.is_general_mask(mask, ref_kspace)          # varies along w: the DC line of reference recurrent_varnet.py:80-90 term by term
        if b != 1 or not (general or ops.is_row_mask(mask, ref_kspace)):
            raise NotImplementedError("training through the HIP path: batch 1")
        sens_maps = self.sens_net(ref_kspace, mask, acs)
        img = ag.CoilReduceFn.apply(ref_kspace, sens_maps, None)                  # (1, t, 1, h, w, 2)
        zf = None if general else ag.CoilReduceFn.apply(ref_kspace, sens_maps, mask)
        state = self.zero_state(t, b, h, w, img)
        for _ in range(self.num_cascades):
            planes = img.view(t, h, w, 2).permute(0, 3, 1, 2).contiguous()        # (t, 2, h, w): frames are the conv batch
            out, state = self.body_train(planes.view(t, 1, 2, h, w), state, planes)
            new_img = out.permute(0, 2, 3, 1).reshape(1, t, 1, h, w, 2)
            if general:     # sens_expand -> soft DC on the coil-wise k-space -> the next cascade's sens_reduce, literally
                k = ops.soft_dc_blend(ag.SensExpandFn.apply(new_img, sens_maps, None), ref_kspace, mask, self.lambda_reg)
                img = ag.SensReduceFn.apply(k.contiguous(), sens_maps, None)
            else:
                img = ag.ImageDcFn.apply(new_img, sens_maps, zf, mask, self.lambda_reg)
        return ag.AbsFn.apply(img.squeeze(2))

    def _forward_infer(self, ref_kspace, mask, acs):
        sens_maps = self.sens_net(ref_kspace, mask, acs)
        b, t, _, h, w, _ = ref_kspace.shape
        if b != 1:
            raise NotImplementedError("the CRNN models assume batch 1, like the reference (recurrent_varnet.py:110-113)")
        hyb = ops.kspace_to_hybrid(ref_kspace)
        img = ops.hybrid_reduce(hyb, sens_maps)                                   # (1, t, 1, h, w, 2)
        rowmask = ops.is_row_mask(mask, ref_kspace)
        if rowmask:                                                               # image-space DC (see VarNet.forward)
            ops.kspace_to_hybrid(ref_kspace, out=hyb, mask=mask)
            zf = ops.hybrid_reduce(hyb, sens_maps)
        state = self.zero_state(t, b, h, w, img)
        tiled = ops.sens_tile_pack(sens_maps) if rowmask else None               # the maps as the DC kernel reads them fastest, once per forward
        for _ in range(self.num_cascades):
            planes, _ = ops.normunet_pack(img.view(t, h, w, 2), norm=False)      # (t, 2, h, w)
            out, state = self.body(planes.view(t, 1, 2, h, w), state, planes)
            new_img = ops.normunet_unpack(out, None, h, w).view(1, t, 1, h, w, 2)
            if rowmask:
                img = ops.image_dc(new_img, sens_maps, zf, mask, self.lambda_reg, sens_tiled=tiled)      # :80-90 + next reduce
            elif ops.is_general_mask(mask, ref_kspace):      # varies along w: the DC line term by term (ops.soft_dc_blend), then the reduce
                k = ops.soft_dc_blend(ops.sens_expand_dc(new_img, sens_maps), ref_kspace, mask, self.lambda_reg.detach())
                img = ops.sens_reduce(k, sens_maps, destroy_input=True)
            else:
                ops.expand_dc_hybrid(new_img, sens_maps, ref_kspace, mask, self.lambda_reg, out=hyb)   # :80-90
                img = ops.hybrid_reduce(hyb, sens_maps)
        return ops.complex_abs(img.squeeze(2))
