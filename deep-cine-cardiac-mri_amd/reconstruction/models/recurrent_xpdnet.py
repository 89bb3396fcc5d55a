"""XPDNet_RNN on the MI355X kernels (drop-in for the reference's models/recurrent_xpdnet.py:14-240)."""
import torch
from torch import nn

from cine_hip import autograd as ag
from cine_hip import ops
from .denoisers.kspace_net import KSpaceCNN
from .recurrent_common import BCRNNlayer, CRNNBody, CRNNcell  # noqa: F401
from .xpdnet import BackwardOperator, ForwardOperator, SensitivityModel


class XPDNet_RNN(CRNNBody):
    def __init__(self, num_cascades: int = 12, sens_chans: int = 8, sens_pools: int = 4, chans: int = 18,
                 primal_only: bool = True, n_primal: int = 5, n_dual: int = 1):
        super().__init__()
        self.num_cascades, self.chans = num_cascades, chans
        self.i_buffer_mode, self.k_buffer_mode = True, not primal_only
        self.i_buffer_size, self.k_buffer_size = n_primal, 1 if primal_only else n_dual
        self.backward_op = BackwardOperator(masked=False)
        self.forward_op = ForwardOperator(masked=True)
        self.sens_net = SensitivityModel(sens_chans, sens_pools)
        self._make_body(2 * (n_primal + 1), chans, 2 * n_primal)
        if not primal_only:
            self.kspace_net = nn.ModuleList([KSpaceCNN(in_chans=2 * (n_dual + 2), out_chans=2 * n_dual, n_convs=3,
                                                       n_filters=16) for _ in range(num_cascades)])
        else:
            self.kspace_net = [self.measurements_residual for _ in range(num_cascades)]

    @staticmethod
    def measurements_residual(concat_kspace: torch.Tensor) -> torch.Tensor:
        return concat_kspace[..., [0, 2]] - concat_kspace[..., [1, 3]]

    def forward(self, ref_kspace: torch.Tensor, mask: torch.Tensor, acs=None) -> torch.Tensor:
        mask = ops.as_mask_u8(mask, ref_kspace)          # any numeric 0 / 1 mask; broadcast along batch / time like the reference
        if ag.grad_mode(self):
            return self._forward_train(ref_kspace, mask, acs)
        with torch.no_grad():
            return self._forward_infer(ref_kspace, mask, acs)

    def _forward_train(self, ref_kspace, mask, acs):
        """The chain of ``_forward_infer`` as an autograd graph: sensitivity network, K step + masked backward operator (image space for the
        primal-only model; forward / k-space net / backward Functions with the dual buffer), CRNN body on the buffer planes."""
        n, nd = self.i_buffer_size, self.k_buffer_size
        b, t, _, h, w, _ = ref_kspace.shape
        general = ops.is_general_mask(mask, ref_kspace)          # varies along w: the literal k-space chain
        if b != 1 or not (general or ops.is_row_mask(mask, ref_kspace)):
            raise NotImplementedError("training through the HIP path: batch 1")
        mf = mask.to(ref_kspace.dtype) if general else None
        pick = lambda buf, k: torch.stack((buf[..., 0], buf[..., k]), dim=-1)
        sens_maps = self.sens_net(ref_kspace, mask, acs)
        image_buffer = ag.CoilReduceFn.apply(ref_kspace, sens_maps, None).repeat_interleave(n, dim=-1)     # (1, t, 1, h, w, 2n)
        if self.k_buffer_mode:
            kbuf = ref_kspace.repeat_interleave(nd, dim=-1)
        elif not general:
            zf = ag.CoilReduceFn.apply(ref_kspace, sens_maps, mask)
        state = self.zero_state(t, b, h, w, image_buffer)
        for i in range(self.num_cascades):
            x0 = pick(image_buffer, n)
            if self.k_buffer_mode:
                fwd = ag.SensExpandFn.apply(x0, sens_maps, None) * mf if general else ag.SensExpandFn.apply(x0, sens_maps, mask)
                cat_k = torch.cat([kbuf[..., :nd], fwd[..., :1], ref_kspace[..., :1], kbuf[..., nd:], fwd[..., 1:], ref_kspace[..., 1:]], dim=-1)
                kbuf = self.kspace_net[i](cat_k)
                if general:
                    bwd = ag.SensReduceFn.apply((pick(kbuf, nd) * mf).contiguous(), sens_maps, None)
                else:
                    bwd = ag.SensReduceFn.apply(pick(kbuf, nd).contiguous(), sens_maps, mask)
            elif general:
                bwd = ag.masked_residual_backward(x0, sens_maps, ref_kspace, mask)
            else:
                bwd = ag.ImageDcFixedFn.apply(x0, sens_maps, zf, mask, 1.0, 0.0, -1.0)
            cat = torch.cat([image_buffer[..., :n], bwd[..., :1], image_buffer[..., n:], bwd[..., 1:]], dim=-1)
            planes = cat.view(t, h, w, 2 * (n + 1)).permute(0, 3, 1, 2).contiguous()                       # (t, 2(n+1), h, w)
            out, state = self.body_train(planes.view(t, 1, 2 * (n + 1), h, w), state, torch.cat([planes[:, :n], planes[:, n + 1:2 * n + 1]], dim=1))
            image_buffer = out.permute(0, 2, 3, 1).reshape(1, t, 1, h, w, 2 * n)
        return ag.AbsFn.apply(pick(image_buffer, n).squeeze(2))

    def _forward_infer(self, ref_kspace, mask, acs):
        n = self.i_buffer_size
        b, t, _, h, w, _ = ref_kspace.shape
        if b != 1:
            raise NotImplementedError("the CRNN models assume batch 1, like the reference")
        sens_maps = self.sens_net(ref_kspace, mask, acs)
        image_buffer = ops.repeat_complex(ops.sens_reduce(ref_kspace, sens_maps), n)       # (1, t, 1, h, w, 2n)
        rowmask = ops.is_row_mask(mask, ref_kspace) and not self.k_buffer_mode
        general = ops.is_general_mask(mask, ref_kspace)          # varies along w (reference recurrent_xpdnet.py multiplies by any broadcastable mask)
        hyb = None if (rowmask or general) else torch.empty_like(ref_kspace)
        if rowmask:
            zf = ops.hybrid_reduce(ops.kspace_to_hybrid(ref_kspace, mask=mask), sens_maps)
        state = self.zero_state(t, b, h, w, image_buffer)
        keep = [i for i in range(2 * (n + 1)) if i not in (n, 2 * n + 1)]                  # channels [:n] and [n+1:-1]
        nd = self.k_buffer_size
        kbuf = ops.repeat_complex(ref_kspace, nd) if self.k_buffer_mode else None
        tiled = ops.sens_tile_pack(sens_maps) if rowmask else None
        for i in range(self.num_cascades):
            x0 = ops.extract_complex(image_buffer, 0, n)
            if self.k_buffer_mode:                                                          # dual buffer + KSpaceCNN
                fwd = ops.sens_expand_dc(x0, sens_maps) * mask + 0.0 if general else ops.sens_expand_dc(x0, sens_maps, None, mask, None, hard_mask=True)
                cat_k = torch.cat([kbuf[..., :nd], fwd[..., :1], ref_kspace[..., :1],
                                   kbuf[..., nd:], fwd[..., 1:], ref_kspace[..., 1:]], dim=-1)
                kbuf = self.kspace_net[i](cat_k).contiguous()
                bwd = ops.sens_reduce(ops.extract_complex(kbuf, 0, nd) * mask + 0.0, sens_maps)
            elif rowmask:
                bwd = ops.image_dc(x0, sens_maps, zf, mask, weights=(1.0, 0.0, -1.0), sens_tiled=tiled)      # A^H M (A x0 - k_ref) (:110-163)
            elif general:
                bwd = ops.masked_residual_backward(x0, sens_maps, ref_kspace, mask)
            else:
                ops.expand_resid_hybrid(x0, sens_maps, ref_kspace, mask, out=hyb)           # K step (:110-140)
                bwd = ops.hybrid_reduce(hyb, sens_maps)                                     # masked backward op (:142-163)
            cat = torch.cat([image_buffer[..., :n], bwd[..., :1], image_buffer[..., n:], bwd[..., 1:]], dim=-1)
            planes = ops.chanlast_to_planes(cat.view(t, h, w, 2 * (n + 1)))                 # (t, 2(n+1), h, w)
            out, state = self.body(planes.view(t, 1, 2 * (n + 1), h, w), state, planes[:, keep].contiguous())
            image_buffer = ops.planes_to_chanlast(out, h, w).view(1, t, 1, h, w, 2 * n)
        return ops.complex_abs(ops.extract_complex(image_buffer, 0, n).squeeze(2))
