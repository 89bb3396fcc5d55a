"""U-Net regulariser whose forward pass runs on the gfx950 conv kernels.

Same constructor, attribute tree and state-dict keys as the reference's
denoisers/unet.py (Unet :6, ConvBlock :128, TransposeConvBlock :185); the
``torch.nn`` layers inside the blocks only HOLD the parameters.  ``forward``
hands the raw weight pointers to ``cine_unet2d_forward`` (MFMA implicit-GEMM
3x3 convs with InstanceNorm/LeakyReLU/pool/concat fused into the operand
staging).  With gradients enabled the 2-D U-Net runs as ``cine_hip.autograd.UnetFn``
(forward keeps the raw layer outputs, backward = cine_unet2d_backward).
"""
import torch
from torch import nn

from cine_hip import autograd as ag
from cine_hip import ops


def _holders(dims: int):
    assert dims in [2, 3], "Dimensions must be either 2 or 3"
    if dims == 2:
        return nn.Conv2d, nn.ConvTranspose2d, nn.InstanceNorm2d, nn.Dropout2d
    return nn.Conv3d, nn.ConvTranspose3d, nn.InstanceNorm3d, nn.Dropout3d


class ConvBlock(nn.Module):
    """Parameter holder for conv3-IN-LReLU(0.2)-Dropout twice (layers.0 / layers.4 carry weights)."""

    def __init__(self, in_chans: int, out_chans: int, drop_prob: float, dims: int):
        super().__init__()
        self.in_chans, self.out_chans, self.drop_prob, self.dims = in_chans, out_chans, drop_prob, dims
        conv, _, norm, drop = _holders(dims)
        half = lambda ci: [conv(ci, out_chans, kernel_size=3, padding=1, bias=False), norm(out_chans),
                           nn.LeakyReLU(negative_slope=0.2, inplace=True), drop(drop_prob)]
        self.layers = nn.Sequential(*half(in_chans), *half(out_chans))

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        """The block on its own (inference; the U-Nets run whole launch sequences instead): (N, in_chans, H, W) -> (N, out_chans, H, W), or volumes
        (N, in_chans, T, H, W) with dims = 3 (reference unet.py:170-182).  In training mode with drop_prob > 0 the Dropout2d / Dropout3d behind each
        LeakyReLU (unet.py:163,167) scales whole (sample, channel) planes by 0 or 1 / (1 - p), drawn from torch's generator."""
        if torch.is_grad_enabled() and (image.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("a stand-alone ConvBlock has no backward pass on the HIP path (the U-Nets train as whole sequences); wrap the call in torch.no_grad()")
        x = image
        for conv in (self.layers[0], self.layers[4]):
            if self.dims == 3:
                y, st = ops.conv3d_in(x, conv.weight)
            else:
                y, st = ops.conv3x3_in([(x, None, 0)], ops.pack_conv3x3(conv.weight), conv.out_channels, x.shape[2], x.shape[3])
            x = ops.instnorm_lrelu_apply(y, st)
            if self.training and self.drop_prob > 0:
                keep = (torch.rand(x.shape[:2], device=x.device) >= self.drop_prob).to(x.dtype) / (1.0 - self.drop_prob)
                x = x * keep.view(*x.shape[:2], *([1] * (x.dim() - 2)))
        return x


class TransposeConvBlock(nn.Module):
    """Parameter holder for tconv(k2,s2)-IN-LReLU(0.2) (layers.0 carries the weight)."""

    def __init__(self, in_chans: int, out_chans: int, dims: int):
        super().__init__()
        self.in_chans, self.out_chans, self.dims = in_chans, out_chans, dims
        _, tconv, norm, _ = _holders(dims)
        self.layers = nn.Sequential(tconv(in_chans, out_chans, kernel_size=2, stride=2, bias=False),
                                    norm(out_chans), nn.LeakyReLU(negative_slope=0.2, inplace=True))

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        """The block on its own (inference): (N, in_chans, H, W) -> (N, out_chans, 2H, 2W), or (N, in_chans, T, H, W) -> (N, out_chans, 2T, 2H, 2W)
        with dims = 3 (reference unet.py:221-233)."""
        if torch.is_grad_enabled() and (image.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("a stand-alone TransposeConvBlock has no backward pass on the HIP path; wrap the call in torch.no_grad()")
        wt = self.layers[0].weight
        if self.dims == 3:
            y, st = ops.tconv3d_in(image, wt)
        else:
            y, st = ops.tconv2x2_in(image, None, 0, ops.pack_tconv2x2(wt), wt.shape[1])
        return ops.instnorm_lrelu_apply(y, st)


class Unet(nn.Module):
    def __init__(self, chans: int = 32, num_pool_layers: int = 4, in_chans: int = 2, out_chans: int = 2,
                 drop_prob: float = 0.0, dims: int = 2):
        super().__init__()
        self.chans, self.num_pool_layers = chans, num_pool_layers
        self.in_chans, self.out_chans = in_chans, out_chans
        self.drop_prob, self.dims = drop_prob, dims
        conv = _holders(dims)[0]

        widths = [chans << d for d in range(num_pool_layers)]
        self.down_sample_layers = nn.ModuleList(
            ConvBlock(ci, co, drop_prob, dims) for ci, co in zip([in_chans] + widths[:-1], widths))
        self.conv = ConvBlock(widths[-1], widths[-1] * 2, drop_prob, dims)
        self.up_conv = nn.ModuleList()
        self.up_transpose_conv = nn.ModuleList()
        for i, ch in enumerate(reversed(widths)):
            self.up_transpose_conv.append(TransposeConvBlock(ch * 2, ch, dims))
            block = ConvBlock(ch * 2, ch, drop_prob, dims)
            if i == num_pool_layers - 1:
                block = nn.Sequential(block, conv(ch, out_chans, kernel_size=1, stride=1))
            self.up_conv.append(block)
        self._hip_weights = None

    def hip_weights(self) -> "ops.UnetWeights":
        if self._hip_weights is None:
            self._hip_weights = ops.UnetWeights([self])
        return self._hip_weights

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        drops = self.training and self.drop_prob > 0          # nn.Dropout2d / 3d are active in training mode, with or without autograd (unet.py:163,167)
        if self.dims == 3:
            if drops or ag.grad_mode(self) or (torch.is_grad_enabled() and image.requires_grad):
                return ag.unet3d(image, self.hip_weights())
            return ops.unet3d_forward(image, self.hip_weights())
        if drops or ag.grad_mode(self) or (torch.is_grad_enabled() and image.requires_grad):
            return ag.unet2d(image, self.hip_weights())          # the training sequence: Dropout2d multipliers folded into the statistics records
        return ops.unet2d_forward(image, self.hip_weights())
