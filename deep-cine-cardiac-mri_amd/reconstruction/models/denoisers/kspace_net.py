"""KSpaceCNN on the gfx950 conv3d kernel (drop-in for the reference's denoisers/kspace_net.py:6-60).

Three Conv3d(3x3x3, 'same', bias) layers with ReLU in between, over (t, h, w) with the coils as batch.
Bias and ReLU ride in the MFMA kernel's epilogue; the layers hold the parameters under the reference's
``layers.{0,2,4}`` names.
"""
import torch
from torch import nn

from cine_hip import autograd as ag
from cine_hip import ops


class KSpaceCNN(nn.Module):
    def __init__(self, in_chans: int, out_chans: int, n_convs: int = 3, n_filters: int = 16):
        super().__init__()
        self.in_chans, self.out_chans, self.n_convs, self.n_filters = in_chans, out_chans, n_convs, n_filters
        convs = nn.ModuleList([nn.Conv3d(in_chans, n_filters, 3, padding='same'), nn.ReLU(inplace=True)])
        for _ in range(1, n_convs - 1):
            convs.append(nn.Conv3d(n_filters, n_filters, 3, padding='same'))
            convs.append(nn.ReLU(inplace=True))
        convs.append(nn.Conv3d(n_filters, out_chans, 3, padding='same'))
        self.layers = nn.Sequential(*convs)

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        """(b, t, coils, h, w, in_chans) -> (b, t, coils, h, w, out_chans)."""
        b, t, c, h, w, ch = inputs.shape
        x = inputs.permute(0, 2, 5, 1, 3, 4).reshape(b * c, ch, t, h, w).contiguous()
        convs = [m for m in self.layers if isinstance(m, nn.Conv3d)]
        train = ag.grad_mode(self) or (torch.is_grad_enabled() and inputs.requires_grad)
        for i, conv in enumerate(convs):
            if train:       # the layer as an autograd node with a HIP backward (cine_hip.autograd.Conv3dBiasReluFn)
                x = ag.Conv3dBiasReluFn.apply(x, conv.weight, conv.bias, i < len(convs) - 1)
            else:
                x = ops.conv3d_bias_relu(x, conv.weight, conv.bias, relu=i < len(convs) - 1)
        return x.reshape(b, c, self.out_chans, t, h, w).permute(0, 3, 1, 4, 5, 2)
