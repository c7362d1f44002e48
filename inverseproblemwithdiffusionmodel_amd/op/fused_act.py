"""``fused_leaky_relu`` / ``FusedLeakyReLU`` (mirror of the reference's ``op/fused_act.py:77-100``) on the
gfx950 bias+activation kernel.  Forward only."""
import torch
from torch import nn

from .. import ops


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    if not input.is_cuda:
        raise RuntimeError("fused_leaky_relu: expected a GPU tensor (this build has no CPU fallback)")
    return ops.fused_bias_act_raw(input, bias, None, 3, 0, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
