"""InstanceNorm++ (mirror of the reference's ``ncsn/models/normalization.py:150-176``), computed as
per-(image, channel) coefficients (mu, scale, shift) that the consuming convolution applies while it
stages its input -- the normalised tensor is never materialised on the hot path."""
import torch
import torch.nn as nn

from ... import ops


class InstanceNorm2dPlus(nn.Module):
    def __init__(self, num_features, bias=True):
        super().__init__()
        self.num_features = num_features
        self.bias = bias
        self.alpha = nn.Parameter(torch.zeros(num_features))
        self.gamma = nn.Parameter(torch.zeros(num_features))
        self.alpha.data.normal_(1, 0.02)
        self.gamma.data.normal_(1, 0.02)
        if bias:
            self.beta = nn.Parameter(torch.zeros(num_features))

    def coef(self, x):
        """(B, C, 3) float32: out = (x - coef[...,0]) * coef[...,1] + coef[...,2]"""
        return ops.instnorm_plus_coef(x, self.alpha.data, self.gamma.data, self.beta.data if self.bias else None)

    def forward(self, x, act=ops.ACT_NONE):
        return ops.affine_act(x, self.coef(x), act)


def get_normalization(config, conditional=True):
    norm = config.model.normalization
    if conditional:
        raise NotImplementedError("conditional normalisation belongs to NCSNv1, which is off the hot path")
    if norm == "InstanceNorm++":
        return InstanceNorm2dPlus
    raise NotImplementedError(f"{norm}: only InstanceNorm++ (every shipped config) has a gfx950 kernel")
