"""Common layers of the score_sde model zoo used by NCSN++ (mirror of the reference's ``models/layers.py``:
get_act :29-41, variance_scaling/default_init :54-91, ddpm_conv1x1/3x3 :100-125, get_timestep_embedding :516-530,
NIN :547-556) on the libipdm.so kernels."""
import math

import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..ncsn.models.layers import _Act


def get_act(config):
    name = config.model.nonlinearity.lower()
    if name not in ("elu", "relu", "lrelu", "swish"):
        raise NotImplementedError('activation function does not exist!')
    return _Act(name)


def variance_scaling(scale, mode, distribution, in_axis=1, out_axis=0, dtype=torch.float32, device='cpu'):
    def init(shape, dtype=dtype, device=device):
        receptive = np.prod(shape) / shape[in_axis] / shape[out_axis]
        fan_in, fan_out = shape[in_axis] * receptive, shape[out_axis] * receptive
        denom = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": (fan_in + fan_out) / 2}[mode]
        variance = scale / denom
        if distribution == "normal":
            return torch.randn(*shape, dtype=dtype, device=device) * np.sqrt(variance)
        if distribution == "uniform":
            return (torch.rand(*shape, dtype=dtype, device=device) * 2. - 1.) * np.sqrt(3 * variance)
        raise ValueError("invalid distribution for variance scaling initializer")
    return init


def default_init(scale=1.):
    scale = 1e-10 if scale == 0 else scale
    return variance_scaling(scale, 'fan_avg', 'uniform')


# GroupNorm statistics from the producing convolution's epilogue partials (IPDM_GN_PARTIALS=0: always from a pass over the tensor)
GN_FROM_PARTIALS = os.environ.get("IPDM_GN_PARTIALS", "1") != "0"


class Conv(ops.PackedWeightMixin, nn.Module):
    """stride-1 'same' convolution with nn.Conv2d's parameter names (weight [Cout,Cin,k,k], bias)"""

    def __init__(self, in_planes, out_planes, kernel_size, bias=True, init_scale=1., dilation=1):
        super().__init__()
        self.in_planes, self.out_planes, self.kernel_size, self.dilation = in_planes, out_planes, kernel_size, dilation
        self.weight = nn.Parameter(default_init(init_scale)((out_planes, in_planes, kernel_size, kernel_size)))
        self.bias = nn.Parameter(torch.zeros(out_planes)) if bias else None
        self._cache = ops.PackedWeightCache()

    # score_sde networks feed raw (un-normalised) skip / progressive streams into convolutions: ops.impl_unbounded()
    def packed(self):
        impl = ops.impl_unbounded()
        return self._cache.get(self.weight, "direct_" + impl, lambda w: ops.conv_weight(w, impl))

    def packed_wino(self):
        impl = ops.impl_unbounded()
        return self._cache.get(self.weight, "wino_" + impl, lambda w: ops.conv_wino_split_weight(w, impl))

    def packed_wino1d(self):
        return self._cache.get(self.weight, "wino1d", ops.conv_wino1d_weight)

    def forward(self, x, residual=None, bounded=False, bias_rows=None, out_scale=1.0, in_amax=None, feeds_conv=False,
                want_stats=False):
        """bounded: x is act(GroupNorm(.)) (possibly FIR-resampled) -- |x| <= |gamma| sqrt(group size) + |beta|, inside the
        f16x2 family's static range; otherwise (raw skip / pyramid / input streams) the convolution runs with the dynamic
        range: the per-image maxima ride on the tensor from its producer (ops.in_amax_for: a convolution epilogue, a FIR
        resampler passing its input's bound on) or are measured once and attached.
        want_stats: a GroupNorm reads this layer's result next: where the convolution kernel has a statistics epilogue its partials
        ride on the result (ops.stats_partials_of) and that GroupNorm does not read the tensor for its statistics.
        feeds_conv: a convolution reads this layer's RESULT raw (a block's output: the next shortcut / the skip connection):
        the epilogue accumulates its maxima.
        bias_rows [B, Cout]: replaces the bias by one row per image (the caller has added self.bias into it);
        out_scale: result = (conv + bias + residual) * out_scale -- both folded into the epilogue (split families only)."""
        impl = ops.impl_unbounded()
        amax = None
        if not bounded and impl == "hx2":
            amax = in_amax if in_amax is not None else ops.in_amax_for(x, impl, always=True)
        produce = feeds_conv and impl == "hx2"
        bias = None if self.bias is None else self.bias.data
        if bias_rows is not None:
            bias = bias_rows
        if (self.kernel_size == 3 and self.dilation == 1 and residual is None and min(self.in_planes, self.out_planes) <= 3
                and ops.conv3x3_thin_ok(self.in_planes, self.out_planes, x.shape[2], x.shape[3])):
            return ops.conv3x3_thin(x, self.weight.data, bias)         # first / last layer: streaming kernels
        if (ops.impl_unbounded() in ops.SPLIT_IMPLS and self.kernel_size == 3
                and ops.wino_bx3_pays(self.in_planes, self.out_planes, x.shape[2], x.shape[3], self.dilation)):
            one_d = (impl == "hx2" and x.data_ptr() % 16 == 0
                     and ops.wino1d_pays(self.in_planes, self.out_planes, x.shape[2], x.shape[3], self.dilation))
            return ops.conv2d_wino_bx3(x, self.packed_wino1d() if one_d else self.packed_wino(), bias, residual,
                                       dilation=self.dilation, in_amax=amax, out_scale=out_scale, want_amax=produce,
                                       want_stats=want_stats and GN_FROM_PARTIALS)
        return ops.conv2d(x, self.packed(), bias, residual=residual, dilation=self.dilation, in_amax=amax, out_scale=out_scale,
                          want_amax=produce)

    def epilogue_folds(self):
        """True when per-image bias rows / out_scale can ride in this convolution's epilogue (split-operand kernels)"""
        return ops.impl_unbounded() in ops.SPLIT_IMPLS and not (self.kernel_size == 3 and min(self.in_planes, self.out_planes) <= 3)

    def forward_parts(self, xs, residual=None, out_scale=1.0, in_amax=None):
        """the convolution of torch.cat(xs, dim=1) WITHOUT the concatenation: sum_i conv(xs[i], weight[:, slice_i]), each part
        taking the previous sum as its residual (the bias rides with the first, out_scale with the last).  1x1 kernels.
        Every part runs with ITS tensor's maxima (in_amax: one bound for all parts instead)."""
        assert self.kernel_size == 1 and sum(t.shape[1] for t in xs) == self.in_planes
        impl = ops.impl_unbounded()
        acc, c0 = residual, 0
        for i, t in enumerate(xs):
            c1 = c0 + t.shape[1]
            packed = self._cache.get(self.weight, f"direct_{impl}_{c0}_{c1}",
                                     lambda w, a=c0, b=c1: ops.conv_weight(w[:, a:b].contiguous(), impl))
            last = i == len(xs) - 1
            am = None
            if impl == "hx2":
                am = in_amax if in_amax is not None else ops.in_amax_for(t, impl, always=True)
            acc = ops.conv2d(t, packed, self.bias.data if (i == 0 and self.bias is not None) else None, residual=acc,
                             in_amax=am, out_scale=out_scale if last else 1.0)
            c0 = c1
        return acc


def ddpm_conv1x1(in_planes, out_planes, stride=1, bias=True, init_scale=1., padding=0):
    assert stride == 1 and padding == 0
    return Conv(in_planes, out_planes, 1, bias=bias, init_scale=init_scale)


def ddpm_conv3x3(in_planes, out_planes, stride=1, bias=True, dilation=1, init_scale=1., padding=1):
    assert stride == 1 and padding == dilation
    return Conv(in_planes, out_planes, 3, bias=bias, init_scale=init_scale, dilation=dilation)


def get_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
    assert len(timesteps.shape) == 1
    half_dim = embedding_dim // 2
    emb = math.log(max_positions) / (half_dim - 1)
    emb = torch.exp(torch.arange(half_dim, dtype=torch.float32, device=timesteps.device) * -emb)
    emb = timesteps.float()[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1), mode='constant')
    return emb


class NIN(nn.Module):
    """network-in-network = 1x1 convolution with the weight stored (in_dim, num_units): exactly the packed [1][Cin][Cout] layout
    of the fp32-MFMA convolution (no repacking there); the split-operand families pack W^T once per parameter version"""

    def __init__(self, in_dim, num_units, init_scale=0.1):
        super().__init__()
        self.W = nn.Parameter(default_init(scale=init_scale)((in_dim, num_units)), requires_grad=True)
        self.b = nn.Parameter(torch.zeros(num_units), requires_grad=True)
        self._cache = ops.PackedWeightCache()

    def invalidate(self):
        self._cache.clear()

    def _load_from_state_dict(self, *args, **kwargs):
        self._cache.clear()
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x, residual=None, bounded=False):
        """bounded: x is a GroupNorm output (no activation: |x| <= |gamma| sqrt(group size) + |beta|)"""
        impl = ops.impl_unbounded()
        if impl in ops.SPLIT_IMPLS and self.W.shape[0] % 16 == 0 and self.W.shape[1] % 32 == 0:
            packed = self._cache.get(self.W, "nin_" + impl,
                                     lambda w: ops.conv_weight(w.t().reshape(w.shape[1], w.shape[0], 1, 1).contiguous(), impl))
            am = None if bounded or impl != "hx2" else ops.in_amax_for(x, impl, always=True)
            return ops.conv2d(x, packed, self.b.data, residual=residual, in_amax=am)
        return ops.conv2d(x, self.W.data.view(1, self.W.shape[0], self.W.shape[1]), self.b.data, residual=residual)


class GroupNorm(nn.Module):
    """torch.nn.GroupNorm parameters (weight, bias), evaluated as per-(b,c) coefficients + one fused
    affine(+activation) pass"""

    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def forward(self, x, act=ops.ACT_NONE, want_amax=False):
        """want_amax: -> (y, per-image max |x| [B]): what a convolution of the RAW x needs for its dynamic range, from the
        statistics pass that reads x anyway"""
        if isinstance(x, (tuple, list)):        # GroupNorm(+act) of torch.cat(x, dim=1) without the concatenation
            return ops.groupnorm_act_cat(x[0], x[1], self.weight.data, self.bias.data, self.num_groups, self.eps, act,
                                         want_amax=want_amax)
        if want_amax:
            coef, am = ops.groupnorm_coef(x, self.weight.data, self.bias.data, self.num_groups, self.eps, want_amax=True)
            return ops.affine_act(x, coef, act), am
        coef = ops.groupnorm_coef(x, self.weight.data, self.bias.data, self.num_groups, self.eps)
        return ops.affine_act(x, coef, act)


class Linear(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(default_init()((out_features, in_features)))
        self.bias = nn.Parameter(torch.zeros(out_features))

    def forward(self, x, act_in=ops.ACT_NONE):
        return ops.linear(x, self.weight.data, self.bias.data, act_in)
