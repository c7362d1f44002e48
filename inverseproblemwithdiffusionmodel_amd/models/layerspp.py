"""NCSN++ layers (mirror of the reference's ``models/layerspp.py``: GaussianFourierProjection :32-41, Combine
:44-59, AttnBlockpp :62-91, Upsample :94-127, Downsample :130-163, ResnetBlockBigGANpp :212-274)."""
import numpy as np
import torch
import torch.nn as nn

from . import layers, up_or_down_sampling
from .. import ops

conv1x1 = layers.ddpm_conv1x1
conv3x3 = layers.ddpm_conv3x3
NIN = layers.NIN
default_init = layers.default_init
GroupNorm = layers.GroupNorm
INV_SQRT2 = float(np.float32(1.0) / np.sqrt(np.float32(2.0)))


def _groups(ch):
    return min(ch // 4, 32)


class GaussianFourierProjection(nn.Module):
    """Gaussian Fourier embeddings for noise levels ((B,) -> (B, 2*embedding_size); host-sized math)"""

    def __init__(self, embedding_size=256, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)

    def forward(self, x):
        x_proj = x[:, None] * self.W[None, :] * 2 * np.pi
        return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


class Combine(nn.Module):
    """Combine information from skip connections."""

    def __init__(self, dim1, dim2, method='cat'):
        super().__init__()
        self.Conv_0 = conv1x1(dim1, dim2)
        self.method = method

    def forward(self, x, y):
        if self.method == 'sum':
            return self.Conv_0(x, residual=y)
        if self.method == 'cat':
            return torch.cat([self.Conv_0(x), y], dim=1)
        raise ValueError(f'Method {self.method} not recognized.')


def _skip(x, h, rescale):
    return ops.axpby(x, h, INV_SQRT2, INV_SQRT2) if rescale else ops.add(x, h)


class _TembBias:
    """`h = Conv_0(.) ; h += Dense_0(act(temb))[:, :, None, None]` (reference models/layerspp.py:190-192, 252-254) as ONE bias row
    per image in the convolution's epilogue: rows = Dense_0(act(temb)) with the convolution's own bias added into the
    Dense bias.  The [Cout] sum is cached under the PARAMETERS' own version counters (a `.data` view has a counter of its own that
    never moves: `p.copy_()` -- load_state_dict, an optimiser step -- bumps `p._version` only); a write through `.data` (the
    reference's EMA swap) bumps nothing, so the owning block also clears the cache in its load_state_dict hook and invalidate()"""

    EPOCH = 0                            # bumped by every clear(): part of the model-level bank's tag (models/ncsnpp.py)

    def __init__(self):
        self._tag, self._sum = None, None
        self.preset = None               # this block's rows out of the model's ONE linear launch for all blocks, when it ran

    def clear(self):
        self._tag, self._sum = None, None
        _TembBias.EPOCH += 1

    def rows(self, dense, conv, temb, code):
        if self.preset is not None:      # (single use: set by NCSNpp.forward for this evaluation)
            rows, self.preset = self.preset, None
            return rows
        b_d, b_c = dense.bias, conv.bias
        tag = (b_d._version, b_d.data_ptr(), b_c._version, b_c.data_ptr())
        if tag != self._tag:
            self._sum, self._tag = (b_d.data + b_c.data).contiguous(), tag
        return ops.linear(temb, dense.weight.data, self._sum, code)


class _TembBiasOwner:
    """for blocks holding `self._temb_bias`: the cached bias sum is dropped whenever the state dict is loaded, and on invalidate()"""

    def invalidate(self):
        self._temb_bias.clear()

    def _load_from_state_dict(self, *args, **kwargs):
        self._temb_bias.clear()
        return super()._load_from_state_dict(*args, **kwargs)


class AttnBlockpp(nn.Module):
    """Channel-wise self-attention block. Modified from DDPM."""

    def __init__(self, channels, skip_rescale=False, init_scale=0.):
        super().__init__()
        self.GroupNorm_0 = GroupNorm(num_groups=_groups(channels), num_channels=channels, eps=1e-6)
        self.NIN_0 = NIN(channels, channels)
        self.NIN_1 = NIN(channels, channels)
        self.NIN_2 = NIN(channels, channels)
        self.NIN_3 = NIN(channels, channels, init_scale=init_scale)
        self.skip_rescale = skip_rescale

    def forward(self, x):
        C = x.shape[1]
        h = self.GroupNorm_0(x)
        q, k, v = self.NIN_0(h, bounded=True), self.NIN_1(h, bounded=True), self.NIN_2(h, bounded=True)
        h = ops.attention(q, k, v, int(C) ** (-0.5))
        if not self.skip_rescale:
            return self.NIN_3(h, residual=x)
        return _skip(x, self.NIN_3(h), True)


class Upsample(nn.Module):
    def __init__(self, in_ch=None, out_ch=None, with_conv=False, fir=False, fir_kernel=(1, 3, 3, 1)):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        if not fir and with_conv:
            self.Conv_0 = conv3x3(in_ch, out_ch)
        elif fir and with_conv:
            self.Conv2d_0 = up_or_down_sampling.Conv2d(in_ch, out_ch, kernel=3, up=True, resample_kernel=fir_kernel,
                                                       use_bias=True, kernel_init=default_init())
        self.fir, self.with_conv, self.fir_kernel, self.out_ch = fir, with_conv, fir_kernel, out_ch

    def forward(self, x):
        if not self.fir:
            h = up_or_down_sampling.naive_upsample_2d(x, 2)              # F.interpolate(..., 'nearest')
            return self.Conv_0(h) if self.with_conv else h
        if not self.with_conv:
            return up_or_down_sampling.upsample_2d(x, self.fir_kernel, factor=2)
        return self.Conv2d_0(x)


class Downsample(nn.Module):
    def __init__(self, in_ch=None, out_ch=None, with_conv=False, fir=False, fir_kernel=(1, 3, 3, 1)):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        if not fir and with_conv:
            self.Conv_0 = conv3x3(in_ch, out_ch)                 # applied with stride 2, padding 0 (see forward)
        if fir and with_conv:
            self.Conv2d_0 = up_or_down_sampling.Conv2d(in_ch, out_ch, kernel=3, down=True, resample_kernel=fir_kernel,
                                                       use_bias=True, kernel_init=default_init())
        self.fir, self.fir_kernel, self.with_conv, self.out_ch = fir, fir_kernel, with_conv, out_ch

    def forward(self, x):
        if not self.fir:
            if self.with_conv:
                # F.pad(x, (0, 1, 0, 1)) then conv3x3(stride 2, padding 0): the stride-1 kernel sampled at the odd positions
                x = torch.nn.functional.pad(x, (0, 1, 0, 1))
                c = self.Conv_0
                return ops.conv2d_stride2_valid(x.contiguous(), c.packed(), None if c.bias is None else c.bias.data, 3)
            return up_or_down_sampling.naive_downsample_2d(x, 2)          # F.avg_pool2d(x, 2, 2)
        if not self.with_conv:
            return up_or_down_sampling.downsample_2d(x, self.fir_kernel, factor=2)
        return self.Conv2d_0(x)


class ResnetBlockDDPMpp(_TembBiasOwner, nn.Module):
    """ResBlock adapted from DDPM (mirror of the reference's ``models/layerspp.py:166-209``): GroupNorm+act -> conv3x3
    -> + Dense(act(temb)) -> GroupNorm+act -> (dropout: identity when sampling) -> conv3x3, shortcut through NIN (or a
    3x3 convolution) when the channel count changes; same parameter names."""

    def __init__(self, act, in_ch, out_ch=None, temb_dim=None, conv_shortcut=False, dropout=0.1, skip_rescale=False,
                 init_scale=0.):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        self.GroupNorm_0 = GroupNorm(num_groups=_groups(in_ch), num_channels=in_ch, eps=1e-6)
        self.Conv_0 = conv3x3(in_ch, out_ch)
        if temb_dim is not None:
            self.Dense_0 = layers.Linear(temb_dim, out_ch)
        self.GroupNorm_1 = GroupNorm(num_groups=_groups(out_ch), num_channels=out_ch, eps=1e-6)
        self.Conv_1 = conv3x3(out_ch, out_ch, init_scale=init_scale)
        if in_ch != out_ch:
            if conv_shortcut:
                self.Conv_2 = conv3x3(in_ch, out_ch)
            else:
                self.NIN_0 = NIN(in_ch, out_ch)
        self.skip_rescale, self.act, self.out_ch, self.conv_shortcut = skip_rescale, act, out_ch, conv_shortcut
        self._temb_bias = _TembBias()

    def forward(self, x, temb=None):
        code = self.act.code
        fold = self.Conv_0.epilogue_folds()
        if temb is not None and fold:
            h = self.Conv_0(self.GroupNorm_0(x, code), bounded=True, want_stats=True,     # GroupNorm_1 reads the result next
                            bias_rows=self._temb_bias.rows(self.Dense_0, self.Conv_0, temb, code))
        else:
            h = self.Conv_0(self.GroupNorm_0(x, code), bounded=True)
            if temb is not None:
                t = self.Dense_0(temb, act_in=code)
                shift = torch.stack([torch.zeros_like(t), torch.ones_like(t), t], dim=-1).contiguous()
                h = ops.affine_act(h, shift, ops.ACT_NONE, out=h)
        if x.shape[1] != self.out_ch:
            x = self.Conv_2(x) if self.conv_shortcut else self.NIN_0(x)
        if fold:                                                   # (x + Conv_1(.)) [/ sqrt 2] in Conv_1's epilogue
            return self.Conv_1(self.GroupNorm_1(h, code), residual=x, bounded=True,
                               out_scale=INV_SQRT2 if self.skip_rescale else 1.0, feeds_conv=True, want_stats=True)
        h = self.Conv_1(self.GroupNorm_1(h, code), bounded=True)
        return _skip(x, h, self.skip_rescale)


class ResnetBlockBigGANpp(_TembBiasOwner, nn.Module):
    def __init__(self, act, in_ch, out_ch=None, temb_dim=None, up=False, down=False, dropout=0.1, fir=False,
                 fir_kernel=(1, 3, 3, 1), skip_rescale=True, init_scale=0.):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        self.GroupNorm_0 = GroupNorm(num_groups=_groups(in_ch), num_channels=in_ch, eps=1e-6)
        self.up, self.down, self.fir, self.fir_kernel = up, down, fir, fir_kernel
        self.Conv_0 = conv3x3(in_ch, out_ch)
        if temb_dim is not None:
            self.Dense_0 = layers.Linear(temb_dim, out_ch)
        self.GroupNorm_1 = GroupNorm(num_groups=_groups(out_ch), num_channels=out_ch, eps=1e-6)
        self.Conv_1 = conv3x3(out_ch, out_ch, init_scale=init_scale)          # Dropout_0 is identity when sampling
        if in_ch != out_ch or up or down:
            self.Conv_2 = conv1x1(in_ch, out_ch)
        self.skip_rescale, self.act, self.in_ch, self.out_ch = skip_rescale, act, in_ch, out_ch
        self._temb_bias = _TembBias()

    def forward(self, x, temb=None, x2=None):
        """x2: the block's input is torch.cat([x, x2], dim=1) (the up path's skip connections, reference models/ncsnpp.py:351)
        -- evaluated without the concatenation: GroupNorm_0 normalises the two tensors into one buffer, the 1x1 shortcut
        Conv_2 runs as two chained parts"""
        code = self.act.code
        parts = None
        # the shortcut Conv_2 convolves the RAW input: its dynamic range comes out of GroupNorm_0's statistics pass (a bound on
        # max |x| also bounds the FIR-resampled x: the taps of every phase are non-negative and sum to one)
        # ... unless the input already carries its maxima from the epilogue that produced it (the usual case: a block's output)
        need_amax = ((self.in_ch != self.out_ch or self.up or self.down) and ops.unbounded_amax() is not None
                     and (ops.amax_of(x) is None or (x2 is not None and ops.amax_of(x2) is None)))
        amax = None
        if x2 is not None:
            if self.Conv_0.epilogue_folds() and self.Conv_2.kernel_size == 1:
                parts = [x, x2]
                h = self.GroupNorm_0((x, x2), code, want_amax=need_amax)
            else:
                x = torch.cat([x, x2], dim=1)
        if parts is None:
            h = self.GroupNorm_0(x, code, want_amax=need_amax)
        if need_amax:
            h, amax = h
        if self.up:
            f = (lambda t: up_or_down_sampling.upsample_2d(t, self.fir_kernel, factor=2)) if self.fir else \
                (lambda t: up_or_down_sampling.naive_upsample_2d(t, factor=2))
            h = f(h)
            if parts is None:
                x = f(x)
            else:
                parts = [f(t) for t in parts]
        elif self.down:
            f = (lambda t: up_or_down_sampling.downsample_2d(t, self.fir_kernel, factor=2)) if self.fir else \
                (lambda t: up_or_down_sampling.naive_downsample_2d(t, factor=2))
            h = f(h)
            if parts is None:
                x = f(x)
            else:
                parts = [f(t) for t in parts]
        fold = self.Conv_0.epilogue_folds()
        if temb is not None and fold:
            # h += Dense_0(act(temb))[:, :, None, None]: one bias row per image in Conv_0's epilogue
            h = self.Conv_0(h, bounded=True, want_stats=True,    # GroupNorm_1 reads the result next
                            bias_rows=self._temb_bias.rows(self.Dense_0, self.Conv_0, temb, code))
        else:
            h = self.Conv_0(h, bounded=True)                     # act(GroupNorm(x)), FIR-resampled: bounded
            if temb is not None:
                t = self.Dense_0(temb, act_in=code)
                shift = torch.stack([torch.zeros_like(t), torch.ones_like(t), t], dim=-1).contiguous()
                h = ops.affine_act(h, shift, ops.ACT_NONE, out=h)
        h = self.GroupNorm_1(h, code)
        if self.in_ch != self.out_ch or self.up or self.down:
            x = self.Conv_2.forward_parts(parts, in_amax=amax) if parts is not None else self.Conv_2(x, in_amax=amax)
        elif parts is not None:
            x = torch.cat(parts, dim=1)
        if fold:                                                   # (x + Conv_1(h)) [/ sqrt 2] in Conv_1's epilogue
            return self.Conv_1(h, residual=x, bounded=True, out_scale=INV_SQRT2 if self.skip_rescale else 1.0,
                               feeds_conv=True, want_stats=True)   # a block's output: shortcuts / skip connections read it raw,
                                                                   # the next block's GroupNorm_0 takes its statistics
        h = self.Conv_1(h, bounded=True)
        return _skip(x, h, self.skip_rescale)
