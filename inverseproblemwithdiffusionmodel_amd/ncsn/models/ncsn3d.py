"""Temporal score network on 8 x 8 x T patches (mirror of the reference's ``ncsn/models/ncsn3d.py``:
NCSN3DShallow :123-224, the default "Diffusion3D" of helpers/load_model.py:26).  Input (B', kx*ky, T) or
(B', 1, kx, ky, T); all stages are dilated 3x3x3 convolutions (no spatial pooling), time is halved by a
(1,1,4)/stride-2 convolution and restored by its transpose -- both run as a tap gather + 1x1 MFMA convolution."""
import numpy as np
import torch
import torch.nn as nn

from . import get_sigmas
from .layers import get_act, get_normalization
from .layers3d import ResidualBlock, RefineBlock, Conv3d
from ... import ops


class _TemporalConv(ops.PackedWeightMixin, nn.Module):
    """nn.Conv3d / nn.ConvTranspose3d with kernel (1,1,4), stride (1,1,2), padding (0,0,1): parameter names and
    shapes of the torch modules ([Cout,Cin,1,1,4] resp. [Cin,Cout,1,1,4]); executed as gather + 1x1 conv."""

    def __init__(self, in_ch, out_ch, transposed):
        super().__init__()
        self.transposed, self.in_ch, self.out_ch = transposed, in_ch, out_ch
        shape = (in_ch, out_ch, 1, 1, 4) if transposed else (out_ch, in_ch, 1, 1, 4)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(out_ch))
        bound = 1.0 / ((out_ch if transposed else in_ch) * 4) ** 0.5
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)
        self._cache = ops.PackedWeightCache()

    def _pack(self, weight):
        w = weight[:, :, 0, 0, :]                                            # [a, b, 4]
        # gathered channel index = ci*4 + k : an ordinary 1x1 kernel [Cout][4*Cin][1][1]
        w1 = (w.permute(1, 0, 2) if self.transposed else w).reshape(self.out_ch, self.in_ch * 4, 1, 1)
        return ops.conv_weight(w1.contiguous())

    def packed(self):
        return self._cache.get(self.weight, "taps_" + ops.CONV_IMPL, self._pack)

    def forward(self, x):
        B, C, D, H, T = x.shape
        taps = ops.temporal_taps(x, 1 if self.transposed else 0)            # [B, 4C, D, H, T']; carries x's maxima (a gather)
        Tn = taps.shape[-1]
        dyn = ops.dynamic_range()
        y = ops.conv2d(taps.view(B, 4 * C, D * H, Tn), self.packed(), self.bias.data,
                       in_amax=ops.in_amax_for(taps) if dyn else None, want_amax=dyn)
        return ops.carry_amax(y, y.view(B, self.out_ch, D, H, Tn))


class NCSN3DShallow(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.logit_transform = config.data.logit_transform
        self.rescaled = config.data.rescaled
        self.norm = get_normalization(config, conditional=False)
        self.ngf = ngf = config.model.ngf
        self.num_classes = config.model.num_classes
        self.act = act = get_act(config)
        self.register_buffer('sigmas', get_sigmas(config))
        self.config = config
        ch3 = config.data.channels_3d
        self.begin_conv = Conv3d(ch3, ngf, 3, full_range=True)
        self.normalizer = self.norm(ngf)
        self.end_conv = Conv3d(ngf, ch3, 3)
        kw = dict(act=act, normalization=self.norm)
        self.res1 = nn.ModuleList([ResidualBlock(ngf, ngf, resample=None, **kw),
                                   ResidualBlock(ngf, ngf, resample=None, **kw)])
        self.res3 = nn.ModuleList([ResidualBlock(ngf, 2 * ngf, resample='down', dilation=2, **kw),
                                   ResidualBlock(2 * ngf, 2 * ngf, resample=None, dilation=2, **kw)])
        self.res4 = nn.ModuleList([ResidualBlock(2 * ngf, 2 * ngf, resample='down', dilation=4, **kw),
                                   ResidualBlock(2 * ngf, 2 * ngf, resample=None, dilation=4, **kw)])
        self.refine1 = RefineBlock([2 * ngf], 2 * ngf, act=act, start=True)
        self.refine2 = RefineBlock([2 * ngf, 2 * ngf], 2 * ngf, act=act)
        self.refine3 = RefineBlock([ngf, ngf], ngf, act=act)
        self.conv_temporal_down = _TemporalConv(2 * ngf, 2 * ngf, transposed=False)
        self.conv_temporal_up = _TemporalConv(2 * ngf, ngf, transposed=True)
        self._in_coef = {}

    def _stage(self, module, x):
        n = len(module)
        for i, m in enumerate(module):
            x = m(x, want_act=(i == n - 1))
        return x

    def forward(self, x, y):
        with ops.amax_scope():                           # one zero-fill for all the per-image maxima slots of the evaluation
            return self._forward(x, y)

    def _forward(self, x, y):
        if not x.is_cuda:
            raise RuntimeError("NCSN3DShallow: expected GPU tensors (no CPU fallback in this build)")
        x_dim = x.dim()
        if x_dim == 3:                                   # (B, kx*ky, T) -> (B, 1, kx, ky, T): a pure view
            k = int(np.sqrt(self.config.data.channels))
            x = x.reshape(x.shape[0], 1, k, k, x.shape[-1])
        x = x.contiguous().float()
        if not self.logit_transform and not self.rescaled:
            key = (x.shape[0], x.shape[1], str(x.device))
            if key not in self._in_coef:
                self._in_coef[key] = torch.tensor([0.5, 2.0, 0.0], device=x.device).repeat(x.shape[0], x.shape[1], 1)
            output = self.begin_conv(x, self._in_coef[key])           # 2x - 1 folded into the input staging
        else:
            output = self.begin_conv(x)
        code = self.act.code
        layer1 = self._stage(self.res1, output)                        # (B, ngf, 8, 8, T)
        layer2 = self._stage(self.res3, layer1[0])                     # (B, 2ngf, 8, 8, T)
        layer3 = self.conv_temporal_down(layer2[0])                    # (B, 2ngf, 8, 8, T/2)
        layer4 = self._stage(self.res4, layer3)
        ref1 = self.refine1([layer4[0]], layer4[0].shape[2:], [layer4[1]], want_act=True)
        ref2 = self.refine2([layer3, ref1[0]], layer3.shape[2:], [None, ref1[1]], want_act=False, feeds_conv=True)
        ref3 = self.conv_temporal_up(ref2)                             # (B, ngf, 8, 8, T)
        output = self.refine3([layer1[0], ref3], layer1[0].shape[2:], [layer1[1], None], want_act=False)
        output = self.end_conv(self.normalizer(output, code))
        sig = self.sigmas if self.sigmas.dtype == torch.float32 else self.sigmas.to(torch.float32)
        output = ops.div_sigma(output, sig, y.to(torch.int64))
        if x_dim == 3:
            output = output.reshape(output.shape[0], -1, output.shape[-1])
        return output
