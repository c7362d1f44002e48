"""NCSNv2 score networks (mirror of the reference's ``ncsn/models/ncsnv2.py``: NCSNv2 :11-101,
NCSNv2Deeper :104-195, NCSNv2Deepest :198-299).  ``scorenet(x (B,C,H,W) f32, labels (B,) int64)`` on GPU
tensors; same constructor (``Ctor(config)``), ``.sigmas`` buffer, ``.config`` and state-dict keys."""
import torch
import torch.nn as nn

from . import get_sigmas
from .layers import ResidualBlock, RefineBlock, Conv2d, get_act, get_normalization
from ... import ops


class _NCSNv2Base(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.logit_transform = config.data.logit_transform
        self.rescaled = config.data.rescaled
        self.norm = get_normalization(config, conditional=False)
        self.ngf = config.model.ngf
        self.num_classes = config.model.num_classes
        self.act = get_act(config)
        self.register_buffer('sigmas', get_sigmas(config))
        self.config = config
        self._in_coef = {}

    def _compute_cond_module(self, module, x):
        """a res stage; its last block also emits the activated copy the RefineNet branch starts from"""
        n = len(module)
        for i, m in enumerate(module):
            x = m(x, want_act=(i == n - 1))
        return x

    def _stage(self, cin, cout, resample=None, dilation=None):
        kw = dict(act=self.act, normalization=self.norm)
        if dilation is not None:
            kw["dilation"] = dilation
        return nn.ModuleList([ResidualBlock(cin, cout, resample=resample, **kw),
                              ResidualBlock(cout, cout, resample=None, **kw)])

    def _begin(self, x):
        if not x.is_cuda:
            raise RuntimeError("score network: expected GPU tensors (no CPU fallback in this build)")
        x = x.contiguous().float()
        if not self.logit_transform and not self.rescaled:
            # h = 2x - 1 folded into begin_conv's input staging as (x - 0.5) * 2 + 0
            key = (x.shape[0], x.shape[1], str(x.device))
            if key not in self._in_coef:
                self._in_coef[key] = torch.tensor([0.5, 2.0, 0.0], device=x.device).repeat(x.shape[0], x.shape[1], 1)
            return self.begin_conv(x, self._in_coef[key])
        return self.begin_conv(x)

    def _end(self, output, x, y):
        output = self.end_conv(self.normalizer(output, self.act.code))
        sig = self.sigmas if self.sigmas.dtype == torch.float32 else self.sigmas.to(torch.float32)
        return ops.div_sigma(output, sig, y.to(torch.int64))

    @staticmethod
    def _refine(block, pairs, shape, want_act=True):
        return block([p[0] for p in pairs], shape, [p[1] for p in pairs], want_act=want_act)


class NCSNv2Deepest(_NCSNv2Base):
    def __init__(self, config):
        super().__init__(config)
        ngf, ch = self.ngf, config.data.channels
        self.begin_conv = Conv2d(ch, ngf, 3, full_range=True)
        self.normalizer = self.norm(ngf)
        self.end_conv = Conv2d(ngf, ch, 3)
        self.res1 = self._stage(ngf, ngf)
        self.res2 = self._stage(ngf, 2 * ngf, 'down')
        self.res3 = self._stage(2 * ngf, 2 * ngf, 'down')
        self.res31 = self._stage(2 * ngf, 2 * ngf, 'down')
        self.res4 = self._stage(2 * ngf, 4 * ngf, 'down', dilation=2)
        self.res5 = self._stage(4 * ngf, 4 * ngf, 'down', dilation=4)
        self.refine1 = RefineBlock([4 * ngf], 4 * ngf, act=self.act, start=True)
        self.refine2 = RefineBlock([4 * ngf, 4 * ngf], 2 * ngf, act=self.act)
        self.refine3 = RefineBlock([2 * ngf, 2 * ngf], 2 * ngf, act=self.act)
        self.refine31 = RefineBlock([2 * ngf, 2 * ngf], 2 * ngf, act=self.act)
        self.refine4 = RefineBlock([2 * ngf, 2 * ngf], ngf, act=self.act)
        self.refine5 = RefineBlock([ngf, ngf], ngf, act=self.act, end=True)

    def forward(self, x, y):
        with ops.amax_scope():                           # one zero-fill for all the per-image maxima slots of the evaluation
            return self._forward(x, y)

    def _forward(self, x, y):
        output = self._begin(x)
        layer1 = self._compute_cond_module(self.res1, output)
        layer2 = self._compute_cond_module(self.res2, layer1[0])
        layer3 = self._compute_cond_module(self.res3, layer2[0])
        layer31 = self._compute_cond_module(self.res31, layer3[0])
        layer4 = self._compute_cond_module(self.res4, layer31[0])
        layer5 = self._compute_cond_module(self.res5, layer4[0])
        ref1 = self._refine(self.refine1, [layer5], layer5[0].shape[2:])
        ref2 = self._refine(self.refine2, [layer4, ref1], layer4[0].shape[2:])
        ref31 = self._refine(self.refine31, [layer31, ref2], layer31[0].shape[2:])
        ref3 = self._refine(self.refine3, [layer3, ref31], layer3[0].shape[2:])
        ref4 = self._refine(self.refine4, [layer2, ref3], layer2[0].shape[2:])
        output = self._refine(self.refine5, [layer1, ref4], layer1[0].shape[2:], want_act=False)
        return self._end(output, x, y)


class NCSNv2Deeper(_NCSNv2Base):
    def __init__(self, config):
        super().__init__(config)
        ngf, ch = self.ngf, config.data.channels
        self.begin_conv = Conv2d(ch, ngf, 3, full_range=True)
        self.normalizer = self.norm(ngf)
        self.end_conv = Conv2d(ngf, ch, 3)
        self.res1 = self._stage(ngf, ngf)
        self.res2 = self._stage(ngf, 2 * ngf, 'down')
        self.res3 = self._stage(2 * ngf, 2 * ngf, 'down')
        self.res4 = self._stage(2 * ngf, 4 * ngf, 'down', dilation=2)
        self.res5 = self._stage(4 * ngf, 4 * ngf, 'down', dilation=4)
        self.refine1 = RefineBlock([4 * ngf], 4 * ngf, act=self.act, start=True)
        self.refine2 = RefineBlock([4 * ngf, 4 * ngf], 2 * ngf, act=self.act)
        self.refine3 = RefineBlock([2 * ngf, 2 * ngf], 2 * ngf, act=self.act)
        self.refine4 = RefineBlock([2 * ngf, 2 * ngf], ngf, act=self.act)
        self.refine5 = RefineBlock([ngf, ngf], ngf, act=self.act, end=True)

    def forward(self, x, y):
        with ops.amax_scope():                           # one zero-fill for all the per-image maxima slots of the evaluation
            return self._forward(x, y)

    def _forward(self, x, y):
        output = self._begin(x)
        layer1 = self._compute_cond_module(self.res1, output)
        layer2 = self._compute_cond_module(self.res2, layer1[0])
        layer3 = self._compute_cond_module(self.res3, layer2[0])
        layer4 = self._compute_cond_module(self.res4, layer3[0])
        layer5 = self._compute_cond_module(self.res5, layer4[0])
        ref1 = self._refine(self.refine1, [layer5], layer5[0].shape[2:])
        ref2 = self._refine(self.refine2, [layer4, ref1], layer4[0].shape[2:])
        ref3 = self._refine(self.refine3, [layer3, ref2], layer3[0].shape[2:])
        ref4 = self._refine(self.refine4, [layer2, ref3], layer2[0].shape[2:])
        output = self._refine(self.refine5, [layer1, ref4], layer1[0].shape[2:], want_act=False)
        return self._end(output, x, y)


class NCSNv2(_NCSNv2Base):
    def __init__(self, config):
        super().__init__(config)
        ngf, ch = self.ngf, config.data.channels
        self.begin_conv = Conv2d(ch, ngf, 3, full_range=True)
        self.normalizer = self.norm(ngf)
        self.end_conv = Conv2d(ngf, ch, 3)
        self.res1 = self._stage(ngf, ngf)
        self.res2 = self._stage(ngf, 2 * ngf, 'down')
        self.res3 = self._stage(2 * ngf, 2 * ngf, 'down', dilation=2)
        # the reference's 28-pixel branch (ncsnv2.py:50-56) passes adjust_padding=True to a DILATED block, whose
        # constructor ignores it (layers.py:410-414): both branches build the same modules (pinned by g23 'v2_28')
        self.res4 = self._stage(2 * ngf, 2 * ngf, 'down', dilation=4)
        self.refine1 = RefineBlock([2 * ngf], 2 * ngf, act=self.act, start=True)
        self.refine2 = RefineBlock([2 * ngf, 2 * ngf], 2 * ngf, act=self.act)
        self.refine3 = RefineBlock([2 * ngf, 2 * ngf], ngf, act=self.act)
        self.refine4 = RefineBlock([ngf, ngf], ngf, act=self.act, end=True)

    def forward(self, x, y):
        with ops.amax_scope():                           # one zero-fill for all the per-image maxima slots of the evaluation
            return self._forward(x, y)

    def _forward(self, x, y):
        output = self._begin(x)
        layer1 = self._compute_cond_module(self.res1, output)
        layer2 = self._compute_cond_module(self.res2, layer1[0])
        layer3 = self._compute_cond_module(self.res3, layer2[0])
        layer4 = self._compute_cond_module(self.res4, layer3[0])
        ref1 = self._refine(self.refine1, [layer4], layer4[0].shape[2:])
        ref2 = self._refine(self.refine2, [layer3, ref1], layer3[0].shape[2:])
        ref3 = self._refine(self.refine3, [layer2, ref2], layer2[0].shape[2:])
        output = self._refine(self.refine4, [layer1, ref3], layer1[0].shape[2:], want_act=False)
        return self._end(output, x, y)
