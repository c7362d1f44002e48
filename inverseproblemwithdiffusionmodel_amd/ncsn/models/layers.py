"""RefineNet-style building blocks of the NCSNv2 family (mirror of the reference's
``ncsn/models/layers.py``: get_act :11-23, conv1x1/conv3x3/dilated_conv3x3 :28-60, CRPBlock :62-83,
RCUBlock :112-134, MSFBlock :165-184, RefineBlock :214-249, ConvMeanPool :291-313, ResidualBlock :401-456).

Same parameter names / state-dict keys as the reference, different execution: every block is a short
chain of libipdm.so launches.  Bias and residual adds live in the convolution epilogue, which can also
emit the ACTIVATED copy of its result: the RCU / CRP chains (`x = act(x); x = conv(x)` over and over) then
never run a separate activation pass -- blocks hand each other (raw, activated) pairs.  InstanceNorm++ +
ELU in front of the ResidualBlock convolutions is one fused affine+activation kernel; the 5x5 max-pool,
2x2 mean-pool and bilinear accumulate(+activation) are single kernels."""
import os
from functools import partial

import torch
import torch.nn as nn

from ... import ops, _lib
from .normalization import InstanceNorm2dPlus, get_normalization  # noqa: F401


class _Act:
    """activation handle: carries the kernel code; callable on tensors like the reference's nn.ELU()"""

    def __init__(self, name):
        self.name = name
        self.code = ops.ACT_CODES[name]

    def __call__(self, x):
        return ops.act(x, self.code)


# Winograd F(2x2,3x3) for eligible layers (3x3, dilation 1, wide images); IPDM_WINOGRAD=0 forces the direct kernel
USE_WINOGRAD = os.environ.get("IPDM_WINOGRAD", "1") != "0"
# ConvMeanPool (+ shortcut + activation) as one pooled-epilogue launch; IPDM_FUSE_POOL=0 runs conv, mean-pool, add, act apart
FUSE_POOL = os.environ.get("IPDM_FUSE_POOL", "1") != "0"


def get_act(config):
    name = config.model.nonlinearity.lower()
    if name not in ("elu", "relu", "lrelu", "swish"):
        raise NotImplementedError("activation function does not exist!")
    return _Act(name)


def _act_code(act):
    if act is None:
        return ops.ACT_NONE
    if isinstance(act, _Act):
        return act.code
    if isinstance(act, nn.ELU):
        return ops.ACT_ELU
    if isinstance(act, nn.ReLU):
        return ops.ACT_RELU
    raise NotImplementedError(f"no kernel code for activation {act!r}")


class Conv2d(ops.PackedWeightMixin, nn.Module):
    """stride-1 'same' convolution, kernel 1 or 3, optional dilation; weight in the reference's layout
    [Cout, Cin, k, k] (what checkpoints carry), repacked once for the selected kernel family (ops.conv_weight)."""

    def __init__(self, in_planes, out_planes, kernel_size=3, dilation=1, bias=True, ndim=2, full_range=False):
        """full_range: the layer reads a RAW network input through a fused input transform (begin_conv's `2x - 1`), whose range no
        producer bounds: on the f16x2 family it runs the three-piece bf16 kernels instead (the whole fp32 exponent range)"""
        super().__init__()
        assert kernel_size in (1, 3) and ndim in (2, 3)
        self.in_planes, self.out_planes, self.kernel_size, self.dilation = in_planes, out_planes, kernel_size, dilation
        self.ndim = ndim
        self.full_range = full_range
        self.weight = nn.Parameter(torch.empty(out_planes, in_planes, *([kernel_size] * ndim)))
        self.bias = nn.Parameter(torch.empty(out_planes)) if bias else None
        bound = 1.0 / (in_planes * kernel_size ** ndim) ** 0.5
        nn.init.uniform_(self.weight, -bound, bound)
        if bias:
            nn.init.uniform_(self.bias, -bound, bound)
        self._cache = ops.PackedWeightCache()

    def _cached(self, kind, build):
        return self._cache.get(self.weight, kind, build)

    def packed_wino(self):
        return self._cached("wino_f32", ops.conv_wino_weight)

    def packed_wino_bx3(self):
        return self._cached("wino_" + ops.CONV_IMPL, ops.conv_wino_split_weight)

    def packed_wino1d(self):
        return self._cached("wino1d", ops.conv_wino1d_weight)

    def fuses_input(self, x, coef):
        """True when this layer takes act(InstanceNorm++(x)) as (raw x, coefficients): the 1-D Winograd kernel applies the affine +
        ELU to the raw rows in its producer, so the normalised tensor is never written (IPDM_WINO1D_FIN=0: the separate pass)"""
        return (ops.WINO1D_FIN and USE_WINOGRAD and ops.split_impl() and self.ndim == 2 and self.kernel_size == 3 and x.dim() == 4
                and x.data_ptr() % 16 == 0 and (not ops.dynamic_range() or getattr(coef, "_ipdm_amax_bound", None) is not None)
                and ops.wino1d_pays(self.in_planes, self.out_planes, x.shape[2], x.shape[3], self.dilation))

    def packed(self):
        if self.full_range and ops.CONV_IMPL == "hx2":
            return self._cached("direct_bx3", lambda w: ops.conv_weight(w, impl="bx3"))
        return self._cached("direct_" + ops.CONV_IMPL, ops.conv_weight)

    def forward(self, x, coef=None, act=ops.ACT_NONE, residual=None, out=None, act_out=ops.ACT_NONE, raw=True,
                want_stats=False, in_amax=None, feeds_conv=True, res_second=False):
        """want_stats: the result goes into an InstanceNorm++ next (the Winograd kernel's statistics epilogue then spares
        that normalisation its pass over the tensor; ignored by the other kernels).
        f16x2 family: the input's per-image maxima come with the tensor from its producer (ops.in_amax_for; `in_amax` overrides)
        and this layer's epilogue accumulates the maxima of what it stores for ITS consumer -- every convolution of the network
        then runs with the dynamic range (any fp32 input is in range) at no pass over any tensor.
        feeds_conv=False: the result is read by normalisations / residual adds / resizes only (they bound or measure what THEY
        hand on), so the epilogue skips the maxima -- an atomic at the end of a launch costs its round trip (~2.5 us)."""
        bias = None if self.bias is None else self.bias.data
        dyn = ops.dynamic_range()
        produce = dyn and feeds_conv
        if in_amax is None and dyn and coef is None and act == ops.ACT_NONE:
            in_amax = ops.in_amax_for(x)
        if coef is not None and act == ops.ACT_ELU and out is None and self.fuses_input(x, coef):
            return ops.conv2d_wino_bx3(x, self.packed_wino1d(), bias, residual, act_out=act_out, raw=raw, want_stats=want_stats,
                                       in_amax=getattr(coef, "_ipdm_amax_bound", None) if dyn else None, want_amax=produce,
                                       res_second=res_second, coef=coef, act=act)
        if (self.ndim == 2 and self.kernel_size == 3 and self.dilation == 1 and act == ops.ACT_NONE
                and residual is None and out is None and act_out == ops.ACT_NONE and raw and x.dim() == 4
                and (self.in_planes <= 3 or (self.out_planes <= 3 and coef is None))
                and ops.conv3x3_thin_ok(self.in_planes, self.out_planes, x.shape[2], x.shape[3])):
            return ops.conv3x3_thin(x, self.weight.data, bias, coef)   # first / last layer: streaming kernels
        if (USE_WINOGRAD and ops.CONV_IMPL == "f32" and self.ndim == 2 and self.kernel_size == 3 and coef is None and act == ops.ACT_NONE
                and out is None
                and ops.conv_wino_supported(self.in_planes, self.out_planes, x.shape[2], x.shape[3], self.dilation)):
            return ops.conv2d_wino(x, self.packed_wino(), bias, residual, act_out=act_out, raw=raw,
                                   dilation=self.dilation)
        if (USE_WINOGRAD and ops.split_impl() and self.ndim == 2 and self.kernel_size == 3 and coef is None
                and act == ops.ACT_NONE and out is None and ops.wino_bx3_pays(self.in_planes, self.out_planes, x.shape[2],
                                                                              x.shape[3], self.dilation)):
            one_d = (x.data_ptr() % 16 == 0 and act_out in (ops.ACT_NONE, ops.ACT_ELU, ops.ACT_COPY)
                     and (ops.WINO1D_STATS or not (want_stats and raw and ops.USE_STATS_EPILOGUE))
                     and ops.wino1d_pays(self.in_planes, self.out_planes, x.shape[2], x.shape[3], self.dilation))
            return ops.conv2d_wino_bx3(x, self.packed_wino1d() if one_d else self.packed_wino_bx3(), bias, residual,
                                       act_out=act_out, raw=raw, dilation=self.dilation, want_stats=want_stats, in_amax=in_amax,
                                       want_amax=produce, res_second=res_second)
        if (self.ndim == 3 and self.kernel_size == 3 and self.dilation == 1 and coef is None and act == ops.ACT_NONE and out is None
                and x.dim() == 5 and x.data_ptr() % 16 == 0 and act_out in (ops.ACT_NONE, ops.ACT_ELU, ops.ACT_COPY)
                and ops.wino1d_vol_pays(self.in_planes, self.out_planes, x.shape[2], x.shape[3], x.shape[4])):
            return ops.conv3d(x, self._cached("wino1d_vol", ops.conv_wino1d_weight3d), bias, residual=residual, act_out=act_out,
                              raw=raw, in_amax=in_amax, want_amax=produce, res_second=res_second)
        if self.ndim == 3:
            return ops.conv3d(x, self.packed(), bias, coef, act, residual, self.dilation, act_out=act_out, raw=raw,
                              in_amax=in_amax, want_amax=produce, res_second=res_second)
        return ops.conv2d(x, self.packed(), bias, coef, act, residual, self.dilation, out=out, act_out=act_out, raw=raw,
                          in_amax=in_amax, want_amax=produce, res_second=res_second)


def conv1x1(in_planes, out_planes, stride=1, bias=True, spec_norm=False, ndim=2):
    assert stride == 1 and not spec_norm
    return Conv2d(in_planes, out_planes, 1, bias=bias, ndim=ndim)


def conv3x3(in_planes, out_planes, stride=1, bias=True, spec_norm=False, ndim=2):
    assert stride == 1 and not spec_norm
    return Conv2d(in_planes, out_planes, 3, bias=bias, ndim=ndim)


def dilated_conv3x3(in_planes, out_planes, dilation, bias=True, spec_norm=False, ndim=2):
    assert not spec_norm
    return Conv2d(in_planes, out_planes, 3, dilation=dilation, bias=bias, ndim=ndim)


def _maxpool(x):
    return ops.maxpool3d5(x) if x.dim() == 5 else ops.maxpool5(x)


class ConvMeanPool(nn.Module):
    def __init__(self, input_dim, output_dim, kernel_size=3, biases=True, adjust_padding=False, spec_norm=False, ndim=2):
        super().__init__()
        if adjust_padding or spec_norm:
            raise NotImplementedError("adjust_padding / spec_norm are unused by every shipped config")
        if ndim != 2:
            raise NotImplementedError("the 8-way 3-D ConvMeanPool is unused by NCSN3DShallow (all its stages are dilated)")
        self.conv = Conv2d(input_dim, output_dim, kernel_size, bias=biases)

    def forward(self, inputs, feeds_conv=True):
        if self.conv.kernel_size == 1:
            # a 1x1 convolution (+ bias) commutes with the 2x2 mean: pool first, a quarter of the multiply-adds and of
            # the bytes (same value up to fp32 rounding order)
            return self.conv(ops.meanpool2(inputs), feeds_conv=feeds_conv)
        return ops.meanpool2(self.conv(inputs, feeds_conv=feeds_conv))

    def fused(self, inputs, residual=None, act_out=ops.ACT_NONE, feeds_conv=True, coef=None, act=ops.ACT_NONE):
        """3x3 ConvMeanPool (+ pooled-size residual, + activated copy) in ONE launch: the Winograd kernel's 2x2 output tile
        is the pooling window.  -> out or (out, out_act); None where the pooled epilogue is not built for this layer."""
        c = self.conv
        if not (FUSE_POOL and USE_WINOGRAD and ops.split_impl() and c.ndim == 2 and c.kernel_size == 3 and c.dilation == 1
                and inputs.shape[2] % 2 == 0 and inputs.shape[3] % 2 == 0
                and ops.wino_bx3_pays(c.in_planes, c.out_planes, inputs.shape[2], inputs.shape[3], 1)):
            return None
        one_d = (inputs.data_ptr() % 16 == 0 and act_out in (ops.ACT_NONE, ops.ACT_ELU, ops.ACT_COPY) and ops.WINO1D_STATS
                 and ops.wino1d_pays(c.in_planes, c.out_planes, inputs.shape[2], inputs.shape[3], 1))
        try:
            if coef is not None:                         # fused input: the 1-D kernel's only (the caller asked c.fuses_input)
                if not one_d:
                    return None
                return ops.conv2d_wino_bx3(inputs, c.packed_wino1d(), None if c.bias is None else c.bias.data, residual,
                                           act_out=act_out, pool2=True, want_stats=True,
                                           in_amax=getattr(coef, "_ipdm_amax_bound", None) if ops.dynamic_range() else None,
                                           want_amax=ops.dynamic_range() and feeds_conv, coef=coef, act=act)
            return ops.conv2d_wino_bx3(inputs, c.packed_wino1d() if one_d else c.packed_wino_bx3(),
                                       None if c.bias is None else c.bias.data, residual,
                                       act_out=act_out, pool2=True, want_stats=True,     # a block's result: normalised next
                                       in_amax=ops.in_amax_for(inputs), want_amax=ops.dynamic_range() and feeds_conv)
        except _lib.IpdmUnsupported:
            return None


class CRPBlock(nn.Module):
    def __init__(self, features, n_stages, act=None, maxpool=True, spec_norm=False, ndim=2):
        super().__init__()
        if not maxpool:
            raise NotImplementedError("average-pool CRP is unused by every shipped config")
        self.convs = nn.ModuleList([conv3x3(features, features, bias=False, spec_norm=spec_norm, ndim=ndim)
                                    for _ in range(n_stages)])
        self.n_stages = n_stages
        self.act = act

    def forward(self, x, x_act=None, want_act=False):
        """reference: x = act(x); path = x; repeat: path = conv(maxpool(path)); x = path + x.
        Takes the already-activated input when the producer emitted it; returns (x, act(x) or None)."""
        code = _act_code(self.act)
        x = x_act if x_act is not None else ops.act(x, code)
        path, out_act = x, None
        for i in range(self.n_stages):
            pooled = _maxpool(path)
            if i == self.n_stages - 1:                       # x = conv(pool(path)) + x in one launch
                if want_act:
                    x, out_act = self.convs[i](pooled, residual=x, act_out=code)
                else:
                    x = self.convs[i](pooled, residual=x)
            elif ops.split_impl():                           # path = conv(pool(path)); x = path + x: both leave ONE epilogue
                path, x = self.convs[i](pooled, residual=x, res_second=True, act_out=ops.ACT_COPY)
            else:
                path = self.convs[i](pooled)
                x = ops.add(path, x)
        return x, out_act


class RCUBlock(nn.Module):
    def __init__(self, features, n_blocks, n_stages, act=None, spec_norm=False, ndim=2):
        super().__init__()
        for i in range(n_blocks):
            for j in range(n_stages):
                setattr(self, '{}_{}_conv'.format(i + 1, j + 1), conv3x3(features, features, bias=False,
                                                                         spec_norm=spec_norm, ndim=ndim))
        self.stride = 1
        self.n_blocks = n_blocks
        self.n_stages = n_stages
        self.act = act

    def forward(self, x, x_act=None, want_act=False, feeds_conv=True):
        """reference: per block  residual = x; (x = act(x); x = conv(x)) x n_stages; x += residual.
        Inner stages only ever feed the next activation, so they write just the activated copy.
        feeds_conv=False: the block's RESULT is normalised next (the network's last RCU): no maxima for it."""
        code = _act_code(self.act)
        for i in range(self.n_blocks):
            residual = x
            a = x_act if x_act is not None else ops.act(x, code)
            for j in range(self.n_stages):
                conv = getattr(self, '{}_{}_conv'.format(i + 1, j + 1))
                if j < self.n_stages - 1:
                    _, a = conv(a, act_out=code, raw=False)
                elif i < self.n_blocks - 1 or want_act:
                    x, x_act = conv(a, residual=residual, act_out=code)
                else:
                    x, x_act = conv(a, residual=residual, feeds_conv=feeds_conv), None
        return x, x_act


class MSFBlock(nn.Module):
    def __init__(self, in_planes, features, spec_norm=False, ndim=2):
        super().__init__()
        assert isinstance(in_planes, (list, tuple))
        self.convs = nn.ModuleList([conv3x3(p, features, bias=True, spec_norm=spec_norm, ndim=ndim) for p in in_planes])
        self.features = features

    def forward(self, xs, shape, act_out=ops.ACT_NONE):
        """sum_i bilinear (3-D: trilinear) (conv_i(xs[i]));  act_out: return act(sum) instead (all the following CRP block needs)"""
        shape = tuple(int(s) for s in shape)
        sums = None
        n = len(self.convs)
        for i, conv in enumerate(self.convs):
            last_act = act_out if i == n - 1 else ops.ACT_NONE
            if tuple(xs[i].shape[2:]) == shape:                   # bilinear resize to the same size is exact
                if last_act != ops.ACT_NONE:
                    _, sums = conv(xs[i], residual=sums, act_out=last_act, raw=False)
                else:
                    sums = conv(xs[i], residual=sums, feeds_conv=False)      # only ever a residual / resize operand
            else:
                h = conv(xs[i], feeds_conv=False)
                resize = ops.trilinear if h.dim() == 5 else ops.bilinear
                sums = resize(h, shape, out=sums, accumulate=sums is not None, act=last_act,
                              want_amax=i == n - 1 and ops.dynamic_range())      # the block's result feeds CRP's convolutions
        return sums


class RefineBlock(nn.Module):
    def __init__(self, in_planes, features, act=None, start=False, end=False, maxpool=True, spec_norm=False, ndim=2):
        super().__init__()
        assert isinstance(in_planes, (tuple, list))
        self.n_blocks = n_blocks = len(in_planes)
        self.adapt_convs = nn.ModuleList([RCUBlock(in_planes[i], 2, 2, act, spec_norm=spec_norm, ndim=ndim)
                                          for i in range(n_blocks)])
        self.output_convs = RCUBlock(features, 3 if end else 1, 2, act, spec_norm=spec_norm, ndim=ndim)
        if not start:
            self.msf = MSFBlock(in_planes, features, spec_norm=spec_norm, ndim=ndim)
        self.crp = CRPBlock(features, 2, act, maxpool=maxpool, spec_norm=spec_norm, ndim=ndim)
        self.act = act

    def forward(self, xs, output_shape, xs_act=None, want_act=False, feeds_conv=None):
        """xs: raw inputs; xs_act: their activated copies where a producer emitted them (else None entries).
        Returns the raw output, or (raw, activated) when want_act.  feeds_conv (default: want_act): a convolution reads the result
        (another refine block, a temporal convolution) -- otherwise it is normalised next and carries no maxima."""
        assert isinstance(xs, (tuple, list))
        code = _act_code(self.act)
        xs_act = [None] * len(xs) if xs_act is None else xs_act
        single = self.n_blocks == 1
        hs = [self.adapt_convs[i](xs[i], xs_act[i], want_act=single) for i in range(len(xs))]
        if single:
            h, h_act = self.crp(hs[0][0], hs[0][1], want_act=True)
        else:
            a = self.msf([h[0] for h in hs], output_shape, act_out=code)
            h, h_act = self.crp(None, a, want_act=True)
        out, out_act = self.output_convs(h, h_act, want_act=want_act, feeds_conv=want_act if feeds_conv is None else feeds_conv)
        return (out, out_act) if want_act else out


class ResidualBlock(nn.Module):
    def __init__(self, input_dim, output_dim, resample=None, act=None, normalization=InstanceNorm2dPlus,
                 adjust_padding=False, dilation=None, spec_norm=False, ndim=2):
        super().__init__()
        conv3x3_ = partial(conv3x3, ndim=ndim)
        conv1x1_ = partial(conv1x1, ndim=ndim)
        dilated_conv3x3_ = partial(dilated_conv3x3, ndim=ndim)
        ConvMeanPool_ = partial(ConvMeanPool, ndim=ndim)
        self.non_linearity = act
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.resample = resample
        self.normalization = normalization
        if resample == 'down':
            if dilation is not None:
                self.conv1 = dilated_conv3x3_(input_dim, input_dim, dilation=dilation, spec_norm=spec_norm)
                self.normalize2 = normalization(input_dim)
                self.conv2 = dilated_conv3x3_(input_dim, output_dim, dilation=dilation, spec_norm=spec_norm)
                conv_shortcut = partial(dilated_conv3x3_, dilation=dilation, spec_norm=spec_norm)
            else:
                self.conv1 = conv3x3_(input_dim, input_dim, spec_norm=spec_norm)
                self.normalize2 = normalization(input_dim)
                self.conv2 = ConvMeanPool_(input_dim, output_dim, 3, adjust_padding=adjust_padding, spec_norm=spec_norm)
                conv_shortcut = partial(ConvMeanPool_, kernel_size=1, adjust_padding=adjust_padding,
                                        spec_norm=spec_norm)
        elif resample is None:
            if dilation is not None:
                conv_shortcut = partial(dilated_conv3x3_, dilation=dilation, spec_norm=spec_norm)
                self.conv1 = dilated_conv3x3_(input_dim, output_dim, dilation=dilation, spec_norm=spec_norm)
                self.normalize2 = normalization(output_dim)
                self.conv2 = dilated_conv3x3_(output_dim, output_dim, dilation=dilation, spec_norm=spec_norm)
            else:
                conv_shortcut = partial(conv1x1_, spec_norm=spec_norm)
                self.conv1 = conv3x3_(input_dim, output_dim, spec_norm=spec_norm)
                self.normalize2 = normalization(output_dim)
                self.conv2 = conv3x3_(output_dim, output_dim, spec_norm=spec_norm)
        else:
            raise Exception('invalid resample value')
        if output_dim != input_dim or resample is not None:
            self.shortcut = conv_shortcut(input_dim, output_dim)
        self.normalize1 = normalization(input_dim)

    def forward(self, x, want_act=False):
        """norm -> act -> conv1 -> norm -> act -> conv2 (+ shortcut).  want_act: also return act(out) -- the last block of a
        stage, whose result the next stage's shortcut convolution and the RefineNet branch read (the others' results meet
        normalisations and residual adds only: no maxima needed)."""
        code = _act_code(self.non_linearity)
        h = self._norm_act_conv(self.normalize1, self.conv1, x, code, want_stats=True, feeds_conv=False)
        a2 = None                                                # (act(normalize2(h)): materialised only where no kernel fuses it)
        if self.output_dim == self.input_dim and self.resample is None:
            shortcut = x
        else:
            shortcut = self.shortcut(x, feeds_conv=False)          # (a residual operand only)
        if isinstance(self.conv2, ConvMeanPool):
            plain_norm = isinstance(self.normalize2, InstanceNorm2dPlus) and self.conv2.conv.ndim == 2
            coef2 = self.normalize2.coef(h) if plain_norm else None
            fin = plain_norm and code == ops.ACT_ELU and self.conv2.conv.fuses_input(h, coef2)
            a2 = h if fin else (ops.affine_act(h, coef2, code) if plain_norm else self.normalize2(h, code))
            fused = self.conv2.fused(a2, residual=shortcut, act_out=code if want_act else ops.ACT_NONE, feeds_conv=want_act,
                                     coef=coef2 if fin else None, act=code if fin else ops.ACT_NONE)
            if fused is not None:                        # conv + 2x2 mean + shortcut (+ activated copy): one launch
                return fused
            if fin:
                a2 = ops.affine_act(h, coef2, code)
            out = ops.add(shortcut, self.conv2(a2))
            return (out, ops.act(out, code)) if want_act else out
        # the block's result is what the next block normalises first
        if want_act:
            return self._norm_act_conv(self.normalize2, self.conv2, h, code, residual=shortcut, act_out=code, want_stats=True)
        return self._norm_act_conv(self.normalize2, self.conv2, h, code, residual=shortcut, want_stats=True, feeds_conv=False)

    @staticmethod
    def _norm_act_conv(norm, conv, x, code, **kw):
        """conv(act(norm(x))): one launch where the convolution kernel takes the normalisation's coefficients (Conv2d.fuses_input),
        else the affine + activation pass and the convolution"""
        if code == ops.ACT_ELU and isinstance(conv, Conv2d) and conv.ndim == 2 and isinstance(norm, InstanceNorm2dPlus):
            coef = norm.coef(x)
            if conv.fuses_input(x, coef):
                return conv(x, coef=coef, act=code, **kw)
            return conv(ops.affine_act(x, coef, code), **kw)
        return conv(norm(x, code), **kw)
