"""FIR up/down-sampling wrappers (mirror of the reference's ``models/up_or_down_sampling.py``: Conv2d :23-56,
naive_* :59-68, upsample_conv_2d :72-141, conv_downsample_2d :144-178, _setup_kernel :181-188, upsample_2d
:195-224, downsample_2d :227-257) on the gfx950 upfirdn2d kernel."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..op import upfirdn2d


def _setup_kernel(k):
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k /= np.sum(k)
    assert k.ndim == 2 and k.shape[0] == k.shape[1]
    return k


_KERNEL_CACHE = {}


def _fir(k, scale, device):
    key = (tuple(np.asarray(k, dtype=np.float32).ravel().tolist()), float(scale), str(device))
    if key not in _KERNEL_CACHE:
        _KERNEL_CACHE[key] = torch.tensor(_setup_kernel(k) * scale, device=device)
    return _KERNEL_CACHE[key]


_GAIN_CACHE = {}


def _max_phase_gain(k, scale, up):
    """max over the output phases of sum |tap| of the (separable or 2-D) FIR `k` normalised and scaled as _fir does, after zero
    insertion by `up`: |resampled x| <= this x max |x|.  1 for every filter the models use ((1,3,3,1), boxes): a bound on the
    input's maximum then bounds the output's, so the per-image maxima of the f16x2 dynamic range pass through the resampler."""
    key = (tuple(np.asarray(k, dtype=np.float32).ravel().tolist()), float(scale), int(up))
    if key not in _GAIN_CACHE:
        kk = np.abs(_setup_kernel(k) * scale)
        _GAIN_CACHE[key] = float(max(kk[i::up, j::up].sum() for i in range(up) for j in range(up)))
    return _GAIN_CACHE[key]


def _carry(x, y, gain):
    if gain <= 1.0 + 1e-6:
        ops.carry_amax(x, y)
    return y


def upsample_2d(x, k=None, factor=2, gain=1):
    """zero-insert by `factor` and low-pass with `k` (normalised, times gain * factor^2)"""
    assert isinstance(factor, int) and factor >= 1
    k = [1] * factor if k is None else k
    fir = _fir(k, gain * (factor ** 2), x.device)
    p = fir.shape[0] - factor
    y = upfirdn2d(x, fir, up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))
    return _carry(x, y, _max_phase_gain(k, gain * (factor ** 2), factor))


def downsample_2d(x, k=None, factor=2, gain=1):
    """low-pass with `k` (normalised, times gain) and keep every `factor`-th sample"""
    assert isinstance(factor, int) and factor >= 1
    k = [1] * factor if k is None else k
    fir = _fir(k, gain, x.device)
    p = fir.shape[0] - factor
    y = upfirdn2d(x, fir, down=factor, pad=((p + 1) // 2, p // 2))
    return _carry(x, y, _max_phase_gain(k, gain, 1))


def naive_upsample_2d(x, factor=2):
    """nearest-neighbour repeat == upfirdn2d with a box filter of ones"""
    return upsample_2d(x, [1] * factor, factor=factor)


def naive_downsample_2d(x, factor=2):
    """block mean == upfirdn2d with a normalised box filter"""
    if factor == 2:
        return ops.meanpool2(x)
    return downsample_2d(x, [1] * factor, factor=factor)


def upsample_conv_2d(x, w, k=None, factor=2, gain=1):
    raise NotImplementedError("upsample_conv_2d raises in the reference itself (negative-step slice, "
                              "models/up_or_down_sampling.py:126); no shipped config reaches it")


def conv_downsample_2d(x, w, k=None, factor=2, gain=1, packed=None, bias=None):
    """FIR low-pass (padding once, before both operations) followed by a stride-`factor` VALID convolution with `w`
    [Cout, Cin, k, k] (models/up_or_down_sampling.py:144-178).  The strided convolution runs as the stride-1 MFMA
    convolution sampled at the odd positions; `packed`: the pre-packed weight (ops.conv_weight(w))."""
    assert isinstance(factor, int) and factor == 2, "only the factor-2 form any model uses is built"
    _outC, _inC, convH, convW = w.shape
    assert convW == convH and convW % 2 == 1
    k = [1] * factor if k is None else k
    fir = _fir(k, gain, x.device)
    p = (fir.shape[0] - factor) + (convW - 1)
    xf = _carry(x, upfirdn2d(x, fir, pad=((p + 1) // 2, p // 2)), _max_phase_gain(k, gain, 1))
    impl = ops.impl_unbounded()
    return ops.conv2d_stride2_valid(xf, ops.conv_weight(w, impl) if packed is None else packed, bias, convW,
                                    in_amax=ops.in_amax_for(xf, impl, always=True))


class Conv2d(ops.PackedWeightMixin, nn.Module):
    """Conv2d layer with optional FIR up/down-sampling (StyleGAN2); plain stride-1 branch implemented."""

    def __init__(self, in_ch, out_ch, kernel, up=False, down=False, resample_kernel=(1, 3, 3, 1), use_bias=True,
                 kernel_init=None):
        super().__init__()
        assert not (up and down)
        assert kernel >= 1 and kernel % 2 == 1
        self.weight = nn.Parameter(torch.zeros(out_ch, in_ch, kernel, kernel))
        if kernel_init is not None:
            self.weight.data = kernel_init(self.weight.data.shape)
        if use_bias:
            self.bias = nn.Parameter(torch.zeros(out_ch))
        self.up, self.down, self.resample_kernel, self.kernel, self.use_bias = up, down, resample_kernel, kernel, use_bias
        self._cache = ops.PackedWeightCache()

    def forward(self, x):
        if self.up:
            return upsample_conv_2d(x, self.weight, k=self.resample_kernel)
        impl = ops.impl_unbounded()
        packed = self._cache.get(self.weight, "direct_" + impl, lambda w: ops.conv_weight(w, impl))
        if self.down:                   # the bias rides in the convolution epilogue (= x + bias.reshape(1, -1, 1, 1) afterwards)
            return conv_downsample_2d(x, self.weight, k=self.resample_kernel, packed=packed,
                                      bias=self.bias.data if self.use_bias else None)
        return ops.conv2d(x, packed, self.bias.data if self.use_bias else None, in_amax=ops.in_amax_for(x, impl, always=True))
