"""numpy restatement of the StyleGAN2 resampling / bias-activation ops (oracle; test infrastructure only).

Reference anchors:
  upfirdn2d        op/upfirdn2d.py:168-209 (upfirdn2d_native) == op/upfirdn2d_kernel.cu:49-105
  bias_act         op/fused_bias_act_kernel.cu:19-49  (act*10+grad switch)
  fused_leaky_relu op/fused_act.py:89-100
  setup_kernel / upsample_2d / downsample_2d   models/up_or_down_sampling.py:181-257
"""
import numpy as np


def upfirdn2d(x, k, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    """x (N, C, H, W) -> (N, C, out_h, out_w); zero-insert, pad/crop, correlate with the flipped
    kernel, decimate.  Accumulates in the input dtype's float type via float64 partial sums of
    kh*kw terms (order-insensitive to fp32 round-off at the test tolerances)."""
    x = np.asarray(x)
    k = np.asarray(k)
    N, C, H, W = x.shape
    kh, kw = k.shape
    up = np.zeros((N, C, H * up_y, W * up_x), dtype=np.float64)
    up[:, :, ::up_y, ::up_x] = x
    pad = np.pad(up, ((0, 0), (0, 0), (max(py0, 0), max(py1, 0)), (max(px0, 0), max(px1, 0))))
    pad = pad[:, :, max(-py0, 0):pad.shape[2] - max(-py1, 0), max(-px0, 0):pad.shape[3] - max(-px1, 0)]
    fh = pad.shape[2] - kh + 1
    fw = pad.shape[3] - kw + 1
    kf = k[::-1, ::-1].astype(np.float64)
    out = np.zeros((N, C, fh, fw), dtype=np.float64)
    for a in range(kh):
        for b in range(kw):
            out += kf[a, b] * pad[:, :, a:a + fh, b:b + fw]
    out = out[:, :, ::down_y, ::down_x]
    out_h = (H * up_y + py0 + py1 - kh) // down_y + 1
    out_w = (W * up_x + px0 + px1 - kw) // down_x + 1
    assert out.shape[2:] == (out_h, out_w), (out.shape, out_h, out_w)
    return out.astype(x.dtype)


def bias_act(x, b=None, ref=None, act=3, grad=0, alpha=0.2, scale=2 ** 0.5):
    """y = act(x + b[channel]) * scale with the kernel's (act, grad) table."""
    x = np.asarray(x)
    v = x.astype(np.float32)
    if b is not None and b.size:
        shape = [1] * x.ndim
        shape[1] = b.size
        v = v + np.asarray(b, dtype=np.float32).reshape(shape)
    code = act * 10 + grad
    if code == 30:
        y = np.where(v > 0, v, v * np.float32(alpha))
    elif code == 31:
        y = np.where(np.asarray(ref) > 0, v, v * np.float32(alpha))
    elif code in (12, 32):
        y = np.zeros_like(v)
    else:
        y = v
    return (y * np.float32(scale)).astype(x.dtype)


def fused_leaky_relu(x, b, negative_slope=0.2, scale=2 ** 0.5):
    return bias_act(x, b, None, 3, 0, negative_slope, scale)


def setup_kernel(k):
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    return k / k.sum()


def upsample_2d(x, k=None, factor=2, gain=1):
    k = setup_kernel([1] * factor if k is None else k) * (gain * factor ** 2)
    p = k.shape[0] - factor
    p0, p1 = (p + 1) // 2 + factor - 1, p // 2
    return upfirdn2d(x, k, factor, factor, 1, 1, p0, p1, p0, p1)


def downsample_2d(x, k=None, factor=2, gain=1):
    k = setup_kernel([1] * factor if k is None else k) * gain
    p = k.shape[0] - factor
    p0, p1 = (p + 1) // 2, p // 2
    return upfirdn2d(x, k, 1, 1, factor, factor, p0, p1, p0, p1)
