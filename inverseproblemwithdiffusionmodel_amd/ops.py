"""torch-tensor front end of the libipdm.so kernels: device memory, shapes and streams come from
PyTorch-ROCm (plumbing), all arithmetic happens in the hand-written HIP kernels (csrc/).

Every function requires GPU tensors and raises otherwise -- there is no CPU path.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import call, P

ACT_NONE, ACT_ELU, ACT_RELU, ACT_LRELU02, ACT_SWISH = 0, 1, 2, 3, 4
ACT_COPY = 5          # identity as an `act_out` code: "produce the second output, unactivated" (with res_second: conv + residual)
CONV_TRACE = None     # set to a list by engine.conv_census(): per-launch shapes + HIP events

ACT_CODES = {"none": ACT_NONE, None: ACT_NONE, "elu": ACT_ELU, "relu": ACT_RELU, "lrelu": ACT_LRELU02,
             "swish": ACT_SWISH}


def _gpu(t, dtype=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"ipdm: {name} must be a GPU tensor (the HIP path has no CPU fallback); got "
                           f"{getattr(t, 'device', type(t))}")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"ipdm: {name} must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return P(t.data_ptr()) if t is not None else P(0)


def _inplace_operand(t, dtype, name):
    """operand of an in-place / raw-pointer kernel: must already be a contiguous GPU tensor of the kernel's dtype
    (a silent .contiguous() copy would make the kernel update the copy; a wrong dtype would be read as raw memory)"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"ipdm: {name} must be a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"ipdm: {name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"ipdm: {name} must be contiguous (in-place kernel operand; call .contiguous() yourself and "
                         "keep the result)")
    return t


# One stream per device.  Two HSA queues on one card corrupt each other's packed-fp32 VALU results while certain MFMA kernels are
# resident (profiles/r03_shared_card_probe.txt; sharding.py), and kernels of ONE stream never overlap -- so the library refuses
# to be driven from a second eager stream of the same device.  Capture streams are exempt (torch.cuda.graph records on a side
# stream that never executes; a graph replays on the stream that launches it).  IPDM_ALLOW_MULTI_STREAM=1 lifts the check.
_STREAM_OF_DEVICE = {}


def _written(*tensors):
    """the kernels write through raw pointers, which torch's version counters never see: every op that stores into an EXISTING
    tensor reports it here, so that whatever was cached under (version, data_ptr) of that storage -- per-image maxima
    (`_ipdm_amax`), statistics partials (`_ipdm_partials`) -- goes stale for every alias (views share the counter).  Without this
    the sampler state, updated in place by the fused Langevin / proximal kernel, kept the maxima measured on its FIRST
    iteration: the segmentation network's first convolution then overflowed fp16 as the state grew (caught by
    test_guided_sense_trajectory_vs_oracle).  Host-side bookkeeping only; no launch."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def _stream():
    st = torch.cuda.current_stream()
    h = st.cuda_stream
    if _STREAM_OF_DEVICE.get(st.device_index, h) != h or st.device_index not in _STREAM_OF_DEVICE:
        _note_stream(st.device_index, h)
    return P(h)


def _note_stream(dev, h):
    if torch.cuda.is_current_stream_capturing():
        return
    first = _STREAM_OF_DEVICE.setdefault(dev, h)
    if first != h and os.environ.get("IPDM_ALLOW_MULTI_STREAM", "0") != "1":
        raise RuntimeError(
            f"ipdm: kernels were launched on stream {first:#x} of cuda:{dev} and now on stream {h:#x}: this library runs on ONE "
            "stream per device (two queues on one MI355X silently corrupt packed-fp32 results, "
            "profiles/r03_shared_card_probe.txt). Use one stream, or set IPDM_ALLOW_MULTI_STREAM=1 to take the risk.")


def _c64_as_f32(t):
    return torch.view_as_real(t)


# ---- StyleGAN2 ops -----------------------------------------------------------------------------
_FLOAT_SUFFIX = {torch.float32: "f32", torch.float16: "f16", torch.float64: "f64"}   # the reference's dispatch types


def _float_suffix(t, what):
    try:
        return _FLOAT_SUFFIX[t.dtype]
    except KeyError:
        raise TypeError(f"ipdm {what}: dtype {t.dtype} (float32 / float16 / float64, as the reference's "
                        "AT_DISPATCH_FLOATING_TYPES_AND_HALF)") from None


def upfirdn2d_raw(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """x [major, in_h, in_w, minor] -> [major, out_h, out_w, minor] (the reference extension's signature);
    float32 (the tuned kernels), float16 or float64 storage; the taps are cast to the input's dtype"""
    sfx = _float_suffix(x, "upfirdn2d")
    x = _gpu(x, x.dtype, "input")
    kernel = _gpu(kernel.to(x.dtype), x.dtype, "kernel")
    major, in_h, in_w, minor = x.shape
    kh, kw = kernel.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    out = torch.empty((major, out_h, out_w, minor), dtype=x.dtype, device=x.device)
    call(f"ipdm_upfirdn2d_{sfx}", _ptr(x), _ptr(kernel), _ptr(out), major, in_h, in_w, minor, kh, kw,
         up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, _stream())
    return out


def fused_bias_act_raw(x, bias, ref, act, grad, alpha, scale):
    sfx = _float_suffix(x, "fused_bias_act")
    x = _gpu(x, x.dtype, "input")
    bias = None if bias is None or bias.numel() == 0 else _gpu(bias.to(x.dtype), x.dtype, "bias")
    ref = None if ref is None or ref.numel() == 0 else _gpu(ref, x.dtype, "refer")
    step_b = 1
    for s in x.shape[2:]:
        step_b *= s
    y = torch.empty_like(x)
    call(f"ipdm_fused_bias_act_{sfx}", _ptr(x), _ptr(bias), _ptr(ref), _ptr(y), x.numel(), step_b,
         0 if bias is None else bias.numel(), act, grad, float(alpha), float(scale), _stream())
    return y


# ---- k-space --------------------------------------------------------------------------------
def fft2c(x, inverse=False):
    x = _gpu(x, name="input")
    if x.dtype != torch.complex64:
        x = x.to(torch.complex64)
    H, W = x.shape[-2:]
    batch = x.numel() // (H * W) if H * W else 0
    out = torch.empty_like(x)
    ws_bytes = _lib.lib.ipdm_fft2c_workspace_bytes(batch, H, W)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device) if ws_bytes else None
    call("ipdm_fft2c_c64", _ptr(x), _ptr(out), batch, H, W, int(bool(inverse)), _ptr(ws), _stream())
    return out


def _mask_u8(mask, W, device):
    """bool/any mask broadcastable over (..., W) -> uint8 [mask_t, W]."""
    m = mask.to(device=device)
    if m.shape[-1] != W:
        raise ValueError(f"mask last dim {m.shape[-1]} != W {W}")
    m = m.reshape(-1, W)
    return (m != 0).to(torch.uint8).contiguous()


def sense_forward(x, sens_f32, mask_u8):
    """sens_f32 None: single coil (S = 1), returns y with a leading axis of length 1"""
    x = _gpu(x, torch.complex64, "x")
    B = x.numel() // (x.shape[-1] * x.shape[-2])
    H, W = x.shape[-2:]
    n = 1 if sens_f32 is None else sens_f32.shape[0]
    y = torch.empty((n,) + tuple(x.shape), dtype=torch.complex64, device=x.device)
    call("ipdm_sense_forward_c64", _ptr(x), _ptr(sens_f32), _ptr(mask_u8), mask_u8.shape[0], _ptr(y), B, n, H, W,
         _stream())
    return y


def sense_workspace(B, n_coils, H, W, device):
    """scratch tensor for the SENSE / single-coil operators at this size (None when the kernels need none)"""
    nbytes = _lib.lib.ipdm_sense_workspace_bytes(B, n_coils, H, W)
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device) if nbytes else None


def _check_work(work, B, n_coils, H, W, what):
    """the SENSE kernels write `work` as n_coils * B complex images: a short / foreign buffer is an out-of-bounds GPU write"""
    need = _lib.lib.ipdm_sense_workspace_bytes(B, n_coils, H, W)
    if need == 0:
        return
    if (not isinstance(work, torch.Tensor) or not work.is_cuda or work.dtype != torch.float32 or not work.is_contiguous()
            or work.numel() * 4 < need):
        raise ValueError(f"{what}: `work` must be a contiguous float32 GPU tensor of >= {need} bytes "
                         f"(ops.sense_workspace({B}, {n_coils}, {H}, {W}, device))")


def _large_image(H, W):
    return H * W > 16384


def sense_adjoint(s, sens_f32, mask_u8=None, apply_mask=False):
    s = _gpu(s, torch.complex64, "s")
    n = s.shape[0]
    H, W = s.shape[-2:]
    B = s[0].numel() // (H * W)
    x = torch.empty(tuple(s.shape[1:]), dtype=torch.complex64, device=s.device)
    ws = sense_workspace(B, n, H, W, s.device) if _large_image(H, W) else None
    call("ipdm_sense_adjoint_c64", _ptr(s), _ptr(sens_f32), _ptr(mask_u8), 1 if mask_u8 is None else mask_u8.shape[0],
         int(bool(apply_mask)), _ptr(x), _ptr(ws), B, n, H, W, _stream())
    return x


def sense_ssos(s):
    s = _gpu(s, torch.complex64, "s")
    n = s.shape[0]
    H, W = s.shape[-2:]
    B = s[0].numel() // (H * W)
    out = torch.empty(tuple(s.shape[1:]), dtype=torch.float32, device=s.device)
    ws = sense_workspace(B, n, H, W, s.device) if _large_image(H, W) else None
    call("ipdm_sense_ssos_c64", _ptr(s), _ptr(out), _ptr(ws), B, n, H, W, _stream())
    return out


def sense_l2prox(z_re, z_im, y, sens_f32, mask_u8, coef, out_re=None, out_im=None, work=None):
    z_re, z_im = _gpu(z_re, torch.float32, "z_re"), _gpu(z_im, torch.float32, "z_im")
    y = _gpu(y, torch.complex64, "y")
    H, W = z_re.shape[-2:]
    B = z_re.numel() // (H * W)
    out_re = torch.empty_like(z_re) if out_re is None else out_re
    out_im = torch.empty_like(z_im) if out_im is None else out_im
    work = sense_workspace(B, sens_f32.shape[0], H, W, z_re.device) if work is None else work
    _check_work(work, B, sens_f32.shape[0], H, W, "sense_l2prox")
    call("ipdm_sense_l2prox_f32", _ptr(z_re), _ptr(z_im), _ptr(y), _ptr(sens_f32), _ptr(mask_u8), mask_u8.shape[0],
         float(coef), _ptr(out_re), _ptr(out_im), _ptr(work), B, sens_f32.shape[0], H, W, _stream())
    _written(out_re, out_im)
    return out_re, out_im


def ald_sense_step(x_re, x_im, g_re, g_im, y, sens_f32, mask_u8, work, step=0.0, noise_scale=0.0, coef=0.0,
                   noise_re=None, noise_im=None, seed=0, sample_offset=0, step_id=0, dev_sched=None):
    """in place on x_re / x_im.  dev_sched: uint8/any device tensor holding an ipdm_sched_t."""
    for t, n in ((x_re, "x_re"), (x_im, "x_im"), (g_re, "g_re"), (g_im, "g_im"), (sens_f32, "sens")):
        _inplace_operand(t, torch.float32, n)
    _inplace_operand(y, torch.complex64, "y")
    for t, n in ((noise_re, "noise_re"), (noise_im, "noise_im")):
        if t is not None:
            _inplace_operand(t, torch.float32, n)
    H, W = x_re.shape[-2:]
    B = x_re.numel() // (H * W)
    if g_re.numel() != x_re.numel() or g_im.numel() != x_im.numel() or y.numel() != sens_f32.shape[0] * B * H * W:
        raise ValueError("ald_sense_step: operand sizes do not match the state")
    _check_work(work, B, sens_f32.shape[0], H, W, "ald_sense_step")
    call("ipdm_ald_sense_step_f32", _ptr(x_re), _ptr(x_im), _ptr(g_re), _ptr(g_im), _ptr(noise_re), _ptr(noise_im),
         float(step), float(noise_scale), int(seed), int(sample_offset), int(step_id), _ptr(dev_sched),
         _ptr(y), _ptr(sens_f32), _ptr(mask_u8), mask_u8.shape[0], float(coef), _ptr(work), B, sens_f32.shape[0], H, W,
         _stream())
    _written(x_re, x_im)


SC_L2PENALTY, SC_CLOSED_FORM, SC_PROJECTION = 0, 1, 2


def singlecoil_prox(z_re, z_im, y, mask_u8, coef, mode, out_re=None, out_im=None, work=None):
    """single-coil data consistency on planar real / imaginary planes (modes: ipdm.h, ipdm_singlecoil_prox_f32)"""
    z_re, z_im = _gpu(z_re, torch.float32, "z_re"), _gpu(z_im, torch.float32, "z_im")
    y = _gpu(y, torch.complex64, "y")
    H, W = z_re.shape[-2:]
    B = z_re.numel() // (H * W)
    if y.numel() != B * H * W:
        raise ValueError(f"singlecoil_prox: measurement {tuple(y.shape)} does not match the image batch {tuple(z_re.shape)}")
    out_re = torch.empty_like(z_re) if out_re is None else out_re
    out_im = torch.empty_like(z_im) if out_im is None else out_im
    if work is None and _large_image(H, W):
        work = sense_workspace(B, 1, H, W, z_re.device)
    if work is not None or _large_image(H, W):
        _check_work(work, B, 1, H, W, "singlecoil_prox")
    call("ipdm_singlecoil_prox_f32", _ptr(z_re), _ptr(z_im), _ptr(y), _ptr(mask_u8), mask_u8.shape[0], float(coef),
         int(mode), _ptr(out_re), _ptr(out_im), _ptr(work), B, H, W, _stream())
    _written(out_re, out_im)
    return out_re, out_im


def ald_singlecoil_step(x_re, x_im, g_re, g_im, y, mask_u8, mode, step=0.0, noise_scale=0.0, coef=0.0, noise_re=None,
                        noise_im=None, seed=0, sample_offset=0, step_id=0, dev_sched=None, work=None):
    """in place on x_re / x_im (contiguous float32 GPU planes)"""
    for t, n in ((x_re, "x_re"), (x_im, "x_im"), (g_re, "g_re"), (g_im, "g_im")):
        _inplace_operand(t, torch.float32, n)
    _inplace_operand(y, torch.complex64, "y")
    H, W = x_re.shape[-2:]
    B = x_re.numel() // (H * W)
    if work is None and _large_image(H, W):
        work = sense_workspace(B, 1, H, W, x_re.device)
    if work is not None or _large_image(H, W):
        _check_work(work, B, 1, H, W, "ald_singlecoil_step")
    call("ipdm_ald_singlecoil_step_f32", _ptr(x_re), _ptr(x_im), _ptr(g_re), _ptr(g_im), _ptr(noise_re), _ptr(noise_im),
         float(step), float(noise_scale), int(seed), int(sample_offset), int(step_id), _ptr(dev_sched), _ptr(y),
         _ptr(mask_u8), mask_u8.shape[0], float(coef), int(mode), _ptr(work), B, H, W, _stream())
    _written(x_re, x_im)


def langevin_step(x, g, step=0.0, noise_scale=0.0, noise=None, seed=0, sample_offset=0, step_id=0, dev_sched=None):
    """x += step*g + noise_scale*noise in place; x [n_samples, ...]."""
    _inplace_operand(x, torch.float32, "x")
    _inplace_operand(g, torch.float32, "g")
    if noise is not None:
        _inplace_operand(noise, torch.float32, "noise")
        if noise.numel() != x.numel():
            raise ValueError("langevin_step: noise does not match x")
    if g.numel() != x.numel():
        raise ValueError("langevin_step: g does not match x")
    n_samples = x.shape[0]
    call("ipdm_langevin_step_f32", _ptr(x), _ptr(g), _ptr(noise), float(step), float(noise_scale), int(seed),
         int(sample_offset), int(step_id), _ptr(dev_sched), n_samples, x.numel() // max(n_samples, 1), _stream())
    _written(x)
    return x


def philox_normal(shape, device, seed=0, sample_offset=0, step_id=0, plane=0):
    out = torch.empty(shape, dtype=torch.float32, device=device)
    n_samples = shape[0]
    call("ipdm_philox_normal_f32", _ptr(out), int(seed), int(sample_offset), int(step_id), int(plane), n_samples,
         out.numel() // max(n_samples, 1), _stream())
    return out


# ---- segmentation-likelihood guidance glue ----------------------------------------------------------
def zero_insert2(x):
    """(..., H, W) -> (..., 2H, 2W): x at the even positions, zeros elsewhere"""
    x = _gpu(x, torch.float32, "x")
    H, W = x.shape[-2:]
    out = torch.empty(tuple(x.shape[:-2]) + (2 * H, 2 * W), dtype=torch.float32, device=x.device)
    call("ipdm_zero_insert2_f32", _ptr(x), _ptr(out), x.numel() // (H * W), H, W, _stream())
    return out


def subsample2(x, offset=(0, 0), size=None):
    """(..., H, W) -> (..., Ho, Wo): x[..., 2i + oy, 2j + ox]; default: the even positions, (H/2, W/2)"""
    x = _gpu(x, torch.float32, "x")
    H, W = x.shape[-2:]
    oy, ox = offset
    Ho, Wo = ((H - oy + 1) // 2, (W - ox + 1) // 2) if size is None else size
    out = torch.empty(tuple(x.shape[:-2]) + (Ho, Wo), dtype=torch.float32, device=x.device)
    call("ipdm_subsample2_f32", _ptr(x), _ptr(out), x.numel() // (H * W), H, W, oy, ox, Ho, Wo, _stream())
    return out


def conv2d_stride2_valid(x, wt, bias=None, ksize=3, in_amax=None):
    """F.conv2d(x, w, stride=2, padding=0) for an odd kernel on the stride-1 MFMA kernel: the 'same' convolution sampled at
    (k//2, k//2) + 2(i, j).  wt: the packed weight of the same layer (conv_weight)."""
    H, W = x.shape[-2:]
    full = conv2d(x, wt, bias, in_amax=in_amax)
    o = ksize // 2
    return subsample2(full, (o, o), ((H - ksize) // 2 + 1, (W - ksize) // 2 + 1))


def in_prelu_fwd(x, slope, eps=1e-5):
    """InstanceNorm (no affine) + PReLU(one slope) on (B, C, H, W) -> (xhat, y, rstd (B*C,))"""
    x = _gpu(x, torch.float32, "x")
    B, C = x.shape[:2]
    hw = x.numel() // max(B * C, 1)
    xhat, y = torch.empty_like(x), torch.empty_like(x)
    rstd = torch.empty(B * C, dtype=torch.float32, device=x.device)
    call("ipdm_in_prelu_fwd_f32", _ptr(x), _ptr(slope), _ptr(xhat), _ptr(y), _ptr(rstd), B * C, hw, float(eps), _stream())
    return xhat, y, rstd


def in_prelu_bwd(gy, xhat, rstd, slope):
    gy = _gpu(gy, torch.float32, "gy")
    B, C = gy.shape[:2]
    gx = torch.empty_like(gy)
    call("ipdm_in_prelu_bwd_f32", _ptr(gy), _ptr(xhat), _ptr(rstd), _ptr(slope), _ptr(gx), B * C, gy.numel() // max(B * C, 1),
         _stream())
    return gx


def seg_loglh_grad(logits, label):
    """d/dlogits of sum log softmax(logits, dim=1)[label]: logits (B, C, H, W) f32, label (B, 1, H, W) int64"""
    logits = _gpu(logits, torch.float32, "logits")
    label = _gpu(label, torch.int64, "label")
    B, C = logits.shape[:2]
    hw = logits.numel() // max(B * C, 1)
    if label.numel() != B * hw:
        raise ValueError(f"seg_loglh_grad: label {tuple(label.shape)} does not match logits {tuple(logits.shape)}")
    g = torch.empty_like(logits)
    call("ipdm_seg_loglh_grad_f32", _ptr(logits), _ptr(label), _ptr(g), B, C, hw, _stream())
    return g


def axpy_sched(y, x, scale=0.0, dev_sched=None, mask=None):
    """y += scale * x (* mask, broadcast over y's leading copies) in place; dev_sched: the device ipdm_sched_t whose
    seg_scale replaces `scale` (graph replay)"""
    _inplace_operand(y, torch.float32, "y")
    _inplace_operand(x, torch.float32, "x")
    if x.numel() != y.numel():
        raise ValueError("axpy_sched: x does not match y")
    period = 0
    if mask is not None:
        _inplace_operand(mask, torch.int64, "mask")
        period = mask.numel()
        if y.numel() % period:
            raise ValueError("axpy_sched: mask does not tile y")
    call("ipdm_axpy_sched_f32", _ptr(y), _ptr(x), _ptr(mask), period, _ptr(dev_sched), float(scale), y.numel(), _stream())
    _written(y)
    return y


# ---- on-device reporting ------------------------------------------------------------------------
def magnitude(x):
    x = _gpu(x, torch.complex64, "x")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    call("ipdm_magnitude_c64", _ptr(x), _ptr(out), x.numel(), _stream())
    return out


def posterior_moment_planes(samples):
    """samples (n, ..., H, W) complex64 -> (7, ..., H, W) float64 partial sums over the n samples:
    sum |x|, sum |x|^2, sum angle, sum angle^2, sum Re, sum Im, sum |angle|"""
    samples = _gpu(samples, torch.complex64, "samples")
    n = samples.shape[0]
    hw = samples[0].numel() if n else 0
    planes = torch.empty((7,) + tuple(samples.shape[1:]), dtype=torch.float64, device=samples.device)
    call("ipdm_posterior_moments_c64", _ptr(samples), _ptr(planes), n, hw, _stream())
    return planes


def tv_value(x):
    """x (n, ..., H, W) complex64 -> (n',) float64 total variation per image (n' = all leading dims flattened)"""
    x = _gpu(x, torch.complex64, "x")
    H, W = x.shape[-2:]
    n = x.numel() // (H * W)
    out = torch.empty(n, dtype=torch.float64, device=x.device)
    call("ipdm_tv_c64", _ptr(x), _ptr(out), n, H, W, _stream())
    return out


def tv_grad(x):
    """gradient of sum-of-moduli total variation w.r.t. the complex image (torch autograd's convention)"""
    x = _gpu(x, torch.complex64, "x")
    H, W = x.shape[-2:]
    g = torch.empty_like(x)
    call("ipdm_tv_grad_c64", _ptr(x), _ptr(g), x.numel() // (H * W), H, W, _stream())
    return g


def nrmse(img, ref):
    """per-image ||img - ref|| / ||img|| (the reference's argument order); img (n, ...), ref same shape or one image"""
    img, ref = _gpu(img, torch.float32, "img"), _gpu(ref, torch.float32, "ref")
    n = img.shape[0]
    elems = img[0].numel()
    bcast = ref.numel() == elems
    if not bcast and ref.numel() != img.numel():
        raise ValueError(f"nrmse: reference {tuple(ref.shape)} does not match {tuple(img.shape)}")
    out = torch.empty(n, dtype=torch.float64, device=img.device)
    call("ipdm_nrmse_f32", _ptr(img), _ptr(ref), _ptr(out), n, elems, int(bcast), _stream())
    return out


def ssim(img, ref, data_range=2.0):
    """per-image mean SSIM of single-channel images (n, H, W) (or (n, 1, H, W)) against ref (same shape or one image)"""
    img, ref = _gpu(img, torch.float32, "img"), _gpu(ref, torch.float32, "ref")
    H, W = img.shape[-2:]
    n = img.numel() // (H * W)
    bcast = ref.numel() == H * W
    if not bcast and ref.numel() != img.numel():
        raise ValueError(f"ssim: reference {tuple(ref.shape)} does not match {tuple(img.shape)}")
    out = torch.empty(n, dtype=torch.float64, device=img.device)
    call("ipdm_ssim_f32", _ptr(img), _ptr(ref), _ptr(out), n, H, W, int(bcast), float(data_range), _stream())
    return out


# ---- per-image activation maxima: the f16x2 family's DYNAMIC RANGE without a pass over the tensor ------------------
# An f16x2 convolution scales every image of its input into fp16's range by an exact power of two derived from an upper bound of
# that image's max |x| (`in_amax`, conv_kernel.h hx_dynamic_scale) -- then ANY fp32 input is in range and values down to 2^-17 of
# the bound keep all 22 significand bits.  The bounds come from the PRODUCERS: convolution epilogues and the resize kernels
# accumulate the exact maxima of what they store with one atomic max per wave (`want_amax`), InstanceNorm++ / GroupNorm hand over
# a bound computed from their coefficients, and pooling / activation / gather kernels pass their input's bound on (they never
# increase |x|).  Maxima ride on the tensors as `_ipdm_amax = (amax [B], (version, data_ptr))`: a tensor written in place since
# loses them, and a consumer that finds none MEASURES its input (ipdm_absmax_f32: one extra pass, counted in AMAX_MEASURED so that
# tests can assert the hot path never does).  IPDM_HX2_DYNAMIC=0 restores round 3's static contract (|x| < 65504) for A/B runs.
HX2_DYNAMIC = os.environ.get("IPDM_HX2_DYNAMIC", "1") != "0"
AMAX_MEASURED = 0                               # fallback absmax passes since import (diagnostics / tests)


def dynamic_range(impl=None):
    """True when convolutions should produce and consume per-image maxima (the f16x2 family with the dynamic range on);
    impl: the kernel family in use where it is not CONV_IMPL (the score_sde networks: impl_unbounded())"""
    return HX2_DYNAMIC and (CONV_IMPL if impl is None else impl) == "hx2"


AMAX_WAYS, AMAX_SLOT = 8, 128                    # include/ipdm.h "maxima vectors": [B][AMAX_SLOT] floats, 8 ways per image, 16 floats apart


def amax_value(am):
    """maxima vector [B, AMAX_SLOT] -> per-image values [B] (the max over an image's ways)"""
    return am.view(am.shape[0], AMAX_WAYS, AMAX_SLOT // AMAX_WAYS)[:, :, 0].amax(dim=1)


class _AmaxArena:
    """zeroed [slots][B][AMAX_SLOT] float32 block: ONE fill per block instead of one per vector (atomic-max ways start at zero)"""

    def __init__(self, B, device, slots):
        self.B, self.device, self.used, self.slots = B, device, 0, slots
        self.buf = torch.zeros((slots, B, AMAX_SLOT), dtype=torch.float32, device=device)

    def take(self):
        if self.used == self.slots:
            return None
        self.used += 1
        return self.buf[self.used - 1]


_ARENAS = []                                    # stack of active amax_scope()s


class amax_scope:
    """`with ops.amax_scope():` around one network evaluation: the zeroed slots of every producer inside come out of a few large
    blocks (one fill kernel per 256 slots).  Allocated INSIDE the scope, so a forward captured into a hipGraph re-zeroes its
    slots on every replay.  Without a scope every slot is its own torch.zeros (correct, one more launch each)."""

    def __enter__(self):
        _ARENAS.append({})
        return self

    def __exit__(self, *exc):
        _ARENAS.pop()
        return False


def amax_slot(B, device):
    """a zeroed maxima vector (float32 [B, AMAX_SLOT]) for a producer's atomic maxima"""
    if _ARENAS:
        blocks = _ARENAS[-1]
        key = (int(B), str(device))
        a = blocks.get(key)
        slot = a.take() if a is not None else None
        if slot is None:
            blocks[key] = a = _AmaxArena(int(B), device, 128)
            slot = a.take()
        return slot
    return torch.zeros((int(B), AMAX_SLOT), dtype=torch.float32, device=device)


def tag_amax(t, amax):
    """attach per-image maxima (or an upper bound of them) to tensor t as it is NOW"""
    if t is not None and amax is not None:
        t._ipdm_amax = (amax, (t._version, t.data_ptr()))
    return t


def amax_of(t):
    """the maxima attached to t, or None (never attached, or t was written in place since)"""
    hit = getattr(t, "_ipdm_amax", None)
    if hit is None:
        return None
    amax, tag = hit
    if tag != (t._version, t.data_ptr()) or amax.shape[0] != t.shape[0]:
        return None
    return amax


def carry_amax(src, dst):
    """dst = f(src) with |f(x)| <= max |src| everywhere (pooling, activations, gathers, views): src's bound bounds dst"""
    return tag_amax(dst, amax_of(src))


def in_amax_for(x, impl=None, always=False):
    """the `in_amax` argument for a convolution reading x: None (static contract / other kernel families), the maxima attached to
    x, or -- where nothing is attached -- a MEASUREMENT (ipdm_absmax_f32: one pass, counted in AMAX_MEASURED), which is then
    attached to x so that its other consumers find it.  always: dynamic whenever the family is f16x2 (the score_sde networks,
    whose raw streams overflow the static contract: HX2_DYNAMIC does not apply to them)"""
    fam = CONV_IMPL if impl is None else impl
    if fam != "hx2" or not (always or HX2_DYNAMIC) or os.environ.get("IPDM_AMAX_CONSUME", "1") == "0":
        return None
    am = amax_of(x)
    if am is not None:
        return am
    global AMAX_MEASURED
    AMAX_MEASURED += 1
    am = absmax_per_image(x)
    tag_amax(x, am)
    return am


# ---- score-network glue -----------------------------------------------------------------------
# the producing convolution's epilogue hands the plane statistics over as partials (conv2d_wino_bx3(want_stats=True) hangs
# them on its result as `_ipdm_partials`), so that InstanceNorm++ does not read the tensor a second time
USE_STATS_EPILOGUE = os.environ.get("IPDM_STATS_EPILOGUE", "1") != "0"


def instnorm_plus_coef(x, alpha, gamma, beta):
    """x [B, C, *spatial] (2-D images or 3-D volumes: the statistics run over all spatial positions)"""
    x = _gpu(x, torch.float32, "x")
    B, C = x.shape[:2]
    coef = torch.empty((B, C, 3), dtype=torch.float32, device=x.device)
    hw = x.numel() // max(B * C, 1)
    # dynamic range: an upper bound of max |normalised value| per image, from the coefficients (affine_act passes it on)
    bound = torch.empty((B, AMAX_SLOT), dtype=torch.float32, device=x.device) if dynamic_range() else None
    if bound is not None:
        coef._ipdm_amax_bound = bound
    part = getattr(x, "_ipdm_partials", None)
    if part is not None:
        del x._ipdm_partials             # single use: whatever touches the tensor afterwards cannot meet stale statistics
        part, tag = part
        if tag != (x._version, x.data_ptr()):        # written in place since the convolution produced it: read the tensor
            part = None
    if part is not None and USE_STATS_EPILOGUE and tuple(part.shape[:2]) == (B, C):
        call("ipdm_instnorm_plus_coef_partials_f32", _ptr(part), int(part.shape[2]), _ptr(alpha), _ptr(gamma), _ptr(beta),
             _ptr(coef), B, C, hw, _ptr(bound), _stream())
        return coef
    call("ipdm_instnorm_plus_coef_f32", _ptr(x), _ptr(alpha), _ptr(gamma), _ptr(beta), _ptr(coef), B, C, hw, _ptr(bound),
         _stream())
    return coef


def affine_act(x, coef, act=ACT_NONE, out=None):
    x = _gpu(x, torch.float32, "x")
    B, C = x.shape[:2]
    out = torch.empty_like(x) if out is None else out
    call("ipdm_affine_act_f32", _ptr(x), _ptr(coef), _ptr(out), B, C, x.numel() // max(B * C, 1), act, _stream())
    _written(out)
    return tag_amax(out, getattr(coef, "_ipdm_amax_bound", None))     # (every activation code shrinks |.|)


def act(x, code, out=None):
    x = _gpu(x, torch.float32, "x")
    out = torch.empty_like(x) if out is None else out
    call("ipdm_act_f32", _ptr(x), _ptr(out), x.numel(), code, _stream())
    am = amax_of(x)                                 # (every activation code shrinks |.|: the input's bound holds, in place too)
    _written(out)
    return tag_amax(out, am)


def scale_shift(x, a, b, out=None):
    x = _gpu(x, torch.float32, "x")
    out = torch.empty_like(x) if out is None else out
    call("ipdm_scale_shift_f32", _ptr(x), _ptr(out), x.numel(), float(a), float(b), _stream())
    _written(out)
    return out


def add(x, y, out=None):
    x, y = _gpu(x, torch.float32, "x"), _gpu(y, torch.float32, "y")
    if x.shape != y.shape:
        raise ValueError(f"ipdm add: shapes differ {tuple(x.shape)} vs {tuple(y.shape)}")
    out = torch.empty_like(x) if out is None else out
    call("ipdm_add_f32", _ptr(x), _ptr(y), _ptr(out), x.numel(), _stream())
    _written(out)
    return out


def div_sigma(x, sigmas, labels=None, out=None):
    """x[b] / sigmas[labels[b]]  (labels None: x[b] / sigmas[b])"""
    x = _gpu(x, torch.float32, "x")
    labels = None if labels is None else _gpu(labels, torch.int64, "labels")
    sigmas = _gpu(sigmas, torch.float32, "sigmas")
    out = torch.empty_like(x) if out is None else out
    B = x.shape[0]
    call("ipdm_div_sigma_f32", _ptr(x), _ptr(sigmas), _ptr(labels), _ptr(out), B, x.numel() // max(B, 1), _stream())
    _written(out)
    return out


def maxpool5(x):
    x = _gpu(x, torch.float32, "x")
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    call("ipdm_maxpool5_f32", _ptr(x), _ptr(out), B * C, H, W, _stream())
    return carry_amax(x, out)


def meanpool2(x):
    x = _gpu(x, torch.float32, "x")
    B, C, H, W = x.shape
    out = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    call("ipdm_meanpool2_f32", _ptr(x), _ptr(out), B * C, H, W, _stream())
    return carry_amax(x, out)


def bilinear(x, size, out=None, accumulate=False, act=ACT_NONE, want_amax=False):
    """want_amax: the per-image maxima of what is written ride on the result (tag_amax)"""
    x = _gpu(x, torch.float32, "x")
    B, C, H, W = x.shape
    oh, ow = int(size[0]), int(size[1])
    if out is None:
        out = torch.empty((B, C, oh, ow), dtype=torch.float32, device=x.device)
        accumulate = False
    slot = amax_slot(B, x.device) if want_amax and B <= 65535 else None
    call("ipdm_bilinear_f32", _ptr(x), _ptr(out), B * C, H, W, oh, ow, int(bool(accumulate)), act, C, _ptr(slot), _stream())
    _written(out)
    return tag_amax(out, slot)


def trilinear(x, size, out=None, accumulate=False, act=ACT_NONE, want_amax=False):
    """F.interpolate(x, size, mode='trilinear', align_corners=True) of (B, C, D, H, W), optionally accumulated into out"""
    x = _gpu(x, torch.float32, "x")
    B, C, D, H, W = x.shape
    od, oh, ow = (int(v) for v in size)
    if out is None:
        out = torch.empty((B, C, od, oh, ow), dtype=torch.float32, device=x.device)
        accumulate = False
    slot = amax_slot(B, x.device) if want_amax and B <= 65535 else None
    call("ipdm_trilinear_f32", _ptr(x), _ptr(out), B * C, D, H, W, od, oh, ow, int(bool(accumulate)), act, C, _ptr(slot),
         _stream())
    _written(out)
    return tag_amax(out, slot)


# ---- NCSN++ / predictor-corrector extras --------------------------------------------------------
def stats_partials_of(x):
    """the statistics partials [B, C, P, 3] the producing convolution's epilogue hung on x (conv2d_wino_bx3(want_stats=True)), or
    None -- also None when x was written since (the tag is its (version, data_ptr))"""
    part = getattr(x, "_ipdm_partials", None)
    if part is None or not USE_STATS_EPILOGUE:
        return None
    part, tag = part
    if tag != (x._version, x.data_ptr()) or tuple(part.shape[:2]) != tuple(x.shape[:2]):
        return None
    return part


def groupnorm_coef(x, weight, bias, groups, eps=1e-6, want_amax=False):
    """-> coef [B, C, 3]; want_amax: -> (coef, per-image max |x| [B]) -- the maxima come out of the statistics pass (planes in
    registers) plus one tiny reduction; where that pass is not the single-read kernel the input is measured separately.
    Where x carries its producer's statistics partials (stats_partials_of) the coefficients come from those: x is not read."""
    x = _gpu(x, torch.float32, "x")
    B, C, H, W = x.shape
    coef = torch.empty((B, C, 3), dtype=torch.float32, device=x.device)
    part = None if want_amax else stats_partials_of(x)
    if part is not None:
        call("ipdm_groupnorm_coef_partials_f32", _ptr(part), C, _ptr(None), 0, int(part.shape[2]), _ptr(weight), _ptr(bias),
             _ptr(coef), B, groups, float(eps), _stream())
        return coef
    if want_amax:
        planes = torch.empty((B, C), dtype=torch.float32, device=x.device)
        try:
            call("ipdm_groupnorm_coef_f32", _ptr(x), _ptr(weight), _ptr(bias), _ptr(coef), B, C, H * W, groups, float(eps),
                 _ptr(planes), _stream())
            return coef, absmax_per_image(planes)
        except _lib.IpdmUnsupported:
            pass
    call("ipdm_groupnorm_coef_f32", _ptr(x), _ptr(weight), _ptr(bias), _ptr(coef), B, C, H * W, groups, float(eps),
         _ptr(None), _stream())
    return (coef, absmax_per_image(x)) if want_amax else coef


def groupnorm_act_cat(x1, x2, weight, bias, groups, eps=1e-6, act=ACT_NONE, want_amax=False):
    """act(GroupNorm(torch.cat([x1, x2], dim=1))) without materialising the concatenation of the RAW tensors; falls back to
    the concatenation where the two-source kernels do not apply.  want_amax: -> (out, per-image max over BOTH tensors [B])"""
    x1, x2 = _gpu(x1, torch.float32, "x1"), _gpu(x2, torch.float32, "x2")
    B, C1, H, W = x1.shape
    C2 = x2.shape[1]
    if tuple(x2.shape) != (B, C2, H, W):
        raise ValueError(f"groupnorm_act_cat: {tuple(x1.shape)} vs {tuple(x2.shape)}")
    coef = torch.empty((B, C1 + C2, 3), dtype=torch.float32, device=x1.device)
    out = torch.empty((B, C1 + C2, H, W), dtype=torch.float32, device=x1.device)
    planes = torch.empty((B, C1 + C2), dtype=torch.float32, device=x1.device) if want_amax else None
    p1, p2 = (None, None) if want_amax else (stats_partials_of(x1), stats_partials_of(x2))
    if p1 is not None and p2 is not None and p1.shape[2] == p2.shape[2]:       # both producers left their statistics: no read
        try:
            call("ipdm_groupnorm_coef_partials_f32", _ptr(p1), C1, _ptr(p2), C2, int(p1.shape[2]), _ptr(weight), _ptr(bias),
                 _ptr(coef), B, groups, float(eps), _stream())
            call("ipdm_affine_act_cat_f32", _ptr(x1), C1, _ptr(x2), C2, _ptr(coef), _ptr(out), B, H * W, act, _stream())
            return out
        except _lib.IpdmUnsupported:
            pass
    try:
        call("ipdm_groupnorm_coef_cat_f32", _ptr(x1), C1, _ptr(x2), C2, _ptr(weight), _ptr(bias), _ptr(coef), B, H * W, groups,
             float(eps), _ptr(planes), _stream())
        call("ipdm_affine_act_cat_f32", _ptr(x1), C1, _ptr(x2), C2, _ptr(coef), _ptr(out), B, H * W, act, _stream())
        return (out, absmax_per_image(planes)) if want_amax else out
    except _lib.IpdmUnsupported:
        x = torch.cat([x1, x2], dim=1)
        if want_amax:
            c, am = groupnorm_coef(x, weight, bias, groups, eps, want_amax=True)
            return affine_act(x, c, act), am
        return affine_act(x, groupnorm_coef(x, weight, bias, groups, eps), act)


def linear(x, weight, bias=None, act_in=ACT_NONE):
    x = _gpu(x, torch.float32, "x")
    B, In = x.shape
    Out = weight.shape[0]
    y = torch.empty((B, Out), dtype=torch.float32, device=x.device)
    call("ipdm_linear_f32", _ptr(x), _ptr(weight), _ptr(bias), _ptr(y), B, In, Out, act_in, _stream())
    return y


def attention(q, k, v, scale):
    q, k, v = (_gpu(t, torch.float32, n) for t, n in ((q, "q"), (k, "k"), (v, "v")))
    B, C, H, W = q.shape
    out = torch.empty_like(q)
    call("ipdm_attention_f32", _ptr(q), _ptr(k), _ptr(v), _ptr(out), B, C, H * W, float(scale), _stream())
    return out


def axpby(x, y, a, b, out=None):
    x, y = _gpu(x, torch.float32, "x"), _gpu(y, torch.float32, "y")
    if x.shape != y.shape:
        raise ValueError(f"ipdm axpby: shapes differ {tuple(x.shape)} vs {tuple(y.shape)}")
    out = torch.empty_like(x) if out is None else out
    call("ipdm_axpby_f32", _ptr(x), _ptr(y), _ptr(out), x.numel(), float(a), float(b), _stream())
    _written(out)
    return out


def sample_axpy2(x, y, a, z=None, c=None, out=None):
    """x + a[:,None]*y (+ c[:,None]*z): per-sample float32 device vectors a, c"""
    x = _gpu(x, torch.float32, "x")
    n = x.shape[0]
    out = torch.empty_like(x) if out is None else out
    a = _gpu(a.to(torch.float32), torch.float32, "a")
    c = None if c is None else _gpu(c.to(torch.float32), torch.float32, "c")
    call("ipdm_sample_axpy2_f32", _ptr(x), _ptr(y), _ptr(z), _ptr(a), _ptr(c), _ptr(out), n,
         x.numel() // max(n, 1), _stream())
    _written(out)
    return out


def sample_norm(x):
    x = _gpu(x, torch.float32, "x")
    n = x.shape[0]
    norms = torch.empty(n, dtype=torch.float32, device=x.device)
    call("ipdm_sample_norm_f32", _ptr(x), _ptr(norms), n, x.numel() // max(n, 1), _stream())
    return norms


# ---- convolution ------------------------------------------------------------------------------
def conv_pack_weight(w):
    """[Cout, Cin, k, k] -> [k*k, Cin, Cout]; a 5-D [Cout, Cin, 3, 3, 3] kernel -> [27, Cin, Cout]"""
    w = _gpu(w, torch.float32, "weight")
    if w.dim() == 5:
        Cout, Cin = w.shape[:2]
        kk = w.shape[2] * w.shape[3] * w.shape[4]
        if kk not in (1, 27):
            raise ValueError("3-D kernels must be 1x1x1 or 3x3x3")
        wt = torch.empty((kk, Cin, Cout), dtype=torch.float32, device=w.device)
        call("ipdm_conv_pack_weight_f32", _ptr(w), _ptr(wt), Cout, Cin, 27 if kk == 27 else 1, _stream())
        return wt
    Cout, Cin, k, k2 = w.shape
    assert k == k2
    wt = torch.empty((k * k, Cin, Cout), dtype=torch.float32, device=w.device)
    call("ipdm_conv_pack_weight_f32", _ptr(w), _ptr(wt), Cout, Cin, k, _stream())
    return wt


def conv_wino_weight(w):
    """[Cout, Cin, 3, 3] -> Winograd-domain weights U = G g G^T, [16, Cin, Cout]"""
    w = _gpu(w, torch.float32, "weight")
    Cout, Cin = w.shape[:2]
    U = torch.empty((16, Cin, Cout), dtype=torch.float32, device=w.device)
    call("ipdm_conv_wino_weight_f32", _ptr(w), _ptr(U), Cout, Cin, _stream())
    return U


def conv_wino_supported(Cin, Cout, H, W, dilation=1):
    return bool(_lib.lib.ipdm_conv2d_wino_supported(Cin, Cout, H, W, dilation))


def conv2d_wino(x, U, bias=None, residual=None, act_out=ACT_NONE, raw=True, dilation=1):
    """3x3 / dilation-1 convolution through the Winograd F(2x2,3x3) kernel (same output options as conv2d)"""
    x = _gpu(x, torch.float32, "x")
    B, Cin, H, W = x.shape
    Cout = U.shape[2]
    want_act = act_out != ACT_NONE
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device) if raw else None
    out_act = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device) if want_act else None
    if CONV_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("ipdm_conv2d_wino_f32", _ptr(x), _ptr(U), _ptr(bias), _ptr(residual), _ptr(out), _ptr(out_act), act_out,
         B, Cin, Cout, H, W, dilation, _stream())
    if CONV_TRACE is not None:
        e1.record()
        CONV_TRACE.append(dict(B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=3, dil=dilation, wino=True, res=residual is not None,
                               n_out=int(raw) + int(want_act), e0=e0, e1=e1))
    return (out, out_act) if want_act else out


def conv3d(x, wt, bias=None, coef=None, act=ACT_NONE, residual=None, dilation=1, act_out=ACT_NONE, raw=True, in_amax=None,
           want_amax=False, res_second=False):
    """x [B,Cin,D,H,W]; wt packed [27 or 1, Cin, Cout]; same fused input/output options as conv2d"""
    if isinstance(wt, PackedBx3) and wt.kk == 36:                # conv_wino1d_weight3d: the 1-D Winograd kernel's volume form
        if coef is not None or act != ACT_NONE or dilation != 1:
            raise ValueError("conv3d: the 1-D Winograd blob serves undilated launches without a fused input")
        return conv3d_wino1d(x, wt, bias, residual, act_out=act_out, raw=raw, in_amax=in_amax, want_amax=want_amax,
                             res_second=res_second)
    if isinstance(wt, PackedBx3):
        return conv_bx3(x, wt, bias, coef, act, residual, dilation, act_out=act_out, raw=raw, in_amax=in_amax,
                        want_amax=want_amax, res_second=res_second)
    if res_second:
        raise _lib.IpdmUnsupported("conv3d: res_second exists on the split-operand kernels only")
    x = _gpu(x, torch.float32, "x")
    B, Cin, D, H, W = x.shape
    kk, Cin_w, Cout = wt.shape
    if Cin_w != Cin:
        raise ValueError(f"conv3d: weight Cin {Cin_w} != input Cin {Cin}")
    k = {1: 1, 27: 3}[kk]
    want_act = act_out != ACT_NONE
    out = torch.empty((B, Cout, D, H, W), dtype=torch.float32, device=x.device) if raw else None
    out_act = torch.empty((B, Cout, D, H, W), dtype=torch.float32, device=x.device) if want_act else None
    if CONV_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("ipdm_conv3d_f32", _ptr(x), _ptr(wt), _ptr(bias), _ptr(coef), act, _ptr(residual), _ptr(out), _ptr(out_act),
         act_out, B, Cin, Cout, D, H, W, k, dilation, _stream())
    if CONV_TRACE is not None:
        e1.record()
        CONV_TRACE.append(dict(B=B * D, Cin=Cin, Cout=Cout, H=H, W=W, k=k, dil=dilation, taps3d=kk, res=residual is not None,
                               n_out=int(raw) + int(want_act), e0=e0, e1=e1))
    return (out, out_act) if want_act else out


def maxpool3d5(x):
    x = _gpu(x, torch.float32, "x")
    B, C, D, H, W = x.shape
    out = torch.empty_like(x)
    call("ipdm_maxpool3d5_f32", _ptr(x), _ptr(out), B * C, D, H, W, _stream())
    return carry_amax(x, out)


def temporal_taps(x, mode):
    """x [B,C,D,H,T] -> [B,4C,D,H,T'] (mode 0: T' = T/2 strided-conv taps; mode 1: T' = 2T transposed-conv taps)"""
    x = _gpu(x, torch.float32, "x")
    B, C, D, H, T = x.shape
    T_out = T // 2 if mode == 0 else T * 2
    out = torch.empty((B, 4 * C, D, H, T_out), dtype=torch.float32, device=x.device)
    call("ipdm_temporal_taps_f32", _ptr(x), _ptr(out), B * C, D * H, T, T_out, mode, _stream())
    return carry_amax(x, out)


def conv2d(x, wt, bias=None, coef=None, act=ACT_NONE, residual=None, dilation=1, pool2=False, out=None,
           act_out=ACT_NONE, raw=True, in_amax=None, out_scale=1.0, want_amax=False, res_second=False):
    """x [B,Cin,H,W]; wt packed [k*k,Cin,Cout].  Input side: optional InstanceNorm++ coefficients / activation.
    Output side: bias, residual add; act_out != NONE additionally returns the activated copy act_out(result)
    (raw=False: ONLY the activated copy is produced).  Returns out, or (out, out_act) when act_out is set
    (out is None when raw=False)."""
    if isinstance(wt, PackedBx3):
        if pool2:
            raise _lib.IpdmUnsupported("conv2d: pool2 epilogue is not fused")
        return conv_bx3(x, wt, bias, coef, act, residual, dilation, out=out, act_out=act_out, raw=raw, in_amax=in_amax,
                        out_scale=out_scale, want_amax=want_amax, res_second=res_second)
    if out_scale != 1.0 or (bias is not None and bias.dim() == 2) or res_second:
        raise _lib.IpdmUnsupported("conv2d: per-image bias / out_scale exist on the split-operand kernels only")
    x = _gpu(x, torch.float32, "x")
    B, Cin, H, W = x.shape
    kk, Cin_w, Cout = wt.shape
    if Cin_w != Cin:
        raise ValueError(f"conv2d: weight Cin {Cin_w} != input Cin {Cin}")
    k = {1: 1, 9: 3}[kk]
    oh, ow = (H // 2, W // 2) if pool2 else (H, W)
    want_act = act_out != ACT_NONE
    if not raw and not want_act:
        raise ValueError("conv2d: raw=False needs act_out")
    if raw:
        out = torch.empty((B, Cout, oh, ow), dtype=torch.float32, device=x.device) if out is None else out
    else:
        out = None
    out_act = torch.empty((B, Cout, oh, ow), dtype=torch.float32, device=x.device) if want_act else None
    if CONV_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("ipdm_conv2d_f32", _ptr(x), _ptr(wt), _ptr(bias), _ptr(coef), act, _ptr(residual), _ptr(out), _ptr(out_act),
         act_out, B, Cin, Cout, H, W, k, dilation, int(bool(pool2)), _stream())
    _written(out)
    if CONV_TRACE is not None:
        e1.record()
        CONV_TRACE.append(dict(B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=k, dil=dilation, res=residual is not None,
                               n_out=int(raw) + int(want_act), e0=e0, e1=e1))
    return (out, out_act) if want_act else out


USE_THIN_CONV = os.environ.get("IPDM_THIN_CONV", "1") != "0"


def conv3x3_thin_ok(Cin, Cout, H, W):
    """3x3, dilation 1, no fused input norm / residual / second output, and a thin side: the streaming kernels"""
    return USE_THIN_CONV and bool(_lib.lib.ipdm_conv3x3_thin_supported(int(Cin), int(Cout), int(H), int(W)))


def conv3x3_thin(x, weight, bias=None, coef=None):
    """first / last layer of a score network: x [B,Cin,H,W], weight [Cout,Cin,3,3] (the reference's layout), Cin <= 3 or
    Cout <= 3; coef [B,Cin,3]: input affine (x - c0) * c1 + c2 inside the image (Cin <= 3 form)"""
    x = _gpu(x, torch.float32, "x")
    weight = _gpu(weight, torch.float32, "weight")
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    if tuple(weight.shape) != (Cout, Cin, 3, 3):
        raise ValueError(f"conv3x3_thin: weight {tuple(weight.shape)} does not match input channels {Cin}")
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    if CONV_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if coef is not None:
        coef = _gpu(coef, torch.float32, "coef")
        if tuple(coef.shape) != (B, Cin, 3):
            raise ValueError(f"conv3x3_thin: coef {tuple(coef.shape)} != {(B, Cin, 3)}")
    call("ipdm_conv3x3_thin_f32", _ptr(x), _ptr(weight), _ptr(bias), _ptr(coef), _ptr(out), B, Cin, Cout, H, W, _stream())
    if CONV_TRACE is not None:
        e1.record()
        CONV_TRACE.append(dict(B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=3, dil=1, res=False, n_out=1, e0=e0, e1=e1, thin=True))
    return out


# ---- fp32 convolution on the 16-bit matrix cores (exactly split operands) ---------------------------------------
# Which kernel family the modules use (conv_weight()):
#   "hx2" (default): operands as two fp16 pieces, three fp16 MFMAs per product (csrc/conv_kernel.h) -- fp32-faithful
#                    for |activation| < 65504, NaN (never a wrong finite value) beyond
#   "bx3":           operands as three bf16 pieces, six bf16 MFMAs per product -- the whole fp32 exponent range
#   "f32":           the fp32-MFMA direct kernel + fp32 Winograd (conv.hip / conv_wino.hip)
CONV_IMPL = os.environ.get("IPDM_CONV_IMPL", "hx2")
SPLIT_IMPLS = ("hx2", "bx3")
if CONV_IMPL not in SPLIT_IMPLS + ("f32",):
    raise ValueError(f"IPDM_CONV_IMPL={CONV_IMPL!r}: expected 'hx2', 'bx3' or 'f32'")


def split_impl():
    """True when the modules run the split-operand kernels (conv_bx3.hip / conv_wino_bx3.hip)"""
    return CONV_IMPL in SPLIT_IMPLS


def impl_unbounded():
    """kernel family for networks whose convolution inputs are NOT bounded by a per-plane normalisation (NCSN++: GroupNorm'ed
    blocks but raw progressive / skip streams -- 1.7e5 in the full-size golden forward g22): the f16x2 family's STATIC range
    contract (|x| < 65504) does not hold there; these networks run it with the DYNAMIC range instead (`unbounded_amax()`:
    one ipdm_absmax_f32 pass per convolution input, every image scaled into fp16's range by an exact power of two).
    IPDM_CONV_IMPL_UNBOUNDED overrides the family (e.g. bx3: three bf16 pieces, the whole fp32 exponent range, no extra pass)."""
    env = os.environ.get("IPDM_CONV_IMPL_UNBOUNDED")
    if env:
        if env not in SPLIT_IMPLS + ("f32",):
            raise ValueError(f"IPDM_CONV_IMPL_UNBOUNDED={env!r}")
        return env
    return CONV_IMPL


def unbounded_amax():
    """the `in_amax` argument those networks pass to their convolutions: True (measure the input) on the f16x2 family"""
    return True if impl_unbounded() == "hx2" else None


class PackedWeightCache:
    """Packed copies of ONE parameter, one entry per (kernel family, layout), each tagged with the parameter's
    (version counter, storage pointer, device).  In-place optimiser steps and `copy_` on the parameter bump the version;
    a write through `.data` (the reference's EMA swap, helpers/utils.py:161-170) does not: modules clear the cache in
    their load_state_dict hook, and anything else that writes `.data` must call the module's invalidate()."""

    def __init__(self):
        self._entries = {}

    def clear(self):
        self._entries.clear()

    def get(self, weight, kind, build):
        tag = (weight._version, weight.data_ptr(), str(weight.device))
        hit = self._entries.get(kind)
        if hit is None or hit[0] != tag:
            hit = (tag, build(weight.data))
            self._entries[kind] = hit
        return hit[1]


class PackedWeightMixin:
    """for conv modules holding `self.weight`: `self._cache` + invalidate() + the load_state_dict hook"""

    def invalidate(self):
        self._cache.clear()

    def _load_from_state_dict(self, *args, **kwargs):
        self._cache.clear()
        return super()._load_from_state_dict(*args, **kwargs)


def conv_weight(w, impl=None):
    """pack a convolution weight for the selected kernel family (impl: override CONV_IMPL); pass the result to conv2d / conv3d"""
    impl = CONV_IMPL if impl is None else impl
    return conv_bx3_weight(w, fmt=impl) if impl in SPLIT_IMPLS else conv_pack_weight(w)


class PackedBx3:
    """weights split into 16-bit pieces and laid out as MFMA A-fragments: fmt "bx3" = three bf16 pieces
    (ipdm_conv_bx3_pack_weight), "hx2" = two fp16 pieces + per-output-channel inverse scales (ipdm_conv_hx2_pack_weight)"""
    __slots__ = ("blob", "Cout", "Cin", "kk", "fmt")

    def __init__(self, blob, Cout, Cin, kk, fmt="bx3"):
        self.blob, self.Cout, self.Cin, self.kk, self.fmt = blob, Cout, Cin, kk, fmt


def absmax_per_image(x):
    """per-image max |x| of an activation tensor as a maxima vector (float32 [B, AMAX_SLOT]; amax_value() -> [B]): the `in_amax`
    of the f16x2 convolutions' dynamic range where no producer handed the maxima over"""
    x = _gpu(x, torch.float32, "x")
    B = x.shape[0]
    out = torch.empty((B, AMAX_SLOT), dtype=torch.float32, device=x.device)
    call("ipdm_absmax_f32", _ptr(x), _ptr(out), B, x.numel() // max(B, 1), _stream())
    return out


def _conv_ext(fmt, in_amax, x, fused_input, bias_per_image=False, out_scale=1.0, Cout=0, out_amax=None, act_amax=None,
              res_second=False):
    """-> the `ext` argument of the split-operand entry points (NULL when every extra is at its default).
    in_amax (hx2 blobs only): None -> static range contract; True -> measure the input here; a tensor -> as given (a fused input
    normalisation / activation makes the raw maximum meaningless: ignored).  bias_per_image: bias is [B, Cout].  out_scale:
    result = (conv + bias + residual) * out_scale.  out_amax / act_amax: zeroed maxima vectors for what is stored.  res_second:
    the residual enters the second output only (out = conv + bias, out_act = act_out(conv + bias + residual))."""
    amax_ptr = None
    if fmt == "hx2" and in_amax is not None and not fused_input:
        if in_amax is True:
            in_amax = absmax_per_image(x)
        if (tuple(in_amax.shape) != (x.shape[0], AMAX_SLOT) or in_amax.dtype != torch.float32 or not in_amax.is_cuda
                or not in_amax.is_contiguous()):
            raise ValueError(f"in_amax: expected a maxima vector -- contiguous float32 GPU tensor [{x.shape[0]}, {AMAX_SLOT}] "
                             "(ops.absmax_per_image / a producer's tag)")
        amax_ptr = in_amax.data_ptr()
        _KEEP.append(in_amax)                    # (the kernel reads it asynchronously: keep the tensor alive until the call returns)
    if (amax_ptr is None and not bias_per_image and out_scale == 1.0 and out_amax is None and act_amax is None
            and not res_second):
        return P(0)
    ext = _lib.ConvExt(amax_ptr, int(Cout) if bias_per_image else 0, float(out_scale), int(bool(res_second)),
                       None if out_amax is None else out_amax.data_ptr(), None if act_amax is None else act_amax.data_ptr())
    _KEEP.append(ext)
    del _KEEP[:-8]
    return ctypes.byref(ext)


_KEEP = []                                       # last few ext structs / amax tensors (host structs are read during the call)


def conv_hx2_weight(w):
    return conv_bx3_weight(w, fmt="hx2")


def conv_bx3_weight(w, fmt="bx3"):
    """[Cout, Cin, k, k] (k = 1 or 3) or [Cout, Cin, 3, 3, 3] / [Cout, Cin, 1, 1, 1] -> PackedBx3"""
    if fmt not in SPLIT_IMPLS:
        raise ValueError(f"conv_bx3_weight: fmt {fmt!r}")
    w = _gpu(w, torch.float32, "weight")
    Cout, Cin = w.shape[:2]
    kk = 1
    for n in w.shape[2:]:
        kk *= int(n)
    if w.dim() == 5:
        if kk not in (1, 27):
            raise ValueError("3-D kernels must be 1x1x1 or 3x3x3")
        k = 27 if kk == 27 else 1
    else:
        k = {1: 1, 9: 3}[kk]
    nbytes = getattr(_lib.lib, f"ipdm_conv_{fmt}_weight_bytes")(Cout, Cin, k)
    blob = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    call(f"ipdm_conv_{fmt}_pack_weight", _ptr(w), _ptr(blob), Cout, Cin, k, _stream())
    return PackedBx3(blob, Cout, Cin, kk, fmt)


def conv_bx3(x, wq, bias=None, coef=None, act=ACT_NONE, residual=None, dilation=1, out=None, act_out=ACT_NONE, raw=True,
             in_amax=None, out_scale=1.0, want_amax=False, res_second=False):
    """2-D ([B,Cin,H,W]) or 3-D ([B,Cin,D,H,W]) convolution, same options / return convention as conv2d / conv3d.
    res_second (needs residual, act_out (ACT_COPY = none) and raw): -> (conv + bias, act_out(conv + bias + residual)).
    in_amax (hx2 blobs only): None = static range contract |x| < 65504, True = measure the input (ipdm_absmax_f32) and scale every
    image into fp16's range, or the per-image maxima themselves"""
    x = _gpu(x, torch.float32, "x")
    bias_per_image = bias is not None and bias.dim() == 2
    if bias_per_image and (tuple(bias.shape) != (x.shape[0], wq.Cout) or bias.stride(1) != 1):
        raise ValueError(f"conv_bx3: per-image bias {tuple(bias.shape)} != {(x.shape[0], wq.Cout)} (rows may be strided, not columns)")
    want_act = act_out != ACT_NONE
    want_amax = bool(want_amax) and x.shape[0] <= 65535 and os.environ.get("IPDM_AMAX_PRODUCE", "1") != "0"
    slot_o = amax_slot(x.shape[0], x.device) if want_amax and raw else None
    slot_a = amax_slot(x.shape[0], x.device) if want_amax and want_act else None
    if res_second and (residual is None or not want_act or not raw):
        raise ValueError("conv_bx3: res_second needs a residual and both outputs (act_out, e.g. ACT_COPY; raw=True)")
    ext = _conv_ext(wq.fmt, in_amax, x, coef is not None or act != ACT_NONE, bias_per_image, out_scale,
                    bias.stride(0) if bias_per_image else 0, slot_o, slot_a, res_second)
    if wq.Cin != x.shape[1]:
        raise ValueError(f"conv_bx3: weight Cin {wq.Cin} != input Cin {x.shape[1]}")
    vol = x.dim() == 5
    k = {1: 1, 9: 3, 27: 3}[wq.kk]
    if (wq.kk == 27) != (vol and k == 3):
        raise ValueError("conv_bx3: kernel / input rank mismatch")
    want_act = act_out != ACT_NONE
    if not raw and not want_act:
        raise ValueError("conv_bx3: raw=False needs act_out")
    shape = (x.shape[0], wq.Cout) + tuple(x.shape[2:])
    if raw:
        out = torch.empty(shape, dtype=torch.float32, device=x.device) if out is None else out
    else:
        out = None
    out_act = torch.empty(shape, dtype=torch.float32, device=x.device) if want_act else None
    if CONV_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if vol:
        B, Cin, D, H, W = x.shape
    else:
        B, Cin, H, W = x.shape
        D = 1
    ksplit = _lib.lib.ipdm_conv_bx3_splitk(B, D, Cin, wq.Cout, H, W, k, dilation) if B else 1
    if ksplit > 1:        # too few tiles to fill the chip: deal the K loop to several workgroups per tile
        work = torch.empty((ksplit,) + shape, dtype=torch.float32, device=x.device)
        call(f"ipdm_conv_{wq.fmt}_splitk_f32", _ptr(x), _ptr(wq.blob), _ptr(bias), _ptr(coef), act, _ptr(residual), _ptr(out),
             _ptr(out_act), act_out, B, Cin, wq.Cout, D, H, W, k, dilation, int(vol), ksplit, _ptr(work), ext, _stream())
    elif vol:
        call(f"ipdm_conv3d_{wq.fmt}_f32", _ptr(x), _ptr(wq.blob), _ptr(bias), _ptr(coef), act, _ptr(residual), _ptr(out),
             _ptr(out_act), act_out, B, Cin, wq.Cout, D, H, W, k, dilation, ext, _stream())
    else:
        call(f"ipdm_conv2d_{wq.fmt}_f32", _ptr(x), _ptr(wq.blob), _ptr(bias), _ptr(coef), act, _ptr(residual), _ptr(out),
             _ptr(out_act), act_out, B, Cin, wq.Cout, H, W, k, dilation, ext, _stream())
    if CONV_TRACE is not None:
        e1.record()
        CONV_TRACE.append(dict(B=B * D, Cin=Cin, Cout=wq.Cout, H=H, W=W, k=k, dil=dilation, bx3=True, fmt=wq.fmt, res=residual is not None,
                               n_out=int(raw) + int(want_act),
                               taps3d=wq.kk if vol else None, e0=e0, e1=e1))
    _written(out)                                   # (a caller's `out=` tensor: whatever was cached on it is stale)
    tag_amax(out, slot_o)
    tag_amax(out_act, slot_a)
    return (out, out_act) if want_act else out


def conv_wino_bx3_weight(w, fmt="bx3"):
    """[Cout, Cin, 3, 3] -> Winograd-domain weights, split into 16-bit pieces in MFMA fragment order (PackedBx3, kk=16)"""
    if fmt not in SPLIT_IMPLS:
        raise ValueError(f"conv_wino_bx3_weight: fmt {fmt!r}")
    w = _gpu(w, torch.float32, "weight")
    Cout, Cin = w.shape[:2]
    nbytes = getattr(_lib.lib, f"ipdm_conv_wino_{fmt}_weight_bytes")(Cout, Cin)
    blob = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    call(f"ipdm_conv_wino_{fmt}_pack_weight", _ptr(w), _ptr(blob), Cout, Cin, _stream())
    return PackedBx3(blob, Cout, Cin, 16, fmt)


def conv_wino_hx2_weight(w):
    return conv_wino_bx3_weight(w, fmt="hx2")


def conv_wino1d_weight(w):
    """[Cout, Cin, 3, 3] -> weights of the 1-D Winograd kernel (F(2,3) along x, the filter's rows as K; f16x2 pieces; PackedBx3, kk=12)"""
    w = _gpu(w, torch.float32, "weight")
    Cout, Cin = w.shape[:2]
    blob = torch.empty(_lib.lib.ipdm_conv_wino1d_weight_bytes(Cout, Cin), dtype=torch.uint8, device=w.device)
    call("ipdm_conv_wino1d_pack_weight", _ptr(w), _ptr(blob), Cout, Cin, _stream())
    return PackedBx3(blob, Cout, Cin, 12, "hx2")


def conv_wino1d_weight3d(w):
    """[Cout, Cin, 3, 3, 3] -> weights of the 1-D Winograd kernel's volume form (36 positions: depth tap x filter row x position)"""
    w = _gpu(w, torch.float32, "weight")
    Cout, Cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3, 3):
        raise ValueError(f"conv_wino1d_weight3d: weight {tuple(w.shape)}")
    blob = torch.empty(_lib.lib.ipdm_conv_wino1d_weight_bytes3d(Cout, Cin), dtype=torch.uint8, device=w.device)
    call("ipdm_conv_wino1d_pack_weight3d", _ptr(w), _ptr(blob), Cout, Cin, _stream())
    return PackedBx3(blob, Cout, Cin, 36, "hx2")


def conv3d_wino1d(x, U, bias=None, residual=None, act_out=ACT_NONE, raw=True, in_amax=None, want_amax=False, res_second=False):
    """3x3x3 'same' convolution of x [B, Cin, D, H, W] on the 1-D Winograd kernel (same output options as conv3d)"""
    x = _gpu(x, torch.float32, "x")
    B, Cin, D, H, W = x.shape
    if U.kk != 36 or U.Cin != Cin:
        raise ValueError("conv3d_wino1d: weight blob does not match the input")
    if res_second and (residual is None or act_out == ACT_NONE or not raw):
        raise ValueError("conv3d_wino1d: res_second needs a residual and both outputs")
    Cout = U.Cout
    if residual is not None and tuple(residual.shape) != (B, Cout, D, H, W):
        raise ValueError(f"conv3d_wino1d: residual {tuple(residual.shape)} != output {(B, Cout, D, H, W)}")
    amax_t = None
    if in_amax is not None:
        amax_t = absmax_per_image(x) if in_amax is True else in_amax
    want_amax = bool(want_amax) and B <= 65535 and os.environ.get("IPDM_AMAX_PRODUCE", "1") != "0"
    want_act = act_out != ACT_NONE
    slot_o = amax_slot(B, x.device) if want_amax and raw else None
    slot_a = amax_slot(B, x.device) if want_amax and want_act else None
    out = torch.empty((B, Cout, D, H, W), dtype=torch.float32, device=x.device) if raw else None
    out_act = torch.empty((B, Cout, D, H, W), dtype=torch.float32, device=x.device) if want_act else None
    if CONV_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    nb = max(1, (0x3fffffff - 1) // (Cin * D * H * W * 4))       # images per launch inside the buffer descriptor's reach
    for b0 in range(0, B, nb):
        b1 = min(B, b0 + nb)
        ext = _conv_ext("hx2", None if amax_t is None else amax_t[b0:b1], x[b0:b1], False, False, 1.0, 0,
                        None if slot_o is None else slot_o[b0:b1], None if slot_a is None else slot_a[b0:b1], res_second)
        call("ipdm_conv3d_wino1d_f32", _ptr(x[b0:b1]), _ptr(U.blob), _ptr(bias), _ptr(None if residual is None else residual[b0:b1]),
             _ptr(None if out is None else out[b0:b1]), _ptr(None if out_act is None else out_act[b0:b1]), act_out, b1 - b0, Cin,
             Cout, D, H, W, ext, _stream())
    if CONV_TRACE is not None:
        e1.record()
        CONV_TRACE.append(dict(B=B * D, Cin=Cin, Cout=Cout, H=H, W=W, k=3, dil=1, wino=True, bx3=True, fmt="hx2", wino1d=True,
                               res=residual is not None, n_out=int(raw) + int(want_act), taps3d=27, e0=e0, e1=e1))
    tag_amax(out, slot_o)
    tag_amax(out_act, slot_a)
    return (out, out_act) if want_act else out


def wino1d_vol_pays(Cin, Cout, D, H, W, dilation=1):
    """3x3x3 layers the 1-D Winograd kernel's volume form takes from the direct kernel (layer shape only)"""
    return (WINO1D and WINO1D_VOL and CONV_IMPL == "hx2" and dilation == 1
            and bool(_lib.lib.ipdm_conv3d_wino1d_supported(int(Cin), int(Cout), int(D), int(H), int(W))))


WINO1D_VOL = os.environ.get("IPDM_WINO1D_VOL", "1") != "0"          # 0: config 4's 3x3x3 layers stay on the direct kernel (A/B)
WINO1D = os.environ.get("IPDM_WINO1D", "1") != "0"          # 0: the 2-D Winograd kernel everywhere (A/B, fallback)
WINO1D_STATS = os.environ.get("IPDM_WINO1D_STATS", "1") != "0"      # the launches with a statistics epilogue too (tuning)
# InstanceNorm++ + ELU of the input inside the 1-D kernel's producer instead of the affine + activation pass: built, tested, and
# measured SLOWER on MI355X (17.24 -> 17.68 ms per iteration, same box: +1.1 ms of convolution for 0.8 ms of passes removed -- the
# exponentials and selects in a 256-register kernel cost more than 8 bytes per element at the HBM roof), so off by default
WINO1D_FIN = os.environ.get("IPDM_WINO1D_FIN", "0") != "0"


def wino1d_pays(Cin, Cout, H, W, dilation=1):
    """layers the 1-D Winograd kernel takes from the 2-D one (a function of the layer SHAPE only): f16x2 family, undilated,
    at least 128 output channels in blocks of 128, rows of 32 pixels or more"""
    return (WINO1D and CONV_IMPL == "hx2" and dilation == 1
            and bool(_lib.lib.ipdm_conv2d_wino1d_supported(int(Cin), int(Cout), int(H), int(W))))


def conv_wino_split_weight(w, impl=None):
    """Winograd-domain weights for the selected split family (CONV_IMPL = "hx2" / "bx3")"""
    impl = CONV_IMPL if impl is None else impl
    return conv_wino_bx3_weight(w, fmt=impl if impl in SPLIT_IMPLS else "bx3")


def conv_wino_bx3_supported(Cin, Cout, H, W, dilation=1):
    return bool(_lib.lib.ipdm_conv2d_wino_bx3_supported(Cin, Cout, H, W, dilation))


def conv2d_wino_bx3(x, U, bias=None, residual=None, act_out=ACT_NONE, raw=True, dilation=1, pool2=False, want_stats=False,
                    in_amax=None, out_scale=1.0, want_amax=False, res_second=False, coef=None, act=ACT_NONE):
    """3x3 convolution through the split-bf16 Winograd kernel (same output options as conv2d).
    res_second (needs residual, act_out (ACT_COPY = none) and raw): -> (conv + bias, act_out(conv + bias + residual)).
    pool2: the ConvMeanPool form -- outputs (and the residual) are [B, Cout, H/2, W/2] 2x2 means of the convolution;
    raises IpdmUnsupported where the pooled epilogue is not built (small / odd images).
    want_stats: the result feeds an InstanceNorm++ -- where the statistics epilogue exists for this shape, its partials
    [B, Cout, P, 3] are hung on the raw result as `_ipdm_partials` (instnorm_plus_coef picks them up)."""
    x = _gpu(x, torch.float32, "x")
    B, Cin, H, W = x.shape
    amax_t = None
    if U.fmt == "hx2" and in_amax is not None:
        if in_amax is True and coef is not None:
            raise ValueError("conv2d_wino_bx3: with a fused input the maxima are the coefficient kernel's bound, not max |x|")
        amax_t = absmax_per_image(x) if in_amax is True else in_amax
    bias_per_image = bias is not None and bias.dim() == 2
    if bias_per_image and (tuple(bias.shape) != (B, U.Cout) or bias.stride(1) != 1):
        raise ValueError(f"conv2d_wino_bx3: per-image bias {tuple(bias.shape)} != {(B, U.Cout)} (rows may be strided, not columns)")

    want_amax = bool(want_amax) and B <= 65535 and os.environ.get("IPDM_AMAX_PRODUCE", "1") != "0"
    slot_o = amax_slot(B, x.device) if want_amax and raw else None
    slot_a = amax_slot(B, x.device) if want_amax and act_out != ACT_NONE else None

    def ext_of(b0, b1):
        return _conv_ext(U.fmt, None if amax_t is None else amax_t[b0:b1], x[b0:b1], False, bias_per_image, out_scale,
                         bias.stride(0) if bias_per_image else 0,
                         None if slot_o is None else slot_o[b0:b1], None if slot_a is None else slot_a[b0:b1], res_second)

    def bias_of(b0, b1):
        return bias[b0:b1] if bias_per_image else bias
    one_d = U.kk == 12                   # conv_wino1d_weight: 1-D Winograd along x (conv_wino1d.hip), plain epilogues only
    if U.kk not in (12, 16) or U.Cin != Cin:
        raise ValueError("conv2d_wino_bx3: weight blob does not match the input")
    if one_d and (dilation != 1 or U.fmt != "hx2"):
        raise ValueError("conv2d_wino_bx3: the 1-D Winograd blob serves undilated f16x2 launches only")
    if coef is not None or act != ACT_NONE:
        # fused input act(InstanceNorm++(x)) (coef from instnorm_plus_coef): the 1-D kernel's producer applies it to the raw rows
        if not one_d or coef is None or act != ACT_ELU:
            raise ValueError("conv2d_wino_bx3: a fused input (coef + ACT_ELU) is the 1-D Winograd kernel's only")
        coef = _gpu(coef, torch.float32, "coef")
        if tuple(coef.shape) != (B, Cin, 3):
            raise ValueError(f"conv2d_wino_bx3: coef {tuple(coef.shape)} != {(B, Cin, 3)}")
    if res_second and (residual is None or act_out == ACT_NONE or not raw):
        raise ValueError("conv2d_wino_bx3: res_second needs a residual and both outputs (act_out, e.g. ACT_COPY; raw=True)")
    Cout = U.Cout
    want_act = act_out != ACT_NONE
    oh, ow = (H // 2, W // 2) if pool2 else (H, W)
    if residual is not None and tuple(residual.shape) != (B, Cout, oh, ow):
        raise ValueError(f"conv2d_wino_bx3: residual {tuple(residual.shape)} != output {(B, Cout, oh, ow)}")
    out = torch.empty((B, Cout, oh, ow), dtype=torch.float32, device=x.device) if raw else None
    out_act = torch.empty((B, Cout, oh, ow), dtype=torch.float32, device=x.device) if want_act else None
    if CONV_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    nb = wino_bx3_max_batch(Cin, H, W, dilation)
    ksplit = 1 if one_d or pool2 or x.data_ptr() % 16 or (WBX3_CO32 and U.fmt == "hx2") else wino_bx3_splitk(Cin, Cout, H, W, dilation)
    if ksplit > 1:                       # 16-pixel layers with few channel tiles: two K halves + a fixed-order reduction
        for b0 in range(0, B, nb):
            b1 = min(B, b0 + nb)
            work = torch.empty((ksplit, b1 - b0, Cout, H, W), dtype=torch.float32, device=x.device)
            call(f"ipdm_conv2d_wino_{U.fmt}_splitk_f32", _ptr(x[b0:b1]), _ptr(U.blob), _ptr(bias_of(b0, b1)),
                 _ptr(None if residual is None else residual[b0:b1]), _ptr(None if out is None else out[b0:b1]),
                 _ptr(None if out_act is None else out_act[b0:b1]), act_out, b1 - b0, Cin, Cout, H, W, dilation, ksplit,
                 _ptr(work), ext_of(b0, b1), _stream())
        if CONV_TRACE is not None:
            e1.record()
            CONV_TRACE.append(dict(B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=3, dil=dilation, wino=True, bx3=True, fmt=U.fmt, res=residual is not None,
                                   n_out=int(raw) + int(want_act), pool2=False, ksplit=ksplit, e0=e0, e1=e1))
        tag_amax(out, slot_o)
        tag_amax(out_act, slot_a)
        return (out, out_act) if want_act else out
    part = None
    if want_stats and raw and USE_STATS_EPILOGUE and x.data_ptr() % 16 == 0:
        P = (int(_lib.lib.ipdm_conv2d_wino1d_stats_partials(Cin, Cout, H, W)) if one_d else
             int(_lib.lib.ipdm_conv2d_wino_bx3_stats_partials(Cin, Cout, H, W, dilation, int(bool(pool2)))))
        if P > 0:
            part = torch.empty((B, Cout, P, 3), dtype=torch.float32, device=x.device)
    for b0 in range(0, B, nb):           # one launch unless the batch outgrows the kernel's 32-bit buffer offsets
        b1 = min(B, b0 + nb)
        args = (_ptr(x[b0:b1]), _ptr(U.blob), _ptr(bias_of(b0, b1)),
                _ptr(None if residual is None else residual[b0:b1]), _ptr(None if out is None else out[b0:b1]),
                _ptr(None if out_act is None else out_act[b0:b1]), act_out, b1 - b0, Cin, Cout, H, W, dilation,
                int(bool(pool2)))
        am = (ext_of(b0, b1),)
        if one_d:
            a1 = args[:3] + (_ptr(None if coef is None else coef[b0:b1]), act) + args[3:-2] + args[-1:]   # (+ fused input, no dilation)
            if part is not None:
                call("ipdm_conv2d_wino1d_stats_f32", *a1, _ptr(part[b0:b1]), *am, _stream())
            else:
                call("ipdm_conv2d_wino1d_f32", *a1, *am, _stream())
            continue
        if part is not None:
            try:
                call(f"ipdm_conv2d_wino_{U.fmt}_stats_f32", *args, _ptr(part[b0:b1]), *am, _stream())
                continue
            except _lib.IpdmUnsupported:         # e.g. IPDM_WBX3_DMA4=0: no statistics epilogue on that kernel form
                part = None
        call(f"ipdm_conv2d_wino_{U.fmt}_f32", *args, *am, _stream())
    if part is not None:
        out._ipdm_partials = (part, (out._version, out.data_ptr()))
    if CONV_TRACE is not None:
        e1.record()
        CONV_TRACE.append(dict(B=B, Cin=Cin, Cout=Cout, H=H, W=W, k=3, dil=dilation, wino=True, bx3=True, fmt=U.fmt, res=residual is not None,
                               n_out=int(raw) + int(want_act), pool2=bool(pool2), wino1d=one_d, stats=part is not None, e0=e0, e1=e1))
    tag_amax(out, slot_o)
    tag_amax(out_act, slot_a)
    return (out, out_act) if want_act else out


# f16x2 family: the 16-pixel layers whose split-K rule says "two K halves" run as ONE launch of 32-channel workgroups instead
# (conv_wino_bx3.hip: CO32; -0.35 % per iteration and 28 reduction launches fewer; 0: the split-K form)
WBX3_CO32 = os.environ.get("IPDM_WBX3_CO32", "1") != "0"


def wino_bx3_pays(Cin, Cout, H, W, dilation=1, B=None):
    """dispatch rule measured on MI355X (scripts/bench_conv.py): the split-bf16 Winograd kernel beats the direct
    split-bf16 kernel wherever it is eligible, except on undilated images of 16 pixels or less with fewer than 512
    output channels (one 8x8-tile workgroup per (image, channel tile): at the production batch only the 512-channel
    layers fill the chip; with 256 channels the direct kernel's finer tiles tie).
    The rule depends on the LAYER SHAPE ONLY -- never on the batch -- so that a sample's bits do not depend on how many
    other samples share its GPU (sharding.py's invariance; `B` is accepted and ignored for old call sites)."""
    if W <= 16 and dilation == 1:
        if wino_bx3_splitk(Cin, Cout, H, W, dilation) > 1:      # ... unless the layer runs as two K halves (shape rule too)
            return True
        if H > 16 or Cout < 512 or Cin < 32:
            return False
    return conv_wino_bx3_supported(Cin, Cout, H, W, dilation)


def wino_bx3_splitk(Cin, Cout, H, W, dilation=1):
    """K parts of the Winograd launch for this layer shape (1 = plain launch); never a function of the batch"""
    return int(_lib.lib.ipdm_conv2d_wino_bx3_splitk(int(Cin), int(Cout), int(H), int(W), int(dilation)))


def wino_bx3_max_batch(Cin, H, W, dilation=1):
    """images per launch the Winograd kernel's 32-bit buffer offsets reach (larger batches run as several launches)"""
    reach = 0x1fffffff if (W < 32 or dilation > 1) else 0x3fffffff
    return max(1, (reach - 1) // (Cin * H * W * 4))


def adam_ascent(x, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8):
    """one torch.optim.Adam step in place on x with param.grad = -g (x, g, m, v: same-shaped float32 GPU tensors)"""
    for t, n in ((x, "x"), (g, "g"), (m, "m"), (v, "v")):
        _inplace_operand(t, torch.float32, n)
        if t.numel() != x.numel():
            raise ValueError(f"adam_ascent: {n} does not match x")
    call("ipdm_adam_ascent_f32", _ptr(x), _ptr(g), _ptr(m), _ptr(v), x.numel(), float(lr), float(betas[0]), float(betas[1]),
         float(eps), int(step), _stream())
    _written(x, m, v)
    return x
