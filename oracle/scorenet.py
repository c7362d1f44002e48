"""torch-CPU functional restatement of the NCSNv2 score network family (oracle; test
infrastructure only).  Works directly on a state dict with the reference's key names.

Reference anchors (ncsn/models/):
  instance_norm_plus   normalization.py:150-176
  conv_mean_pool       layers.py:291-313
  residual_block       layers.py:401-456
  rcu_block            layers.py:112-134
  crp_block            layers.py:62-83     (MaxPool2d(5, 1, 2))
  msf_block            layers.py:165-184   (bilinear, align_corners=True)
  refine_block         layers.py:214-249
  ncsnv2_deepest       ncsnv2.py:198-299
  ncsnv2 / deeper      ncsnv2.py:11-101 / 104-195
"""
import torch
import torch.nn.functional as F


def sub(sd, prefix):
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


def instance_norm_plus(x, p):
    means = x.mean(dim=(2, 3))
    m = means.mean(dim=-1, keepdim=True)
    v = means.var(dim=-1, keepdim=True)                      # unbiased over channels
    means = (means - m) / torch.sqrt(v + 1e-5)
    h = F.instance_norm(x, eps=1e-5)
    h = h + means[..., None, None] * p["alpha"][None, :, None, None]
    out = p["gamma"][None, :, None, None] * h
    if "beta" in p:
        out = out + p["beta"][None, :, None, None]
    return out


def mean_pool2(x):
    return (x[:, :, ::2, ::2] + x[:, :, 1::2, ::2] + x[:, :, ::2, 1::2] + x[:, :, 1::2, 1::2]) / 4.0


def conv(x, p, dilation=1):
    w = p["weight"]
    pad = (w.shape[-1] // 2) * dilation
    return F.conv2d(x, w, p.get("bias"), padding=pad, dilation=dilation)


def residual_block(x, p, dilation=None, act=F.elu):
    d = 1 if dilation is None else dilation
    h = act(instance_norm_plus(x, sub(p, "normalize1")))
    h = conv(h, sub(p, "conv1"), d)
    h = act(instance_norm_plus(h, sub(p, "normalize2")))
    if "conv2.conv.weight" in p:                              # ConvMeanPool branch
        h = mean_pool2(conv(h, sub(p, "conv2.conv")))
    else:
        h = conv(h, sub(p, "conv2"), d)
    if "shortcut.conv.weight" in p:
        sc = mean_pool2(conv(x, sub(p, "shortcut.conv")))
    elif "shortcut.weight" in p:
        sp = sub(p, "shortcut")
        sc = conv(x, sp, d if sp["weight"].shape[-1] == 3 else 1)
    else:
        sc = x
    return sc + h


def rcu_block(x, p, n_blocks, n_stages=2, act=F.elu):
    for i in range(n_blocks):
        r = x
        for j in range(n_stages):
            x = conv(act(x), {"weight": p[f"{i + 1}_{j + 1}_conv.weight"]})
        x = x + r
    return x


def crp_block(x, p, n_stages=2, act=F.elu):
    x = act(x)
    path = x
    for i in range(n_stages):
        path = F.max_pool2d(path, 5, 1, 2)
        path = conv(path, {"weight": p[f"convs.{i}.weight"]})
        x = path + x
    return x


def msf_block(xs, p, shape):
    out = None
    for i, x in enumerate(xs):
        h = conv(x, sub(p, f"convs.{i}"))
        h = F.interpolate(h, size=shape, mode="bilinear", align_corners=True)
        out = h if out is None else out + h
    return out


def refine_block(xs, p, shape, end=False, act=F.elu):
    hs = [rcu_block(x, sub(p, f"adapt_convs.{i}"), 2, 2, act) for i, x in enumerate(xs)]
    h = msf_block(hs, sub(p, "msf"), shape) if len(xs) > 1 else hs[0]
    h = crp_block(h, sub(p, "crp"), 2, act)
    return rcu_block(h, sub(p, "output_convs"), 3 if end else 1, 2, act)


def _stage(x, sd, name, dilation=None):
    for i in range(2):
        x = residual_block(x, sub(sd, f"{name}.{i}"), dilation)
    return x


def ncsnv2_deepest(x, labels, sd, sigmas=None, rescale_input=True):
    """x (B, C, H, W) f32, labels (B,) int64 -> score, same shape."""
    sigmas = sd["sigmas"] if sigmas is None else sigmas
    h = 2 * x - 1.0 if rescale_input else x
    out = conv(h, sub(sd, "begin_conv"))
    l1 = _stage(out, sd, "res1")
    l2 = _stage(l1, sd, "res2")
    l3 = _stage(l2, sd, "res3")
    l31 = _stage(l3, sd, "res31")
    l4 = _stage(l31, sd, "res4", 2)
    l5 = _stage(l4, sd, "res5", 4)
    r1 = refine_block([l5], sub(sd, "refine1"), l5.shape[2:])
    r2 = refine_block([l4, r1], sub(sd, "refine2"), l4.shape[2:])
    r31 = refine_block([l31, r2], sub(sd, "refine31"), l31.shape[2:])
    r3 = refine_block([l3, r31], sub(sd, "refine3"), l3.shape[2:])
    r4 = refine_block([l2, r3], sub(sd, "refine4"), l2.shape[2:])
    o = refine_block([l1, r4], sub(sd, "refine5"), l1.shape[2:], end=True)
    o = F.elu(instance_norm_plus(o, sub(sd, "normalizer")))
    o = conv(o, sub(sd, "end_conv"))
    return o / sigmas[labels].view(-1, 1, 1, 1)
