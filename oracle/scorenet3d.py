"""torch-CPU functional restatement of the temporal score network NCSN3DShallow (oracle; test infrastructure only --
imported by tests/ alone).  Works directly on a state dict with the reference's key names, in whatever dtype the
tensors carry (float64 for the range studies).

Reference anchors (ncsn/models/):
  instance_norm_plus   normalization3d.py:164-182   (statistics over all spatial dims)
  residual_block       layers3d.py:465-478          (all stages dilated: no pooling)
  rcu_block / crp_block / msf_block / refine_block   layers3d.py:77-84 (MaxPool3d(5,1,2)), :179-187 (trilinear,
                       align_corners=True), RCU / Refine as the 2-D blocks with Conv3d
  ncsn3d_shallow       ncsn3d.py:184-224
Pinned by the reference's own forward: tests/golden/g16_ncsn3d.npz (tests/test_oracle_golden.py)."""
import numpy as np
import torch
import torch.nn.functional as F

from .scorenet import sub


def instance_norm_plus(x, p):
    dims = tuple(range(2, x.dim()))
    means = x.mean(dim=dims)
    m = means.mean(dim=-1, keepdim=True)
    v = means.var(dim=-1, keepdim=True)
    means = (means - m) / torch.sqrt(v + 1e-5)
    h = F.instance_norm(x, eps=1e-5)
    tail = (None,) * len(dims)
    h = h + means[(...,) + tail] * p["alpha"][(...,) + tail]
    out = p["gamma"].view(1, -1, *([1] * len(dims))) * h
    if "beta" in p:
        out = out + p["beta"].view(1, -1, *([1] * len(dims)))
    return out


def conv(x, p, dilation=1):
    w = p["weight"]
    pad = (w.shape[-1] // 2) * dilation
    return F.conv3d(x, w, p.get("bias"), padding=pad, dilation=dilation)


def residual_block(x, p, dilation=None, act=F.elu):
    d = 1 if dilation is None else dilation
    h = act(instance_norm_plus(x, sub(p, "normalize1")))
    h = conv(h, sub(p, "conv1"), d)
    h = act(instance_norm_plus(h, sub(p, "normalize2")))
    h = conv(h, sub(p, "conv2"), d)
    if "shortcut.weight" in p:
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
        path = F.max_pool3d(path, 5, 1, 2)
        path = conv(path, {"weight": p[f"convs.{i}.weight"]})
        x = path + x
    return x


def msf_block(xs, p, shape):
    out = None
    for i, x in enumerate(xs):
        h = conv(x, sub(p, f"convs.{i}"))
        h = F.interpolate(h, size=tuple(shape), mode="trilinear", align_corners=True)
        out = h if out is None else out + h
    return out


def refine_block(xs, p, shape, end=False, act=F.elu):
    hs = [rcu_block(x, sub(p, f"adapt_convs.{i}"), 2, 2, act) for i, x in enumerate(xs)]
    h = msf_block(hs, sub(p, "msf"), shape) if len(xs) > 1 else hs[0]
    h = crp_block(h, sub(p, "crp"), 2, act)
    return rcu_block(h, sub(p, "output_convs"), 3 if end else 1, 2, act)


def ncsn3d_shallow(x, labels, sd, sigmas=None):
    """x (B, kx*ky, T) or (B, 1, kx, ky, T), labels (B,) int64 -> score, same shape (ncsn3d.py:184-224)"""
    sigmas = sd["sigmas"] if sigmas is None else sigmas
    flat = x.dim() == 3
    if flat:
        k = int(np.sqrt(x.shape[1]))
        x = x.reshape(x.shape[0], 1, k, k, x.shape[-1])
    out = conv(2 * x - 1.0, sub(sd, "begin_conv"))
    l1 = out
    for i in range(2):
        l1 = residual_block(l1, sub(sd, f"res1.{i}"))
    l2 = l1
    for i in range(2):
        l2 = residual_block(l2, sub(sd, f"res3.{i}"), 2)
    l3 = F.conv3d(l2, sd["conv_temporal_down.weight"], sd["conv_temporal_down.bias"], stride=(1, 1, 2), padding=(0, 0, 1))
    l4 = l3
    for i in range(2):
        l4 = residual_block(l4, sub(sd, f"res4.{i}"), 4)
    r1 = refine_block([l4], sub(sd, "refine1"), l4.shape[2:])
    r2 = refine_block([l3, r1], sub(sd, "refine2"), l3.shape[2:])
    r3 = F.conv_transpose3d(r2, sd["conv_temporal_up.weight"], sd["conv_temporal_up.bias"], stride=(1, 1, 2), padding=(0, 0, 1))
    o = refine_block([l1, r3], sub(sd, "refine3"), l1.shape[2:])
    o = F.elu(instance_norm_plus(o, sub(sd, "normalizer")))
    o = conv(o, sub(sd, "end_conv"))
    o = o / sigmas[labels].view(-1, 1, 1, 1, 1)
    return o.reshape(o.shape[0], -1, o.shape[-1]) if flat else o
