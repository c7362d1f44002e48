"""Synthetic inputs for the benchmark and the parity tests (no data or checkpoints ship with the
reference: SURVEY.md 8c/8d).

* ``synth_state_dict``  -- deterministic random-init weights for any score net, keyed by the
  reference's state-dict names, independent of module construction order.
* ``phantom_image``     -- 6-ellipse magnitude phantom with a smooth phase, mirroring what
  ``helpers/load_data.py:372-387`` (add_phase: bicubic upsampling of an N(0,1) 5x5 patch) produces.

Everything here is host-side torch-CPU so the very same tensors are produced on the build
container and on the GPU box.
"""
import hashlib
import math

import torch
import torch.nn.functional as F


def _key_generator(key: str, seed: int) -> torch.Generator:
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    return torch.Generator().manual_seed(int.from_bytes(h[:7], "little"))


def synth_state_dict(shapes, seed=0):
    """shapes: {state-dict key: shape tuple}.  Conv / linear weights ~ U(-b, b), b = 1/sqrt(fan_in)
    (the scale torch's default conv init gives), norm scales ~ N(1, 0.02), shifts ~ N(0, 0.02)."""
    out = {}
    for key, shape in shapes.items():
        if key == "sigmas" or key.endswith(".sigmas"):
            continue
        g = _key_generator(key, seed)
        shape = tuple(shape)
        leaf = key.rsplit(".", 1)[-1]
        if len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            b = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * b
        elif leaf in ("alpha", "gamma"):
            t = 1.0 + 0.02 * torch.randn(shape, generator=g)
        elif leaf == "weight":            # 1-D weight = affine norm scale
            t = 1.0 + 0.02 * torch.randn(shape, generator=g)
        else:                             # bias / beta
            t = 0.02 * torch.randn(shape, generator=g)
        out[key] = t.float()
    return out


def phantom_image(H=128, W=128, seed=0, n_ellipses=6, phase_patch=(5, 5)):
    """complex64 (1, 1, H, W): magnitude in [0, 1], smooth phase."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    mag = torch.zeros(H, W)
    for _ in range(n_ellipses):
        cx, cy = (torch.rand(2, generator=g) * 1.0 - 0.5).tolist()
        ax, ay = (torch.rand(2, generator=g) * 0.45 + 0.1).tolist()
        th = float(torch.rand(1, generator=g)) * math.pi
        val = float(torch.rand(1, generator=g)) * 0.6 + 0.2
        xr = (xx - cx) * math.cos(th) + (yy - cy) * math.sin(th)
        yr = -(xx - cx) * math.sin(th) + (yy - cy) * math.cos(th)
        mag = mag + val * ((xr / ax) ** 2 + (yr / ay) ** 2 <= 1.0).float()
    mag = mag / mag.max().clamp_min(1e-6)
    patch = torch.randn(1, 1, *phase_patch, generator=g)
    phase = F.interpolate(patch, size=(H, W), mode="bicubic", align_corners=True)[0, 0]
    img = torch.polar(mag, phase).to(torch.complex64)
    return img[None, None]
