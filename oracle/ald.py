"""Annealed Langevin Dynamics sampler loops, CPU restatement (oracle; test infrastructure only).

Reference anchors (ncsn/models/ALD_optimizers.py):
  ald_unconditional      :66-137   (ALDOptimizer.__call__ / ALDUnconditionalSampler)
  ald_sense_real_imag    :172-270  (ALDInvSegProximalRealImag.__call__) with
                         :288-327  (post_processing -> proximal(x, y, step_lr*lr_scaled, 1.))
Segmentation guidance: off (weight 0, seg_start_time = 1, ALD_optimizers.py:27-28) unless a seg_grad_fn and the
lh_weights ramp are passed.  Noise is INJECTED (noise_fn) because the reference draws it on the
compute device.
"""
import numpy as np
import torch

from . import kspace


def step_size_of(step_lr, sigma, sigma_last):
    """float32 tensor arithmetic exactly as `step_lr * (sigma / sigmas[-1]) ** 2` (:217)."""
    return step_lr * (sigma / sigma_last) ** 2


def ald_unconditional(score_fn, sigmas, x0, step_lr, n_steps_each, denoise, noise_fn):
    sigmas = torch.as_tensor(sigmas)
    x = x0.clone()
    B = x.shape[0]
    for c, sigma in enumerate(sigmas):
        labels = torch.full((B,), c, dtype=torch.long)
        step = step_size_of(step_lr, sigma, sigmas[-1])
        for _ in range(n_steps_each):
            grad = score_fn(x, labels)
            noise = noise_fn(x)
            x = x + step * grad + noise * torch.sqrt(step * 2)
    if denoise:
        last = torch.full((B,), len(sigmas) - 1, dtype=torch.long)
        x = x + sigmas[-1] ** 2 * score_fn(x, last)
    return x


def ald_sense_real_imag(score_fn, sigmas, measurement, maps, mask, step_lr, n_steps_each, lr_scaled,
                        denoise, noise_fn, n_levels=None, start_level=0, x_init=None, seg_grad_fn=None, lh_weights=None):
    """measurement (n_coils, B, 1, H, W) complex64 numpy.  Returns complex64 numpy (B, 1, H, W).
    n_levels / start_level / x_init let the CPU baseline time a bounded slice of the schedule."""
    sigmas = torch.as_tensor(sigmas)
    x = kspace.sense_adjoint(measurement, maps) if x_init is None else x_init
    x_re = torch.from_numpy(np.ascontiguousarray(x.real))
    x_im = torch.from_numpy(np.ascontiguousarray(x.imag))
    B = x_re.shape[0]
    levels = range(start_level, len(sigmas) if n_levels is None else min(len(sigmas), start_level + n_levels))
    for c in levels:
        sigma = sigmas[c]
        labels = torch.full((B,), c, dtype=torch.long)
        step = step_size_of(step_lr, sigma, sigmas[-1])
        for _ in range(n_steps_each):
            g_re = score_fn(x_re, labels)
            g_im = score_fn(x_im, labels)
            if seg_grad_fn is not None:          # adjust_grad (:272-286): grad + grad_log_lh_seg / sigma * lamda
                g_re = g_re + seg_grad_fn(x_re) / sigma * lh_weights[c]
                g_im = g_im + seg_grad_fn(x_im) / sigma * lh_weights[c]
            x_re = x_re + step * g_re + noise_fn(x_re) * torch.sqrt(step * 2)
            x_im = x_im + step * g_im + noise_fn(x_im) * torch.sqrt(step * 2)
            z = (x_re.numpy() + 1j * x_im.numpy()).astype(np.complex64)
            z = kspace.l2_penalty_sense(z, measurement, step_lr * lr_scaled, 1.0, maps, mask)
            x_re = torch.from_numpy(np.ascontiguousarray(z.real))
            x_im = torch.from_numpy(np.ascontiguousarray(z.imag))
    if denoise and n_levels is None:
        last = torch.full((B,), len(sigmas) - 1, dtype=torch.long)
        x_re = x_re + sigmas[-1] ** 2 * score_fn(x_re, last)
        x_im = x_im + sigmas[-1] ** 2 * score_fn(x_im, last)
    return (x_re.numpy() + 1j * x_im.numpy()).astype(np.complex64)
