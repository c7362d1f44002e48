"""Reconstruction metrics with the reference's definitions (mirror of ``helpers/metrics.py:21-102``).  The reference
calls scikit-image (not installed here); these are numpy restatements of skimage's published algorithms:
``normalized_root_mse(image_true, image_test, 'euclidean')`` -- NB the reference passes the RECONSTRUCTION first,
so it is the normaliser (:72) -- and ``structural_similarity`` with its defaults (7x7 uniform window, K1 0.01,
K2 0.03, sample covariance, mean over the valid interior).  The numpy functions keep the reference's host API; the
``*_device`` functions below run the same definitions on the GPU (libipdm.so: ipdm_nrmse_f32, ipdm_ssim_f32,
ipdm_posterior_moments_c64, ipdm_magnitude_c64) so that a 105-sample study is scored without a host round trip."""
from collections import defaultdict
from typing import Dict, List

import numpy as np
from scipy.ndimage import uniform_filter


def add_first_channel(img: np.ndarray) -> np.ndarray:
    return img[None] if img.ndim == 3 else img


def MAE(img: np.ndarray, img_orig: np.ndarray) -> float:
    return float(np.abs(img - img_orig).mean())


def NRMSE_wrapper(img: np.ndarray, img_orig: np.ndarray) -> float:
    img = np.asarray(img, dtype=np.float64)
    img_orig = np.asarray(img_orig, dtype=np.float64)
    return float(np.sqrt(np.mean((img - img_orig) ** 2)) / np.sqrt(np.mean(img ** 2)))


def _ssim2d(a, b, data_range, win=7, K1=0.01, K2=0.03):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    NP = win ** a.ndim
    cov_norm = NP / (NP - 1)
    ux, uy = uniform_filter(a, win), uniform_filter(b, win)
    vx = cov_norm * (uniform_filter(a * a, win) - ux * ux)
    vy = cov_norm * (uniform_filter(b * b, win) - uy * uy)
    vxy = cov_norm * (uniform_filter(a * b, win) - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return float(S[tuple(slice(pad, s - pad) for s in S.shape)].mean())


def SSIM_wrapper(img: np.ndarray, img_orig: np.ndarray, data_range: float = 2.0) -> float:
    """img, img_orig: (C, H, W).  data_range defaults to skimage's float range (-1..1 -> 2.0), as the reference's call"""
    if img.shape[0] > 1:
        return float(np.mean([_ssim2d(img[c], img_orig[c], data_range) for c in range(img.shape[0])]))
    return _ssim2d(img[0], img_orig[0], data_range)


REGISTERED_METRICS = {"NRMSE": NRMSE_wrapper, "SSIM": SSIM_wrapper, "MAE": MAE}
REGISTERED_REDUCTION = {"mean": np.mean, "std": np.std}


def compute_metrics(metric_names: List[str], img: np.ndarray, img_orig: np.ndarray, reduce=None) -> Dict[str, List[float]]:
    """img: (B, C, H, W)"""
    out = defaultdict(list)
    img, img_orig = add_first_channel(img), add_first_channel(img_orig)
    for name in metric_names:
        fn = REGISTERED_METRICS[name]
        vals = [fn(img[i], img_orig[0 if img_orig.shape[0] == 1 else i]) for i in range(img.shape[0])]
        out[name] = np.array(vals)
        if reduce is not None:
            out[name] = REGISTERED_REDUCTION[reduce](out[name])
    return out


def compute_mean_and_std(imgs: np.ndarray):
    """(B, C, H, W): real -> (mean, std of |.|); complex -> (mag_mean, phase_mean, mag_std, phase_std)"""
    assert imgs.shape[0] > 1
    if not np.iscomplexobj(imgs):
        return np.mean(imgs, axis=0), np.std(np.abs(imgs), axis=0)
    mag_mean, mag_std = compute_mean_and_std(np.abs(imgs))
    phase_mean, phase_std = compute_mean_and_std(np.angle(imgs))
    return mag_mean, phase_mean, mag_std, phase_std


def compute_snr(imgs: np.ndarray):
    imgs = np.abs(imgs)
    axes = tuple(range(1, len(imgs.shape)))
    return 20 * np.log10(imgs.max(axis=axes) / np.std(imgs, axis=axes))


# ---- on-device variants -----------------------------------------------------------------------
def _magnitude_device(x):
    import torch
    from .. import ops
    if x.is_complex():
        return ops.magnitude(x.to(torch.complex64).contiguous())
    return x.float().contiguous()


def compute_metrics_device(metric_names: List[str], img, img_orig, reduce=None, data_range: float = 2.0):
    """compute_metrics on GPU tensors.  img (B, 1, H, W) (complex: scored on magnitudes, as helpers/visualizations.py:93),
    img_orig (1 or B, 1, H, W).  Returns {name: float64 numpy array (B,)} (or the reduced scalar)."""
    from .. import ops
    if not img.is_cuda:
        raise RuntimeError("compute_metrics_device: expected GPU tensors (use compute_metrics for numpy arrays)")
    a, b = _magnitude_device(img), _magnitude_device(img_orig.to(img.device))
    if a.dim() != 4 or a.shape[1] != 1:
        raise NotImplementedError("device metrics score single-channel (B, 1, H, W) images")
    out = {}
    for name in metric_names:
        if name == "NRMSE":
            v = ops.nrmse(a, b)
        elif name == "SSIM":
            v = ops.ssim(a, b, data_range)
        elif name == "MAE":
            v = (a - b).abs().double().mean(dim=(1, 2, 3))
        else:
            raise KeyError(name)
        v = v.cpu().numpy()
        out[name] = REGISTERED_REDUCTION[reduce](v) if reduce is not None else v
    return out


def compute_mean_and_std_device(imgs):
    """compute_mean_and_std for a (B, C, H, W) complex GPU tensor -> (mag_mean, phase_mean, mag_std, phase_std) GPU tensors
    (population std, as np.std), from the six moment planes of one kernel"""
    from .. import sharding
    assert imgs.shape[0] > 1 and imgs.is_complex()
    post = sharding.posterior_from_moments(sharding.moment_planes(imgs), imgs.shape[0])
    return post["mag_mean"], post["phase_mean"], post["mag_std"], post["phase_std"]
