"""NRMSE / SSIM / posterior moments as the reference defines them (oracle; test infrastructure only).

Reference anchors: helpers/metrics.py:55-92.  The reference calls scikit-image, which is NOT
installed in this image: these are restatements of skimage's published algorithms
(skimage.metrics.normalized_root_mse 'euclidean'; structural_similarity with its defaults: 7x7
uniform window, K1=0.01, K2=0.03, sample covariance, mean over the valid interior) -- parity for
this file is UNPINNED against skimage itself (none here to generate vectors from); what pins it is
tests/test_oracle_golden.py::test_ssim_against_an_independent_restatement: the published definition
evaluated by plain loops over windows (no routine shared with this file) and closed forms, to 1e-12.
"""
import numpy as np
from scipy.ndimage import uniform_filter


def nrmse(img, img_orig):
    """helpers/metrics.py:70-74: normalised by the FIRST argument's RMS."""
    img = np.asarray(img, dtype=np.float64)
    img_orig = np.asarray(img_orig, dtype=np.float64)
    return float(np.sqrt(np.mean((img - img_orig) ** 2)) / np.sqrt(np.mean(img ** 2)))


def ssim(a, b, data_range=None, win=7, K1=0.01, K2=0.03):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if data_range is None:
        data_range = 2.0                      # skimage's float default (dtype range -1..1)
    NP = win ** a.ndim
    cov_norm = NP / (NP - 1)
    ux, uy = uniform_filter(a, win), uniform_filter(b, win)
    uxx, uyy, uxy = uniform_filter(a * a, win), uniform_filter(b * b, win), uniform_filter(a * b, win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return float(S[tuple(slice(pad, s - pad) for s in S.shape)].mean())


def posterior_moments(samples):
    """helpers/metrics.py:77-92 for complex input: compute_mean_and_std on |x| and on angle(x), each through the
    real-valued branch `np.mean(imgs), np.std(np.abs(imgs))` (:82-83) -- so the phase "std" is the std of |angle|."""
    mag, ph = np.abs(samples), np.angle(samples)
    return mag.mean(0), ph.mean(0), np.abs(mag).std(0), np.abs(ph).std(0)
