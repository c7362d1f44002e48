"""Forward operators of the inverse problem (host-side mirror of the reference's
``ncsn/linear_transforms/__init__.py``: LinearTransform :6-33, i2k_complex :36-45, k2i_complex :48-57,
generate_mask :60-76).  Image-space tensors live on the GPU; the transforms run in libipdm.so.
Mask generation is host-side integer logic (numpy MT19937) exactly as in the reference."""
import abc

import numpy as np
import torch

from ... import ops

# (sw, sm, sa) sets for generate_mask.  R20/R16/R8 are the reference's (undersampling_fourier.py:68-73);
# R40 is this build's choice: the reference ships no R=40 set.  seed 0, N=128 -> 3 lines.
MASK_PARAMS = {
    20: dict(sw=0.07, sm=0.3, sa=0.01782),
    16: dict(sw=0.07926, sm=0.42, sa=0.02),
    8: dict(sw=0.196, sm=0.5, sa=0.02),
    40: dict(sw=0.07, sm=0.11, sa=0.0065),
}


class LinearTransform(abc.ABC):
    """All inputs: (B, C, H, W)"""

    @abc.abstractmethod
    def __call__(self, X: torch.Tensor) -> torch.Tensor:
        return X

    @abc.abstractmethod
    def conj_op(self, S: torch.Tensor) -> torch.Tensor:
        return S

    @abc.abstractmethod
    def projection(self, X: torch.Tensor, S: torch.Tensor, lamda: float) -> torch.Tensor:
        return X

    def log_lh_grad(self, X: torch.Tensor, S: torch.Tensor, lamda: float = 1.) -> torch.Tensor:
        """grad = -lamda * A'(Ax - s)"""
        return -self.conj_op(self(X) - S) * lamda


def i2k_complex(X):
    """centred orthonormal 2-D FFT over the last two dims -> complex64"""
    return ops.fft2c(X, inverse=False)


def k2i_complex(X):
    return ops.fft2c(X, inverse=True)


def generate_mask(T: int, N: int, sw=0.3, sm=0.7, sa=0.045, T_max=1000, dev=0.01, seed=None):
    """variable-density random line mask: bool (1, N) if T == 1 else (T, 1, N).  The draw order of the
    legacy numpy generator (seed, rand, choice) is part of the contract: masks are bit-exact."""
    np.random.seed(seed)
    grid = np.linspace(-1, 1, N)
    density = sm * np.exp(-np.abs(grid) / sw) + sa
    candidates = np.random.rand(N, T_max) <= density[:, None]
    mid = N // 2
    candidates[mid - 1:mid + 1, :] = True
    per_mask = candidates.mean(axis=0)
    good = candidates[:, np.abs(per_mask - candidates.mean()) < dev]
    picks = np.random.choice(good.shape[1], T)
    chosen = np.ascontiguousarray(good[:, picks].T)
    if T == 1:
        return torch.from_numpy(chosen[0:1, :])
    return torch.from_numpy(chosen[:, None, :])
