"""Undersampled-Fourier and multi-coil SENSE operators (mirror of the reference's
``ncsn/linear_transforms/undersampling_fourier.py``: UndersamplingFourier :10-36, RandomUndersamplingFourier :39-97,
SENSE :100-176).

Differences the reference forces on a drop-in, all explicit:
* the checked-in ``_generate_mask`` ignores ``R`` and is hard-wired to T=24 / "R=16" parameters
  (:63-75).  Here ``mask_T`` selects the variant: ``mask_T=24`` reproduces the live code bit for bit,
  ``mask_T=1`` (default) is the single-frame variant the reference keeps commented out (:72-73) with the
  (sw, sm, sa) set looked up from ``R`` (``MASK_PARAMS``); ``mask_params=`` overrides the set.
* ``mask_mode="uniform"`` is the LEGACY mask the reference keeps commented out (:50-61): ``rand(1, 1, W) <= 1 / R`` from torch's
  generator plus a fully sampled centre window of ``int(W * center_lines_frac)`` lines -- the only mode in which ``R`` and
  ``center_lines_frac`` act as the constructor's signature promises, for any R (bit-exact against tests/golden/g30).
* coil maps are kept float64 on the host (``.sens_maps``, as the reference) and float32 on the device.
"""
import warnings

import numpy as np
import torch

from . import LinearTransform, generate_mask, i2k_complex, k2i_complex, MASK_PARAMS
from .masking import SkipLines
from ... import ops


def _check_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a GPU tensor (no CPU fallback in this build)")


class UndersamplingFourier(LinearTransform):
    """every ``num_skip_lines``-th k-space ROW of the centred FFT (reference :10-36): S = P M F x, adjoint F^-1 M^T P^T"""

    def __init__(self, num_skip_lines, in_shape):
        self.skip_lines = SkipLines(num_skip_lines, in_shape)

    def __call__(self, X: torch.Tensor) -> torch.Tensor:
        _check_gpu(X, "UndersamplingFourier")
        return self.skip_lines(i2k_complex(X.to(torch.complex64))).contiguous()

    def conj_op(self, S: torch.Tensor) -> torch.Tensor:
        _check_gpu(S, "UndersamplingFourier.conj_op")
        return k2i_complex(self.skip_lines.conj_op(S.to(torch.complex64)))

    def projection(self, X: torch.Tensor, S: torch.Tensor, lamda: float) -> torch.Tensor:
        warnings.warn("Not used!")
        return X


class RandomUndersamplingFourier(LinearTransform):
    def __init__(self, R, center_lines_frac, in_shape, seed=None, mask_T=1, mask_params=None, mask_mode="variable"):
        """in_shape: (C, H, W); mask_mode "variable" (generate_mask, the live reference) or "uniform" (the legacy formula)"""
        if mask_mode not in ("variable", "uniform"):
            raise ValueError(f"mask_mode {mask_mode!r}: 'variable' or 'uniform'")
        self.R = R
        self.center_lines_frac = center_lines_frac
        self.in_shape = in_shape
        self.seed = seed
        self.mask_T = mask_T
        self.mask_params = mask_params
        self.mask_mode = mask_mode
        self.mask = self._generate_mask()
        self._dev = {}

    def _generate_uniform_mask(self):
        """reference :50-61 (commented out there): float (1, 1, W), torch's default generator seeded with `seed`"""
        torch.random.manual_seed(self.seed if self.seed is not None else torch.seed())
        W = self.in_shape[-1]
        mask = (torch.rand(1, 1, W) <= 1 / self.R).float()
        win_size = int(W * self.center_lines_frac)
        half_win_size = W // 2
        start_idx = half_win_size - win_size // 2
        end_idx = start_idx + win_size
        mask[..., start_idx:end_idx] = 1.
        return mask

    def _generate_mask(self):
        if self.mask_mode == "uniform":
            return self._generate_uniform_mask()
        torch.random.manual_seed(self.seed if self.seed is not None else torch.seed())
        W = self.in_shape[-1]
        if self.mask_params is not None:
            params = self.mask_params
        elif self.mask_T == 24:
            params = MASK_PARAMS[16]                      # the live reference ignores R
        elif self.R in MASK_PARAMS:
            params = MASK_PARAMS[self.R]
        else:
            raise ValueError(f"no variable-density mask parameters for R={self.R}; pass mask_params=dict(sw, sm, sa) or "
                             "mask_mode='uniform' (the legacy rand <= 1/R mask, any R)")
        mask = generate_mask(self.mask_T, W, seed=self.seed, **params)
        return mask.unsqueeze(1)                          # (1, 1, W) or (T, 1, 1, W)

    def mask_u8(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = ops._mask_u8(self.mask, self.in_shape[-1], device)
        return self._dev[key]

    def __call__(self, X: torch.Tensor) -> torch.Tensor:
        """S = mask * i2k_complex(X): one kernel (FFT in LDS, mask applied on the way out)"""
        _check_gpu(X, "RandomUndersamplingFourier")
        X = X.to(torch.complex64)
        return ops.sense_forward(X, None, self.mask_u8(X.device))[0]

    def conj_op(self, S: torch.Tensor) -> torch.Tensor:
        return k2i_complex(S)

    def projection(self, X: torch.Tensor, S: torch.Tensor, lamda: float) -> torch.Tensor:
        """k-space mix F^-1[lamda S + (1 - lamda) M F X + (1 - M) F X] (:89-97).  The reference's `(1 - mask)` raises for
        its own bool masks; the documented formula is applied, as one kernel."""
        _check_gpu(X, "RandomUndersamplingFourier.projection")
        zr = torch.view_as_real(X.to(torch.complex64))
        o_re, o_im = ops.singlecoil_prox(zr[..., 0].contiguous(), zr[..., 1].contiguous(),
                                         S.to(torch.complex64).contiguous(), self.mask_u8(X.device), float(lamda),
                                         ops.SC_PROJECTION)
        return torch.complex(o_re, o_im)


class SENSE(LinearTransform):
    def __init__(self, sens_type, num_sens, R, center_lines_frac, in_shape, seed, mask_T=1, mask_params=None,
                 mask_mode="variable"):
        assert sens_type in ["exp"]
        self.random_under_fourier = RandomUndersamplingFourier(R, center_lines_frac, in_shape, seed, mask_T,
                                                               mask_params, mask_mode)
        maps = []
        for i in range(num_sens):
            s = self.random_under_fourier.seed
            maps.append(self._generate_sens_map(sens_type, None if s is None else s + i))
        maps = torch.stack(maps, dim=0)                                     # (num_sens, H, W) float64
        self.sens_maps = maps / torch.sqrt((torch.abs(maps) ** 2).sum(dim=0))
        energy = (torch.abs(self.sens_maps) ** 2).sum(dim=0)
        assert torch.allclose(energy, torch.ones_like(energy))
        self._dev = {}

    def _generate_sens_map(self, sens_type, seed=0, **kwargs):
        """exp(-dist / (2 l)) around a random anchor, l = max(dist) / 2.  The reference builds the pixel list
        from np.mgrid[0:W, 0:H] and reshapes the distances to (H, W) (:131-134); kept literally."""
        H, W = self.random_under_fourier.in_shape[-2:]
        anchor = kwargs.get("anchor", None)
        if anchor is None:
            np.random.seed(seed)
            anchor = np.array([np.random.choice(H), np.random.choice(W)])
        ww, hh = np.mgrid[0:W, 0:H]
        dist = np.sqrt((ww.ravel() - anchor[0]).astype(np.float64) ** 2 +
                       (hh.ravel() - anchor[1]).astype(np.float64) ** 2)
        length = kwargs.get("l", dist.max() / 2)
        return torch.exp(-torch.tensor(dist.reshape(H, W)) / (2 * length))

    # device-side cached copies (the reference re-uploads on every call, :143,154)
    def sens_f32(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = self.sens_maps.to(torch.float32).to(device).contiguous()
        return self._dev[key]

    def mask_u8(self, device):
        return self.random_under_fourier.mask_u8(device)

    def __call__(self, X: torch.Tensor) -> torch.Tensor:
        """X (B, C, H, W) complex -> (num_sens, B, C, H, W) complex64"""
        _check_gpu(X, "SENSE")
        X = X.to(torch.complex64)
        return ops.sense_forward(X, self.sens_f32(X.device), self.mask_u8(X.device))

    def conj_op(self, S: torch.Tensor) -> torch.Tensor:
        _check_gpu(S, "SENSE.conj_op")
        S = S.to(torch.complex64)
        return ops.sense_adjoint(S, self.sens_f32(S.device))

    def SSOS(self, S: torch.Tensor) -> torch.Tensor:
        _check_gpu(S, "SENSE.SSOS")
        return ops.sense_ssos(S.to(torch.complex64))

    def projection(self, X: torch.Tensor, S: torch.Tensor, lamda: float) -> torch.Tensor:
        warnings.warn("Not implemented!")
        return X
