#!/usr/bin/env python3
"""Round-4 golden fixtures (same rules as make_golden.py: runs ONLY in the build container, imports the Python reference from
/root/reference on CPU, stores DATA only -- inputs and the reference's own outputs).

    python tests/golden/make_golden_r4.py

  g30_legacy_operators
      skip_*     SkipLines (ncsn/linear_transforms/masking.py:6-44): __call__, conj_op and projection of the REFERENCE class for
                 num_skip_lines 2 and 3 on a (2, 1, 12, 10) complex input
      uf_*       UndersamplingFourier (undersampling_fourier.py:10-36): __call__ and conj_op of the REFERENCE class
      umask_*    the legacy uniform mask of RandomUndersamplingFourier (undersampling_fourier.py:50-61).  The reference keeps this
                 method body COMMENTED OUT (the live _generate_mask ignores R), so it cannot be called; the fixture is that formula
                 evaluated here, statement by statement, with torch's generator (torch.random.manual_seed(seed);
                 torch.rand(1, 1, W) <= 1 / R; the centre window set to one) for (R, center_lines_frac, W, seed) in UMASK_CASES
                 -- it pins the draw order and window arithmetic against torch-version drift.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (performs the stubbed import of the reference)
from make_golden import np, torch, npy, save  # noqa: E402

from InverseProblemWithDiffusionModel.ncsn.linear_transforms.masking import SkipLines as RefSkipLines  # noqa: E402
from InverseProblemWithDiffusionModel.ncsn.linear_transforms.undersampling_fourier import UndersamplingFourier as RefUF  # noqa: E402

UMASK_CASES = [(4, 0.08, 128, 0), (8, 0.04, 128, 7), (20, 0.04, 128, 0), (40, 0.04, 128, 3), (3, 0.1, 50, 11), (40, 0.0, 64, 5)]


def legacy_uniform_mask(R, center_lines_frac, W, seed):
    """undersampling_fourier.py:50-61, the commented-out body, verbatim in its arithmetic"""
    torch.random.manual_seed(seed)
    mask = (torch.rand(1, 1, W) <= 1 / R).float()
    win_size = int(W * center_lines_frac)
    half_win_size = W // 2
    start_idx = half_win_size - win_size // 2
    end_idx = start_idx + win_size
    mask[..., start_idx:end_idx] = 1.
    return mask


def g30():
    out = {}
    gen = torch.Generator().manual_seed(30)
    X = torch.complex(torch.randn(2, 1, 12, 10, generator=gen), torch.randn(2, 1, 12, 10, generator=gen))
    out["x"] = npy(X)
    for n in (2, 3):
        op = RefSkipLines(n, (1, 12, 10))
        S = op(X)
        out[f"skip{n}_y"] = npy(S)
        out[f"skip{n}_adj"] = npy(op.conj_op(S))
        S2 = torch.complex(torch.randn(S.shape, generator=gen), torch.randn(S.shape, generator=gen))
        out[f"skip{n}_s"] = npy(S2)
        out[f"skip{n}_proj"] = npy(op.projection(X, S2, 0.3))
        uf = RefUF(n, (1, 12, 10))
        Y = uf(X)
        out[f"uf{n}_y"] = npy(Y)
        out[f"uf{n}_adj"] = npy(uf.conj_op(Y))
    for i, (R, frac, W, seed) in enumerate(UMASK_CASES):
        out[f"umask_{i}"] = npy(legacy_uniform_mask(R, frac, W, seed))
    out["umask_cases"] = np.array(UMASK_CASES, dtype=np.float64)
    save("g30_legacy_operators", **out)


if __name__ == "__main__":
    g30()
