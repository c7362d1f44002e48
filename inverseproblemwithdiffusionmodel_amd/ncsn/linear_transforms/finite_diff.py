"""Periodic finite differences along one axis and the temporal-TV sub-gradient (mirror of the reference's
``ncsn/linear_transforms/finite_diff.py:7-35``).  Pure shifts and subtractions on device tensors."""
from typing import Tuple, Union

import torch

from . import LinearTransform


class FiniteDiff(LinearTransform):
    def __init__(self, dims: Union[int, Tuple[int]]):
        self.dims = dims

    def __call__(self, X: torch.Tensor) -> torch.Tensor:
        return torch.roll(X, -1, self.dims) - X

    def conj_op(self, S: torch.Tensor) -> torch.Tensor:
        return torch.roll(S, 1, self.dims) - S

    def projection(self, X, S, lamda):
        return X

    def log_lh_grad(self, X: torch.Tensor, S: torch.Tensor = None, lamda: float = 1) -> torch.Tensor:
        """grad = -lamda * nabla' sign(nabla X)"""
        return -lamda * self.conj_op(torch.sign(self(X)))
