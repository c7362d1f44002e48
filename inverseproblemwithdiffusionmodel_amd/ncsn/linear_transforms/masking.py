"""Line-skipping sampling operator (mirror of the reference's ``ncsn/linear_transforms/masking.py:6-44``).

``A = P M``: keep every ``num_skip_lines``-th ROW of a (B, C, H, W) tensor.  Pure data movement (strided views and
index copies on whatever device the tensor lives on): there is no arithmetic to put into a kernel except ``projection``'s
convex mix of the retained rows, which torch's elementwise kernels do on the strided view."""
import torch

from . import LinearTransform


class SkipLines(LinearTransform):
    def __init__(self, num_skip_lines, in_shape):
        super().__init__()
        self.num_skip_lines = num_skip_lines
        self.in_shape = in_shape

    def __call__(self, X: torch.Tensor) -> torch.Tensor:
        """X (B, C, H, W) -> (B, C, ceil(H / n), W): rows 0, n, 2n, ..."""
        return X[:, :, 0::self.num_skip_lines, :]

    def conj_op(self, S: torch.Tensor) -> torch.Tensor:
        """zero-filled adjoint: (B, C, H', W) -> (B, *in_shape)"""
        out = torch.zeros((S.shape[0], *self.in_shape), dtype=S.dtype, device=S.device)
        out[:, :, 0::self.num_skip_lines] = S
        return out

    def projection(self, X: torch.Tensor, S: torch.Tensor, lamda: float) -> torch.Tensor:
        """x <- lamda * s + (1 - lamda) * M x on the retained rows, x on the others (masking.py:29-44)"""
        out = X.clone()
        out[:, :, 0::self.num_skip_lines, :] = lamda * S + (1 - lamda) * self(X)
        return out
