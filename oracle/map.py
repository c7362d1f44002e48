"""ORACLE -- test infrastructure only (see oracle/__init__.py).  CPU restatement of the reference's MAP baseline
`MAPOptimizer` / `SENSEMAP` (ncsn/models/MAP_optimizers.py:55-116): gradient ascent on the log posterior with
torch.optim.Adam(betas=(0.5, 0.5)) semantics, restated (no torch.optim): per iteration
    grad = -A^H(A x - y) + lamda * (s(Re x, 1) + i s(Im x, 1)),   param.grad = -grad          (:90-105)
Pinned by tests/golden/g18_map.npz (the reference's own SENSEMAP on the tiny score net)."""
import numpy as np
import torch


def adam_ascent_step(x, grad, m, v, lr, step, b1=0.5, b2=0.5, eps=1e-8):
    """one Adam step on float32 arrays with param.grad = -grad (torch/optim/adam.py single-tensor form)"""
    g = (-grad).astype(np.float32)
    m += (g - m) * np.float32(1.0 - b1)
    v *= np.float32(b2)
    v += np.float32(1.0 - b2) * (g * g)
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)
    x -= np.float32(lr / bc1) * (m / denom)
    return x


def sense_map(x_init, measurement, score_fn, op_forward, op_adjoint, lamda, lr, n_iters):
    """x_init complex64 (B,1,H,W); score_fn(real float32 (B,1,H,W), labels) -> same shape (numpy in / out);
    op_forward / op_adjoint: the SENSE operator and its adjoint on complex64 numpy arrays"""
    x = np.array(x_init, dtype=np.complex64)
    xr = np.ascontiguousarray(np.stack([x.real, x.imag], axis=-1)).astype(np.float32)      # view_as_real layout
    m, v = np.zeros_like(xr), np.zeros_like(xr)
    labels = np.ones(x.shape[0], dtype=np.int64)
    for it in range(1, n_iters + 1):
        xc = (xr[..., 0] + 1j * xr[..., 1]).astype(np.complex64)
        grad_data = -op_adjoint(op_forward(xc) - measurement)
        grad_prior = score_fn(np.ascontiguousarray(xc.real), labels) + 1j * score_fn(np.ascontiguousarray(xc.imag), labels)
        grad = (grad_data + lamda * grad_prior).astype(np.complex64)
        gr = np.stack([grad.real, grad.imag], axis=-1).astype(np.float32)
        adam_ascent_step(xr, gr, m, v, lr, it)
    return (xr[..., 0] + 1j * xr[..., 1]).astype(np.complex64)
