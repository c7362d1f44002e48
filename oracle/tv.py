"""Total variation of the TV baseline, CPU restatement (oracle; test infrastructure only).

Reference anchors: scripts/acdc_SENSE_TV.py:76 (reg = kornia.losses.TotalVariation()), ncsn/models/MAP_optimizers.py:41-48
(MAPModel.forward: loss = |A X - S|^2 / 2 + reg_weight * reg(X)), helpers/pl_helpers.py:436-437 (Adam(lr)).
kornia is not installed here and is not vendored by the reference: `total_variation` restates kornia's published
definition (kornia/losses/total_variation.py: sum over the image of |x[1:, :] - x[:-1, :]| plus |x[:, 1:] - x[:, :-1]|,
reduction 'sum') -- parity UNPINNED against kornia itself; the gradient the kernels must produce is pinned to torch
autograd of that definition (tests), the optimiser to torch.optim.Adam."""
import torch


def total_variation(x):
    """x (..., H, W) real or complex tensor -> (...) sum of |first differences| along H and along W"""
    d1 = x[..., 1:, :] - x[..., :-1, :]
    d2 = x[..., :, 1:] - x[..., :, :-1]
    return d1.abs().sum(dim=(-2, -1)) + d2.abs().sum(dim=(-2, -1))


def tv_map(measurement, forward, adjoint, reg_weight, lr, num_epochs):
    """MAPModel + TrainMAPModel on CPU with autograd and torch.optim.Adam: X0 = A^H S"""
    X = torch.nn.Parameter(adjoint(measurement).clone())
    opt = torch.optim.Adam([X], lr=lr)
    for _ in range(num_epochs):
        opt.zero_grad()
        loss = (torch.abs(forward(X) - measurement) ** 2).sum() / 2 + reg_weight * total_variation(X).sum()
        loss.backward()
        opt.step()
    return X.detach()
