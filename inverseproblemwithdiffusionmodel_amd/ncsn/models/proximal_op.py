"""Data-consistency proximal operators (mirror of the reference's ``ncsn/models/proximal_op.py``).

``L2Penalty`` in the reference is ONE SGD(lr=0.05) step through autograd on
``0.5|x-z|^2.mean + 0.5 (alpha/lamda) |Ax-y|^2.sum(1,2,3).mean`` starting at x = z (:19-51).  Its update is
the closed form  x = z - 0.05 (alpha/lamda) A^H(A z - y) / K  with K = num_sens * W for SENSE (the .mean()
runs over the (num_sens, W) axes that .sum(dim=(1,2,3)) leaves) and K = B for a single-coil operator
(pinned by tests/golden/g05_prox.npz).  That closed form runs as one fused HIP kernel."""
import torch

from ..linear_transforms import LinearTransform
from ..linear_transforms.undersampling_fourier import RandomUndersamplingFourier, SENSE
from ... import ops

SGD_LR = 5e-2


def _singlecoil(lin_tfm, z, y, coef, mode):
    """z, y (B, C, H, W) complex -> the single-coil operator `mode` of ipdm_singlecoil_prox_f32"""
    zr = torch.view_as_real(z.to(torch.complex64))
    o_re, o_im = ops.singlecoil_prox(zr[..., 0].contiguous(), zr[..., 1].contiguous(),
                                     y.to(torch.complex64).contiguous(), lin_tfm.mask_u8(z.device), coef, mode)
    return torch.complex(o_re, o_im)


class Proximal(object):
    def __init__(self, lin_tfm: LinearTransform):
        self.lin_tfm = lin_tfm

    def __call__(self, *args, **kwargs):
        pass


class L2Penalty(Proximal):
    def coef(self, alpha, lamda, z_shape):
        if isinstance(self.lin_tfm, SENSE):
            K = self.lin_tfm.sens_maps.shape[0] * z_shape[-1]
        else:
            K = z_shape[0]
        return SGD_LR * (alpha / lamda) / K

    def __call__(self, z, y, alpha, lamda, num_steps=1):
        """x <- one gradient step on 1/2 |x - z|^2 + 1/2 alpha/lamda |Ax - y|^2 from x = z"""
        if num_steps != 1:
            raise NotImplementedError("the closed form covers the reference's num_steps=1 only")
        if not z.is_cuda:
            raise RuntimeError("L2Penalty: expected GPU tensors (no CPU fallback in this build)")
        c = self.coef(float(alpha), float(lamda), z.shape)
        z = z.to(torch.complex64)
        if isinstance(self.lin_tfm, SENSE):
            zr = torch.view_as_real(z)
            o_re, o_im = ops.sense_l2prox(zr[..., 0].contiguous(), zr[..., 1].contiguous(), y,
                                          self.lin_tfm.sens_f32(z.device), self.lin_tfm.mask_u8(z.device), c)
            return torch.complex(o_re, o_im)
        if isinstance(self.lin_tfm, RandomUndersamplingFourier):
            return _singlecoil(self.lin_tfm, z, y, c, ops.SC_L2PENALTY)
        raise NotImplementedError(f"L2Penalty: no kernel chain for {type(self.lin_tfm).__name__}")

    @torch.no_grad()
    def check_solution(self, x_sol, z, y, alpha, lamda):
        b = z + alpha / lamda * self.lin_tfm.conj_op(y)
        lhs = x_sol + alpha / lamda * self.lin_tfm.conj_op(self.lin_tfm(x_sol))
        return (torch.abs(lhs - b) ** 2).sum(dim=(1, 2, 3)).mean()


class Constrained(Proximal):
    """Proximal operator from Yang et al (MRI)."""

    def __call__(self, X: torch.Tensor, S: torch.Tensor, lamda: float):
        return self.lin_tfm.projection(X, S, lamda)


class SingleCoil(Proximal):
    def __init__(self, lin_tfm: RandomUndersamplingFourier):
        super(SingleCoil, self).__init__(lin_tfm)
        assert isinstance(self.lin_tfm, RandomUndersamplingFourier), "only supporting RandomUnversamplingFourier"

    def __call__(self, z, y, alpha, lamda):
        """closed form  x = F' diag(1 / (1 + alpha M)) F (z + alpha F' y), one kernel (two LDS-resident FFTs)"""
        if not z.is_cuda:
            raise RuntimeError("SingleCoil: expected GPU tensors (no CPU fallback in this build)")
        return _singlecoil(self.lin_tfm, z, y, self.coef(float(alpha), float(lamda), z.shape), ops.SC_CLOSED_FORM)

    def coef(self, alpha, lamda, z_shape=None):
        return alpha / lamda

    @torch.no_grad()
    def check_solution(self, x_out, z, y, alpha, lamda):
        alpha = alpha / lamda
        lhs = x_out + alpha * self.lin_tfm.conj_op(self.lin_tfm(x_out))
        rhs = alpha * self.lin_tfm.conj_op(y) + z
        return (torch.abs(lhs - rhs) ** 2).sum(dim=(1, 2, 3)).mean()


def get_proximal(proximal_name: str):
    assert proximal_name in ["L2Penalty", "Constrained", "SingleCoil"]
    return {"L2Penalty": L2Penalty, "Constrained": Constrained, "SingleCoil": SingleCoil}[proximal_name]
