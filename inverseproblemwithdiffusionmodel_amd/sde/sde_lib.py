"""Forward / reverse SDEs (mirror of the reference's ``sde/sde_lib.py``: SDE :7-109, VPSDE :112-164,
subVPSDE :167-204, VESDE :207-255).  Per-sample coefficients are (B,)-sized host-style tensor math; every
image-sized product or sum goes through the libipdm.so per-sample axpy kernels when the state lives on a GPU."""
import abc

import numpy as np
import torch

from .. import ops


def _bcast(v):
    return v[:, None, None, None]


def scale_samples(x, a):
    """a[:, None, None, None] * x   (GPU tensors only: no CPU fallback)"""
    return ops.sample_axpy2(torch.zeros_like(x), x, a)


def axpy_samples(x, y, a):
    """x + a[:, None, None, None] * y   (GPU tensors only: no CPU fallback)"""
    return ops.sample_axpy2(x, y, a)


class SDE(abc.ABC):
    """SDE abstract class. Functions are designed for a mini-batch of inputs."""

    def __init__(self, N):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        pass

    @abc.abstractmethod
    def sde(self, x, t):
        pass

    @abc.abstractmethod
    def marginal_prob(self, x, t):
        pass

    @abc.abstractmethod
    def prior_sampling(self, shape):
        pass

    @abc.abstractmethod
    def prior_logp(self, z):
        pass

    def discretize(self, x, t):
        """x_{i+1} = x_i + f_i(x_i) + G_i z_i; Euler-Maruyama by default"""
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t)
        f = scale_samples(drift, torch.full_like(t, dt))
        G = diffusion * torch.sqrt(torch.tensor(dt, device=t.device))
        return f, G

    def reverse(self, score_fn, probability_flow=False):
        """the reverse-time SDE / probability-flow ODE as an object of the same class"""
        N, T = self.N, self.T
        sde_fn, discretize_fn = self.sde, self.discretize
        half = 0.5 if probability_flow else 1.

        class RSDE(self.__class__):
            def __init__(self):
                self.N = N
                self.probability_flow = probability_flow

            @property
            def T(self):
                return T

            def sde(self, x, t):
                drift, diffusion = sde_fn(x, t)
                drift = axpy_samples(drift, score_fn(x, t), -(diffusion ** 2) * half)
                return drift, (0. if self.probability_flow else diffusion)

            def discretize(self, x, t):
                f, G = discretize_fn(x, t)
                rev_f = axpy_samples(f, score_fn(x, t), -(G ** 2) * half)
                return rev_f, (torch.zeros_like(G) if self.probability_flow else G)

        return RSDE()


class VPSDE(SDE):
    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0, self.beta_1, self.N = beta_min, beta_max, N
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1. - self.discrete_betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1. - self.alphas_cumprod)

    @property
    def T(self):
        return 1

    def sde(self, x, t):
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        return scale_samples(x, -0.5 * beta_t), torch.sqrt(beta_t)

    def _log_mean_coeff(self, t):
        return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0

    def marginal_prob(self, x, t):
        c = self._log_mean_coeff(t)
        return scale_samples(x, torch.exp(c)), torch.sqrt(1. - torch.exp(2. * c))

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def prior_logp(self, z):
        N = np.prod(z.shape[1:])
        return -N / 2. * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.

    def discretize(self, x, t):
        timestep = (t * (self.N - 1) / self.T).long()
        beta = self.discrete_betas.to(x.device)[timestep]
        alpha = self.alphas.to(x.device)[timestep]
        return scale_samples(x, torch.sqrt(alpha) - 1.), torch.sqrt(beta)


class subVPSDE(SDE):
    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0, self.beta_1, self.N = beta_min, beta_max, N

    @property
    def T(self):
        return 1

    def sde(self, x, t):
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        discount = 1. - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t ** 2)
        return scale_samples(x, -0.5 * beta_t), torch.sqrt(beta_t * discount)

    def marginal_prob(self, x, t):
        c = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return scale_samples(x, torch.exp(c)), 1 - torch.exp(2. * c)

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def prior_logp(self, z):
        N = np.prod(z.shape[1:])
        return -N / 2. * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.


class VESDE(SDE):
    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        super().__init__(N)
        self.sigma_min, self.sigma_max, self.N = sigma_min, sigma_max, N
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(self.sigma_min), np.log(self.sigma_max), N))

    @property
    def T(self):
        return 1

    def marginal_prob_std(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def sde(self, x, t):
        sigma = self.marginal_prob_std(t)
        diffusion = sigma * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min)),
                                                    device=t.device))
        return torch.zeros_like(x), diffusion

    def marginal_prob(self, x, t):
        return x, self.marginal_prob_std(t)

    def prior_sampling(self, shape):
        return torch.randn(*shape) * self.sigma_max

    def prior_logp(self, z):
        N = np.prod(z.shape[1:])
        return -N / 2. * np.log(2 * np.pi * self.sigma_max ** 2) - torch.sum(z ** 2, dim=(1, 2, 3)) / (2 * self.sigma_max ** 2)

    def discretize(self, x, t):
        timestep = (t * (self.N - 1) / self.T).long()
        sigmas = self.discrete_sigmas.to(t.device)
        sigma = sigmas[timestep]
        adjacent_sigma = torch.where(timestep == 0, torch.zeros_like(t), sigmas[timestep - 1])
        return torch.zeros_like(x), torch.sqrt(sigma ** 2 - adjacent_sigma ** 2)
