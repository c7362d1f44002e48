"""Predictor-corrector / ODE samplers of the score_sde API (mirror of the reference's ``sde/sampling.py``:
registries :34-81, get_sampling_fn :84-127, Predictor/Corrector :130-178, predictors :181-255, correctors
:258-335, shared_*_update_fn :338-357, get_pc_sampler :360-416, get_ode_sampler :419-490, vanilla_pc_sampler
:493-530).  Image-sized updates run as per-sample axpy kernels, norms as a per-sample reduction kernel and the
Gaussian noise comes from the Philox kernel (``set_noise_source`` injects a recorded stream for parity runs)."""
import abc
import functools

import numpy as np
import torch
from scipy import integrate

from . import sde_lib
from .sde_lib import axpy_samples
from .. import ops, sharding
from ..models import utils as mutils
from ..models.utils import from_flattened_numpy, to_flattened_numpy, get_score_fn

_CORRECTORS = {}
_PREDICTORS = {}


class _Noise:
    """N(0,1) draws shaped like x: Philox kernel keyed by (seed, GLOBAL sample id, call counter) on GPUs"""

    def __init__(self, seed=0):
        self.seed, self.count, self.fn, self.sample_offset = seed, 0, None, 0

    def __call__(self, x):
        if self.fn is not None:
            return self.fn(x).to(x.device)
        self.count += 1
        return ops.philox_normal(tuple(x.shape), x.device, seed=self.seed, sample_offset=self.sample_offset, step_id=self.count)


noise_like = _Noise()
_SHARD = {"total": None, "lo": 0}      # set_shard(): this process holds samples [lo, lo + B) of a batch of `total`


def set_noise_source(fn=None, seed=0, sample_offset=0):
    """fn(like) -> tensor replaces the generator (None: Philox with `seed`; sample_offset: first global sample id of
    this process' block, so that a sample's noise does not depend on how the batch is sharded)"""
    noise_like.fn, noise_like.seed, noise_like.count, noise_like.sample_offset = fn, seed, 0, sample_offset


def set_shard(total=None, lo=0):
    """a batch sharded over the ranks of torch.distributed: LangevinCorrector's step size couples the batch through two
    means over per-sample norms (sde/sampling.py:281-283 of the reference); with set_shard(total, lo) those means are taken
    over ALL `total` samples -- every rank adds its per-sample norms into its rows of a zero vector, one all-reduce(SUM)
    (exact: the other ranks add zeros), then the same .mean() a single process computes: bit-identical to the unsharded run."""
    _SHARD["total"], _SHARD["lo"] = total, lo


def _sample_norm_mean(v):
    norms = ops.sample_norm(v)
    total = _SHARD["total"]
    if total is not None and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        buf = torch.zeros(total, dtype=norms.dtype, device=norms.device)
        buf[_SHARD["lo"]:_SHARD["lo"] + norms.shape[0]] = norms
        sharding.all_reduce(buf)
        return buf.mean()
    return norms.mean()


def register_predictor(cls=None, *, name=None):
    def _register(cls):
        local_name = cls.__name__ if name is None else name
        if local_name in _PREDICTORS:
            raise ValueError(f'Already registered model with name: {local_name}')
        _PREDICTORS[local_name] = cls
        return cls
    return _register if cls is None else _register(cls)


def register_corrector(cls=None, *, name=None):
    def _register(cls):
        local_name = cls.__name__ if name is None else name
        if local_name in _CORRECTORS:
            raise ValueError(f'Already registered model with name: {local_name}')
        _CORRECTORS[local_name] = cls
        return cls
    return _register if cls is None else _register(cls)


def get_predictor(name):
    return _PREDICTORS[name]


def get_corrector(name):
    return _CORRECTORS[name]


def get_sampling_fn(config, sde, shape, inverse_scaler, eps):
    sampler_name = config.sampling.method
    if sampler_name.lower() == 'ode':
        return get_ode_sampler(sde=sde, shape=shape, inverse_scaler=inverse_scaler,
                               denoise=config.sampling.noise_removal, eps=eps, device=config.device)
    if sampler_name.lower() == 'pc':
        return get_pc_sampler(sde=sde, shape=shape, predictor=get_predictor(config.sampling.predictor.lower()),
                              corrector=get_corrector(config.sampling.corrector.lower()),
                              inverse_scaler=inverse_scaler, snr=config.sampling.snr,
                              n_steps=config.sampling.n_steps_each,
                              probability_flow=config.sampling.probability_flow,
                              continuous=config.training.continuous, denoise=config.sampling.noise_removal, eps=eps,
                              device=config.device)
    raise ValueError(f"Sampler name {sampler_name} unknown.")


class Predictor(abc.ABC):
    """The abstract class for a predictor algorithm."""

    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__()
        self.sde = sde
        self.rsde = sde.reverse(score_fn, probability_flow)
        self.score_fn = score_fn

    @abc.abstractmethod
    def update_fn(self, x, t):
        pass


class Corrector(abc.ABC):
    """The abstract class for a corrector algorithm."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__()
        self.sde, self.score_fn, self.snr, self.n_steps = sde, score_fn, snr, n_steps

    @abc.abstractmethod
    def update_fn(self, x, t):
        pass


@register_predictor(name='euler_maruyama')
class EulerMaruyamaPredictor(Predictor):
    def update_fn(self, x, t):
        dt = -1. / self.rsde.N
        z = noise_like(x)
        drift, diffusion = self.rsde.sde(x, t)
        x_mean = axpy_samples(x, drift, torch.full_like(t, dt))
        return axpy_samples(x_mean, z, diffusion * np.sqrt(-dt)), x_mean


@register_predictor(name='reverse_diffusion')
class ReverseDiffusionPredictor(Predictor):
    def update_fn(self, x, t):
        f, G = self.rsde.discretize(x, t)
        z = noise_like(x)
        x_mean = axpy_samples(x, f, -torch.ones_like(t))
        return axpy_samples(x_mean, z, G), x_mean


@register_predictor(name='ancestral_sampling')
class AncestralSamplingPredictor(Predictor):
    """The ancestral sampling predictor. Currently only supports VE/VP SDEs."""

    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__(sde, score_fn, probability_flow)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
        assert not probability_flow, "Probability flow not supported by ancestral sampling"

    def vesde_update_fn(self, x, t):
        sde = self.sde
        timestep = (t * (sde.N - 1) / sde.T).long()
        sigmas = sde.discrete_sigmas.to(t.device)
        sigma = sigmas[timestep]
        adjacent_sigma = torch.where(timestep == 0, torch.zeros_like(t), sigmas[timestep - 1])
        x_mean = axpy_samples(x, self.score_fn(x, t), sigma ** 2 - adjacent_sigma ** 2)
        std = torch.sqrt((adjacent_sigma ** 2 * (sigma ** 2 - adjacent_sigma ** 2)) / (sigma ** 2))
        return axpy_samples(x_mean, noise_like(x), std), x_mean

    def vpsde_update_fn(self, x, t):
        sde = self.sde
        timestep = (t * (sde.N - 1) / sde.T).long()
        beta = sde.discrete_betas.to(t.device)[timestep]
        x_mean = sde_lib.scale_samples(axpy_samples(x, self.score_fn(x, t), beta), 1. / torch.sqrt(1. - beta))
        return axpy_samples(x_mean, noise_like(x), torch.sqrt(beta)), x_mean

    def update_fn(self, x, t):
        if isinstance(self.sde, sde_lib.VESDE):
            return self.vesde_update_fn(x, t)
        return self.vpsde_update_fn(x, t)


@register_predictor(name='none')
class NonePredictor(Predictor):
    """An empty predictor that does nothing."""

    def __init__(self, sde, score_fn, probability_flow=False):
        pass

    def update_fn(self, x, t):
        return x, x


def _alpha(sde, t):
    if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
        timestep = (t * (sde.N - 1) / sde.T).long()
        return sde.alphas.to(t.device)[timestep]
    return torch.ones_like(t)


@register_corrector(name='langevin')
class LangevinCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, t):
        """NB the step size couples the batch through the two .mean()s (sde/sampling.py:281-283); set_shard() makes them
        means over the whole sharded batch (one small all-reduce each)."""
        alpha = _alpha(self.sde, t)
        x_mean = x
        for _ in range(self.n_steps):
            grad = self.score_fn(x, t)
            noise = noise_like(x)
            grad_norm, noise_norm = _sample_norm_mean(grad), _sample_norm_mean(noise)
            step_size = (self.snr * noise_norm / grad_norm) ** 2 * 2 * alpha
            x_mean = axpy_samples(x, grad, step_size)
            x = axpy_samples(x_mean, noise, torch.sqrt(step_size * 2))
        return x, x_mean


@register_corrector(name='ald')
class AnnealedLangevinDynamics(Corrector):
    """The original annealed Langevin dynamics predictor in NCSN/NCSNv2."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, t):
        alpha = _alpha(self.sde, t)
        std = self.sde.marginal_prob(x, t)[1]
        x_mean = x
        for _ in range(self.n_steps):
            grad = self.score_fn(x, t)
            noise = noise_like(x)
            step_size = (self.snr * std) ** 2 * 2 * alpha
            x_mean = axpy_samples(x, grad, step_size)
            x = axpy_samples(x_mean, noise, torch.sqrt(step_size * 2))
        return x, x_mean


@register_corrector(name='none')
class NoneCorrector(Corrector):
    """An empty corrector that does nothing."""

    def __init__(self, sde, score_fn, snr, n_steps):
        pass

    def update_fn(self, x, t):
        return x, x


def shared_predictor_update_fn(x, t, sde, model, predictor, probability_flow, continuous):
    score_fn = mutils.get_score_fn(sde, model, train=False, continuous=continuous)
    obj = NonePredictor(sde, score_fn, probability_flow) if predictor is None else predictor(sde, score_fn, probability_flow)
    return obj.update_fn(x, t)


def shared_corrector_update_fn(x, t, sde, model, corrector, continuous, snr, n_steps):
    score_fn = mutils.get_score_fn(sde, model, train=False, continuous=continuous)
    obj = NoneCorrector(sde, score_fn, snr, n_steps) if corrector is None else corrector(sde, score_fn, snr, n_steps)
    return obj.update_fn(x, t)


def get_pc_sampler(sde, shape, predictor, corrector, inverse_scaler, snr, n_steps=1, probability_flow=False,
                   continuous=False, denoise=True, eps=1e-3, device='cuda'):
    predictor_update_fn = functools.partial(shared_predictor_update_fn, sde=sde, predictor=predictor,
                                            probability_flow=probability_flow, continuous=continuous)
    corrector_update_fn = functools.partial(shared_corrector_update_fn, sde=sde, corrector=corrector,
                                            continuous=continuous, snr=snr, n_steps=n_steps)

    def pc_sampler(model, n_iters=None, x_init=None):
        """-> (samples, number of function evaluations).  n_iters / x_init (extras) run a slice of the schedule."""
        with torch.no_grad():
            x = (sde.prior_sampling(shape) if x_init is None else x_init).to(device).float()
            timesteps = torch.linspace(sde.T, eps, sde.N, device=device).float()
            x_mean = x
            for i in range(sde.N if n_iters is None else n_iters):
                vec_t = torch.ones(shape[0], device=device).float() * timesteps[i]
                x, x_mean = corrector_update_fn(x.float(), vec_t, model=model)
                x, x_mean = predictor_update_fn(x.float(), vec_t, model=model)
            return inverse_scaler(x_mean if denoise else x), sde.N * (n_steps + 1)

    return pc_sampler


def get_ode_sampler(sde, shape, inverse_scaler, denoise=False, rtol=1e-5, atol=1e-5, method='RK45', eps=1e-3,
                    device='cuda'):
    """probability-flow ODE through scipy's black-box solver (host round trip per function evaluation, as the
    reference)"""

    def denoise_update_fn(model, x):
        score_fn = get_score_fn(sde, model, train=False, continuous=True)
        predictor_obj = ReverseDiffusionPredictor(sde, score_fn, probability_flow=False)
        vec_eps = torch.ones(x.shape[0], device=x.device) * eps
        _, x = predictor_obj.update_fn(x, vec_eps)
        return x

    def drift_fn(model, x, t):
        score_fn = get_score_fn(sde, model, train=False, continuous=True)
        return sde.reverse(score_fn, probability_flow=True).sde(x, t)[0]

    def ode_sampler(model, z=None):
        with torch.no_grad():
            x = sde.prior_sampling(shape).to(device) if z is None else z

            def ode_func(t, x):
                x = from_flattened_numpy(x, shape).to(device).type(torch.float32)
                vec_t = torch.ones(shape[0], device=x.device) * t
                return to_flattened_numpy(drift_fn(model, x, vec_t))

            solution = integrate.solve_ivp(ode_func, (sde.T, eps), to_flattened_numpy(x), rtol=rtol, atol=atol,
                                           method=method)
            x = torch.tensor(solution.y[:, -1]).reshape(shape).to(device).type(torch.float32)
            if denoise:
                x = denoise_update_fn(model, x)
            return inverse_scaler(x), solution.nfev

    return ode_sampler


@torch.no_grad()
def vanilla_pc_sampler(score_model, sde, snr, eps=1e-3, shape=(1, 1, 28, 28), save_dir=None, device='cuda', **kwargs):
    """reverse-diffusion predictor + one Langevin corrector step per time step, score_model(x, t) called directly"""
    predictor = ReverseDiffusionPredictor(sde, score_model)
    corrector = LangevinCorrector(sde, score_model, snr, n_steps=1)
    timesteps = torch.linspace(sde.T, eps, sde.N, device=device)
    x = sde.prior_sampling(shape).to(device)
    x_mean = x
    for i in range(sde.N):
        vec_t = torch.ones(shape[0], device=device) * timesteps[i]
        x, x_mean = corrector.update_fn(x, vec_t)
        x, x_mean = predictor.update_fn(x, vec_t)
    return x_mean
