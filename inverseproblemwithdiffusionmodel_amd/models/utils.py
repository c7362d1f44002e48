"""score_sde model utilities (mirror of the reference's ``models/utils.py``: registry :27-47, get_sigmas :50-60,
create_model :88-94, get_model_fn :97-126, get_score_fn :129-178, flatten helpers :181-188)."""
import numpy as np
import torch

from ..sde import sde_lib

_MODELS = {}


def register_model(cls=None, *, name=None):
    """decorator registering a model class under `name` (default: the class name)"""

    def _register(cls):
        local_name = cls.__name__ if name is None else name
        if local_name in _MODELS:
            raise ValueError(f'Already registered model with name: {local_name}')
        _MODELS[local_name] = cls
        return cls

    return _register if cls is None else _register(cls)


def get_model(name):
    return _MODELS[name]


def get_sigmas(config):
    """SMLD noise levels: geometric sigma_max -> sigma_min in num_scales steps (float64 numpy)"""
    return np.exp(np.linspace(np.log(config.model.sigma_max), np.log(config.model.sigma_min), config.model.num_scales))


def create_model(config):
    """the reference wraps the model in nn.DataParallel (:93); here one process drives one GPU and samples are
    sharded across processes, so the bare module is returned (``.module`` kept for code that unwraps it)"""
    score_model = get_model(config.model.name)(config).to(config.device)
    score_model.module = score_model
    return score_model


def get_model_fn(model, train=False):
    if train:
        raise NotImplementedError("training is outside the sampling hot path")

    def model_fn(x, labels):
        model.eval()
        return model(x, labels)

    return model_fn


def get_score_fn(sde, model, train=False, continuous=False):
    """wrap the model output into a time-dependent score function"""
    model_fn = get_model_fn(model, train=train)
    if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
        def score_fn(x, t):
            if continuous or isinstance(sde, sde_lib.subVPSDE):
                labels = t * 999
                score = model_fn(x, labels)
                std = sde.marginal_prob(torch.zeros_like(x), t)[1]
            else:
                labels = t * (sde.N - 1)
                score = model_fn(x, labels)
                std = sde.sqrt_1m_alphas_cumprod.to(labels.device)[labels.long()]
            return -score / std[:, None, None, None]
    elif isinstance(sde, sde_lib.VESDE):
        def score_fn(x, t):
            if continuous:
                labels = sde.marginal_prob_std(t)
            else:
                labels = torch.round((sde.T - t) * (sde.N - 1)).long()
            return model_fn(x, labels)
    else:
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
    return score_fn


def to_flattened_numpy(x):
    return x.detach().cpu().numpy().reshape((-1,))


def from_flattened_numpy(x, shape):
    return torch.from_numpy(x.reshape(shape))
