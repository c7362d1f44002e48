"""Noise schedules (mirror of the reference's ``ncsn/models/__init__.py:10-38``)."""
import numpy as np
import torch


def _schedule(section):
    if section.sigma_dist == "geometric":
        s = np.exp(np.linspace(np.log(section.sigma_begin), np.log(section.sigma_end), section.num_classes))
    elif section.sigma_dist == "uniform":
        s = np.linspace(section.sigma_begin, section.sigma_end, section.num_classes)
    else:
        raise NotImplementedError("sigma distribution not supported")
    return torch.tensor(s).float()          # float64 -> float32 rounding as in the reference


def get_sigmas(config, mode="unconditioned"):
    assert mode in ("unconditioned", "recons")
    section = config.recons if mode == "recons" else config.model
    return _schedule(section).to(config.device)
