"""Config and tensor helpers on the hot path (mirror of the reference's ``helpers/utils.py``:
load_yml_file :173-180, dict2namespace :183-191, data_transform :212-226, reshape_temporal_dim :330-359).
Visualisation helpers of the reference are no-ops here (they are host-side PNG dumps)."""
import argparse

import einops
import torch
import yaml


def load_yml_file(filename: str):
    assert ".yml" in filename
    with open(filename, "r") as rf:
        data = yaml.load(rf, yaml.Loader)
    return dict2namespace(data)


def dict2namespace(config):
    namespace = argparse.Namespace()
    for key, value in config.items():
        setattr(namespace, key, dict2namespace(value) if isinstance(value, dict) else value)
    return namespace


def logit_transform(image, lam=1e-6):
    image = lam + (1 - 2 * lam) * image
    return torch.log(image) - torch.log1p(-image)


def data_transform(config, X):
    if config.data.uniform_dequantization:
        X = X / 256. * 255. + torch.rand_like(X) / 256.
    if config.data.gaussian_dequantization:
        X = X + torch.randn_like(X) * 0.01
    if config.data.rescaled:
        X = 2 * X - 1.
    elif config.data.logit_transform:
        X = logit_transform(X)
    if hasattr(config, 'image_mean'):
        return X - config.image_mean.to(X.device)[None, ...]
    return X


def reshape_temporal_dim(x: torch.Tensor, kx, ky, direction="forward", img_size=None):
    """forward: (N, T, H, W) -> (N*H*W/(kx*ky), kx*ky, T); backward is the inverse (pure layout)."""
    assert direction in ["forward", "backward"]
    if direction == "forward":
        N, T, H, W = x.shape
        assert H % kx == 0 and W % ky == 0
        return einops.rearrange(x, "N T (H1 kx) (W1 ky) -> (N H1 W1) (kx ky) T", kx=kx, ky=ky)
    assert img_size is not None
    H, W = img_size
    assert H % kx == 0 and W % ky == 0
    assert x.shape[1] == kx * ky
    return einops.rearrange(x, "(N H1 W1) (kx ky) T -> N T (H1 kx) (W1 ky)", H1=H // kx, W1=W // ky, kx=kx, ky=ky)


def vis_images(*args, **kwargs):
    return None


def vis_multi_channel_signal(*args, **kwargs):
    return None


def undersample_seg_mask(label: torch.Tensor, fraction=1., seed=None):
    """keep a random `fraction` of the labelled pixels (mirror of helpers/utils.py:314-327; same torch RNG calls)"""
    assert 0. <= fraction <= 1.
    if seed is not None:
        torch.random.manual_seed(seed)
    non_zeros = torch.nonzero(label.cpu(), as_tuple=True)
    num_samples = max(1, int(non_zeros[0].shape[0] * fraction))
    sample_indices = torch.randperm(non_zeros[0].shape[0])[:num_samples]
    indices = [ind[sample_indices] for ind in non_zeros]
    label_out = torch.zeros_like(label.cpu())
    label_out[indices] = 1
    return label_out.to(label.device)
