"""score_sde flavour of the NCSNv2 family (mirror of the reference's ``models/ncsnv2.py``: get_network :31-40,
NCSNv2 :43-132 'ncsnv2_64', NCSNv2_128 :221-312, NCSNv2_256 :315-416).  Architecturally these are
ncsn.models.ncsnv2.{NCSNv2, NCSNv2Deeper, NCSNv2Deepest}; only the config keys differ (``data.centered``,
``data.channels``, ``model.nf``, ``model.num_scales``, ``model.sigma_max/min``), so the classes below adapt the
config and reuse the same modules and kernels (same state-dict keys)."""
import functools
from argparse import Namespace

import torch

from . import utils
from ..ncsn.models import ncsnv2 as _ncsn


def _adapt(config):
    """score_sde ConfigDict -> the Namespace layout the ncsn-side constructors read"""
    ns = Namespace(
        device=getattr(config, "device", torch.device("cpu")),
        data=Namespace(channels=config.data.channels if "channels" in config.data else config.data.num_channels,
                       image_size=config.data.image_size, logit_transform=False, rescaled=bool(config.data.centered)),
        model=Namespace(ngf=config.model.nf, num_classes=config.model.num_scales, sigma_begin=config.model.sigma_max,
                        sigma_end=config.model.sigma_min, sigma_dist="geometric",
                        normalization=config.model.normalization, nonlinearity=config.model.nonlinearity,
                        spec_norm=False))
    return ns


def _wrap(base, name):
    class _Model(base):
        def __init__(self, config):
            super().__init__(_adapt(config))
            self.centered = config.data.centered
            self.nf = config.model.nf
            self.config = config
            # float32 cast of the float64 geometric schedule, as `torch.tensor(get_sigmas(config))` in the reference
            self.sigmas = torch.tensor(utils.get_sigmas(config)).to(self.sigmas.device)

        def forward(self, x, y):
            out = super().forward(x, y.long())
            return out.to(torch.float64) if self.sigmas.dtype == torch.float64 else out
    _Model.__name__ = _Model.__qualname__ = name
    return _Model


NCSNv2 = utils.register_model(name='ncsnv2_64')(_wrap(_ncsn.NCSNv2, "NCSNv2"))
NCSNv2_128 = utils.register_model(name='ncsnv2_128')(_wrap(_ncsn.NCSNv2Deeper, "NCSNv2_128"))
NCSNv2_256 = utils.register_model(name='ncsnv2_256')(_wrap(_ncsn.NCSNv2Deepest, "NCSNv2_256"))


def get_network(config):
    if config.data.image_size < 96:
        return functools.partial(NCSNv2, config=config)
    if 96 <= config.data.image_size <= 128:
        return functools.partial(NCSNv2_128, config=config)
    if 128 < config.data.image_size <= 256:
        return functools.partial(NCSNv2_256, config=config)
    raise NotImplementedError(f'No network suitable for {config.data.image_size}px implemented yet.')
