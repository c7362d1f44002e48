"""``load_config`` (mirror of the reference's ``helpers/load_data.py:301-321``): YAML -> nested Namespace,
``config.device`` injected, ``mode == "complex"`` switches to 2 channels.  Dataset loaders are out of scope
(no data ships with the reference); synthetic inputs come from ``synthetic.py``."""
import os

import torch

from .utils import load_yml_file

_CFG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ncsn", "configs")
REGISTERED_DATA_CONFIG_FILENAME = {
    "MNIST": os.path.join(_CFG_DIR, "mnist.yml"),
    "CINE127": os.path.join(_CFG_DIR, "cine127.yml"),
    "CINE127_1D": os.path.join(_CFG_DIR, "cine127_1d.yml"),
    "ACDC": os.path.join(_CFG_DIR, "acdc.yml"),
}


def load_config(ds_name, mode="real-valued", device=None, **kwargs):
    assert mode in ["real-valued", "mag", "complex", "real-imag", "real-imag-random"]
    assert ds_name in REGISTERED_DATA_CONFIG_FILENAME.keys()
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    cfg = load_yml_file(REGISTERED_DATA_CONFIG_FILENAME[ds_name])
    cfg.device = device
    if mode == "complex":
        cfg.data.channels = 2
    return cfg
