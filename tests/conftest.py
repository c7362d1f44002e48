import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]
    return load


def state_dict_from_golden(d, prefix):
    """g07_layers stores 'prefix__a__b' -> tensor; rebuild {'a.b': torch tensor}."""
    import torch
    p = prefix + "__"
    return {k[len(p):].replace("__", "."): torch.from_numpy(d[k]) for k in d.files if k.startswith(p)}
