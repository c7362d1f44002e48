"""3-D clones of the RefineNet blocks for the temporal prior (mirror of the reference's
``ncsn/models/layers3d.py`` = layers.py with Conv3d / MaxPool3d / trilinear).  The blocks in layers.py are
dimension-generic (``ndim``); these are the ndim=3 bindings with the reference's names."""
from functools import partial

from . import layers
from .layers import get_act, get_normalization  # noqa: F401
from .normalization import InstanceNorm2dPlus as InstanceNorm3dPlus  # per-(b,c) statistics over D*H*W  # noqa: F401

conv1x1 = partial(layers.conv1x1, ndim=3)
conv3x3 = partial(layers.conv3x3, ndim=3)
dilated_conv3x3 = partial(layers.dilated_conv3x3, ndim=3)
CRPBlock = partial(layers.CRPBlock, ndim=3)
RCUBlock = partial(layers.RCUBlock, ndim=3)
MSFBlock = partial(layers.MSFBlock, ndim=3)
RefineBlock = partial(layers.RefineBlock, ndim=3)
ResidualBlock = partial(layers.ResidualBlock, ndim=3)
Conv3d = partial(layers.Conv2d, ndim=3)
