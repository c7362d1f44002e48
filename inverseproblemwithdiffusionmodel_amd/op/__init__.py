from .fused_act import FusedLeakyReLU, fused_leaky_relu  # noqa: F401
from .upfirdn2d import upfirdn2d  # noqa: F401
