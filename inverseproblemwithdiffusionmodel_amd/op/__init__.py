"""StyleGAN2 resampling / activation operators on the gfx950 kernels (`ipdm_upfirdn2d_f32`, `ipdm_fused_bias_act_f32`).

Public names of the package, as the score_sde models import them: ``upfirdn2d`` (function), ``fused_leaky_relu``
(function) and ``FusedLeakyReLU`` (module)."""
import importlib

_fused = importlib.import_module(__name__ + ".fused_act")
_resample = importlib.import_module(__name__ + ".upfirdn2d")

fused_leaky_relu = _fused.fused_leaky_relu
FusedLeakyReLU = _fused.FusedLeakyReLU
upfirdn2d = _resample.upfirdn2d

__all__ = ["upfirdn2d", "fused_leaky_relu", "FusedLeakyReLU"]
