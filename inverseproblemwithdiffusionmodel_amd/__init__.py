"""MI355X-native Annealed Langevin Dynamics reconstruction engine.

Host-side mirror of the reference's hot-path API (``ncsn.linear_transforms``,
``ncsn.models.{ALD_optimizers,proximal_op,ncsnv2}``, ``op.{upfirdn2d,fused_act}``) above a
C-ABI shared library of hand-written gfx950 HIP kernels (``csrc/``, declared in ``include/ipdm.h``).
"""
__version__ = "0.1.0"


def install_reference_alias():
    """Make ``import InverseProblemWithDiffusionModel.<pkg>`` resolve to this package, so scripts
    written against the reference's absolute imports (e.g. ALD_optimizers.py:6,10,18) run unchanged."""
    import sys
    sys.modules.setdefault("InverseProblemWithDiffusionModel", sys.modules[__name__])
