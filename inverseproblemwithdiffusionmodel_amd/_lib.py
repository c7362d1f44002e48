"""ctypes binding of libipdm.so (the C ABI declared in include/ipdm.h).

The HIP library is the product: there is NO CPU or eager-PyTorch fallback.  Importing this module
raises if the shared object has not been built (``python -m inverseproblemwithdiffusionmodel_amd.csrc.build``
or ``__graft_entry__.build()``), and every op raises when handed a tensor that does not live on a GPU.
"""
import ctypes
import os

import torch  # noqa: F401  -- MUST precede the dlopen below: libipdm.so then binds to the HIP runtime that
#                              PyTorch-ROCm already loaded (its bundled libamdhip64), so both share one
#                              runtime, device context and stream table.  Loading /opt/rocm's copy first
#                              leaves torch and the kernels on two different runtimes ("no device", error 100).
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p


class ConvExt(ctypes.Structure):
    """ipdm_conv_ext_t (include/ipdm.h): optional extras of the split-operand convolution calls"""
    _fields_ = [("in_amax", c_void_p), ("bias_bstride", c_int), ("out_scale", c_float), ("res_second", c_int),
                ("out_amax", c_void_p), ("act_amax", c_void_p)]


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPDM_LIB") or os.path.join(_HERE, "libipdm.so")     # IPDM_LIB: a diagnostic build (scripts/build_variant.sh)

P = c_void_p   # device pointer

# name -> argtypes (restype is int unless listed in _RESTYPES).  Order matches include/ipdm.h.
SIGNATURES = {
    "ipdm_abi_version": [],
    "ipdm_build_arch": [],
    "ipdm_upfirdn2d_f32": [P, P, P] + [c_int] * 14 + [P],
    "ipdm_fused_bias_act_f32": [P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, P],
    "ipdm_upfirdn2d_f16": [P, P, P] + [c_int] * 14 + [P],
    "ipdm_upfirdn2d_f64": [P, P, P] + [c_int] * 14 + [P],
    "ipdm_fused_bias_act_f16": [P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, P],
    "ipdm_fused_bias_act_f64": [P, P, P, P, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, P],
    "ipdm_fft2c_c64": [P, P, c_int, c_int, c_int, c_int, P, P],
    "ipdm_fft2c_workspace_bytes": [c_int, c_int, c_int],
    "ipdm_sense_forward_c64": [P, P, P, c_int, P, c_int, c_int, c_int, c_int, P],
    "ipdm_sense_workspace_bytes": [c_int, c_int, c_int, c_int],
    "ipdm_sense_adjoint_c64": [P, P, P, c_int, c_int, P, P, c_int, c_int, c_int, c_int, P],
    "ipdm_sense_ssos_c64": [P, P, P, c_int, c_int, c_int, c_int, P],
    "ipdm_sense_l2prox_f32": [P, P, P, P, P, c_int, c_float, P, P, P, c_int, c_int, c_int, c_int, P],
    "ipdm_ald_sense_step_f32": [P, P, P, P, P, P, c_float, c_float, c_uint64, c_int64, c_int64, P,
                                P, P, P, c_int, c_float, P, c_int, c_int, c_int, c_int, P],
    "ipdm_singlecoil_prox_f32": [P, P, P, P, c_int, c_float, c_int, P, P, P, c_int, c_int, c_int, P],
    "ipdm_ald_singlecoil_step_f32": [P, P, P, P, P, P, c_float, c_float, c_uint64, c_int64, c_int64, P,
                                     P, P, c_int, c_float, c_int, P, c_int, c_int, c_int, P],
    "ipdm_langevin_step_f32": [P, P, P, c_float, c_float, c_uint64, c_int64, c_int64, P, c_int64, c_int64, P],
    "ipdm_philox_normal_f32": [P, c_uint64, c_int64, c_int64, c_int, c_int64, c_int64, P],
    "ipdm_philox_block_host": [c_uint64, c_int64, c_int64, c_int, ctypes.c_uint32, P],
    "ipdm_instnorm_plus_coef_f32": [P, P, P, P, P, c_int, c_int, c_int, P, P],
    "ipdm_affine_act_f32": [P, P, P, c_int, c_int, c_int, c_int, P],
    "ipdm_act_f32": [P, P, c_int64, c_int, P],
    "ipdm_scale_shift_f32": [P, P, c_int64, c_float, c_float, P],
    "ipdm_add_f32": [P, P, P, c_int64, P],
    "ipdm_div_sigma_f32": [P, P, P, P, c_int, c_int64, P],
    "ipdm_maxpool5_f32": [P, P, c_int, c_int, c_int, P],
    "ipdm_meanpool2_f32": [P, P, c_int, c_int, c_int, P],
    "ipdm_bilinear_f32": [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    "ipdm_conv3x3_thin_supported": [c_int, c_int, c_int, c_int],
    "ipdm_conv3x3_thin_f32": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "ipdm_trilinear_f32": [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    "ipdm_groupnorm_coef_f32": [P, P, P, P, c_int, c_int, c_int, c_int, c_float, P, P],
    "ipdm_groupnorm_coef_partials_f32": [P, c_int, P, c_int, c_int, P, P, P, c_int, c_int, c_float, P],
    "ipdm_groupnorm_coef_cat_f32": [P, c_int, P, c_int, P, P, P, c_int, c_int, c_int, c_float, P, P],
    "ipdm_affine_act_cat_f32": [P, c_int, P, c_int, P, P, c_int, c_int, c_int, P],
    "ipdm_linear_f32": [P, P, P, P, c_int, c_int, c_int, c_int, P],
    "ipdm_attention_f32": [P, P, P, P, c_int, c_int, c_int, c_float, P],
    "ipdm_axpby_f32": [P, P, P, c_int64, c_float, c_float, P],
    "ipdm_sample_axpy2_f32": [P, P, P, P, P, P, c_int, c_int64, P],
    "ipdm_sample_norm_f32": [P, P, c_int, c_int64, P],
    "ipdm_conv_pack_weight_f32": [P, P, c_int, c_int, c_int, P],
    "ipdm_conv2d_f32": [P, P, P, P, c_int, P, P, P, c_int] + [c_int] * 8 + [P],
    "ipdm_conv_wino_weight_f32": [P, P, c_int, c_int, P],
    "ipdm_conv2d_wino_supported": [c_int, c_int, c_int, c_int, c_int],
    "ipdm_conv2d_wino_f32": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "ipdm_debug_set_stamp_buffer": [P],
    "ipdm_conv3d_f32": [P, P, P, P, c_int, P, P, P, c_int] + [c_int] * 8 + [P],
    "ipdm_maxpool3d5_f32": [P, P, c_int, c_int, c_int, c_int, P],
    "ipdm_temporal_taps_f32": [P, P, c_int, c_int, c_int, c_int, c_int, P],
    "ipdm_conv_bx3_weight_bytes": [c_int, c_int, c_int],
    "ipdm_conv_bx3_pack_weight": [P, P, c_int, c_int, c_int, P],
    "ipdm_conv2d_bx3_f32": [P, P, P, P, c_int, P, P, P, c_int] + [c_int] * 7 + [P, P],
    "ipdm_conv3d_bx3_f32": [P, P, P, P, c_int, P, P, P, c_int] + [c_int] * 8 + [P, P],
    "ipdm_conv_bx3_splitk": [c_int] * 8,
    "ipdm_conv_bx3_splitk_f32": [P, P, P, P, c_int, P, P, P, c_int] + [c_int] * 10 + [P, P, P],
    "ipdm_adam_ascent_f32": [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_int, P],
    "ipdm_conv_wino_bx3_weight_bytes": [c_int, c_int],
    "ipdm_conv_wino_bx3_pack_weight": [P, P, c_int, c_int, P],
    "ipdm_conv2d_wino_bx3_supported": [c_int, c_int, c_int, c_int, c_int],
    "ipdm_conv2d_wino_bx3_f32": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    "ipdm_conv2d_wino_bx3_stats_partials": [c_int, c_int, c_int, c_int, c_int, c_int],
    "ipdm_conv2d_wino_bx3_splitk": [c_int, c_int, c_int, c_int, c_int],
    "ipdm_conv2d_wino_bx3_splitk_f32": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P],
    "ipdm_conv2d_wino_bx3_stats_f32": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P],
    "ipdm_conv_hx2_weight_bytes": [c_int, c_int, c_int],
    "ipdm_conv_hx2_pack_weight": [P, P, c_int, c_int, c_int, P],
    "ipdm_absmax_f32": [P, P, c_int, c_int64, P],
    "ipdm_conv2d_hx2_f32": [P, P, P, P, c_int, P, P, P, c_int] + [c_int] * 7 + [P, P],
    "ipdm_conv3d_hx2_f32": [P, P, P, P, c_int, P, P, P, c_int] + [c_int] * 8 + [P, P],
    "ipdm_conv_hx2_splitk_f32": [P, P, P, P, c_int, P, P, P, c_int] + [c_int] * 10 + [P, P, P],
    "ipdm_conv_wino_hx2_weight_bytes": [c_int, c_int],
    "ipdm_conv_wino1d_weight_bytes": [c_int, c_int],
    "ipdm_conv_wino1d_pack_weight": [P, P, c_int, c_int, P],
    "ipdm_conv2d_wino1d_supported": [c_int, c_int, c_int, c_int],
    "ipdm_conv_wino1d_weight_bytes3d": [c_int, c_int],
    "ipdm_conv_wino1d_pack_weight3d": [P, P, c_int, c_int, P],
    "ipdm_conv3d_wino1d_supported": [c_int, c_int, c_int, c_int, c_int],
    "ipdm_conv3d_wino1d_f32": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    "ipdm_conv2d_wino1d_f32": [P, P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    "ipdm_conv2d_wino1d_stats_partials": [c_int, c_int, c_int, c_int],
    "ipdm_conv2d_wino1d_stats_f32": [P, P, P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P],
    "ipdm_conv_wino_hx2_pack_weight": [P, P, c_int, c_int, P],
    "ipdm_conv2d_wino_hx2_f32": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    "ipdm_conv2d_wino_hx2_splitk_f32": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P],
    "ipdm_conv2d_wino_hx2_stats_f32": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P],
    "ipdm_instnorm_plus_coef_partials_f32": [P, c_int, P, P, P, P, c_int, c_int, c_int, P, P],
    "ipdm_zero_insert2_f32": [P, P, c_int, c_int, c_int, P],
    "ipdm_subsample2_f32": [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "ipdm_in_prelu_fwd_f32": [P, P, P, P, P, c_int, c_int, c_float, P],
    "ipdm_in_prelu_bwd_f32": [P, P, P, P, P, c_int, c_int, P],
    "ipdm_seg_loglh_grad_f32": [P, P, P, c_int, c_int, c_int64, P],
    "ipdm_axpy_sched_f32": [P, P, P, c_int64, P, c_float, c_int64, P],
    "ipdm_magnitude_c64": [P, P, c_int64, P],
    "ipdm_posterior_moments_c64": [P, P, c_int, c_int64, P],
    "ipdm_tv_c64": [P, P, c_int, c_int, c_int, P],
    "ipdm_tv_grad_c64": [P, P, c_int, c_int, c_int, P],
    "ipdm_nrmse_f32": [P, P, P, c_int, c_int64, c_int, P],
    "ipdm_ssim_f32": [P, P, P, c_int, c_int, c_int, c_int, ctypes.c_double, P],
}
_RESTYPES = {"ipdm_build_arch": c_char_p, "ipdm_fft2c_workspace_bytes": c_int64, "ipdm_sense_workspace_bytes": c_int64, "ipdm_conv_bx3_weight_bytes": c_int64,
             "ipdm_conv_wino_bx3_weight_bytes": c_int64, "ipdm_conv_hx2_weight_bytes": c_int64,
             "ipdm_conv_wino_hx2_weight_bytes": c_int64, "ipdm_conv_wino1d_weight_bytes": c_int64, "ipdm_conv_wino1d_weight_bytes3d": c_int64}

IPDM_EINVAL = -1
IPDM_EUNSUPPORTED = -2


class IpdmError(RuntimeError):
    pass


class IpdmUnsupported(IpdmError, NotImplementedError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP kernels are not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or inverseproblemwithdiffusionmodel_amd/csrc/build.py). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the library lacks a declared entry point
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    return lib


lib = _load()


def check(rc, what):
    if rc == 0:
        return
    if rc == IPDM_EUNSUPPORTED:
        raise IpdmUnsupported(f"{what}: no gfx950 kernel for this size/combination (IPDM_EUNSUPPORTED)")
    if rc == IPDM_EINVAL:
        raise IpdmError(f"{what}: invalid argument (IPDM_EINVAL)")
    raise IpdmError(f"{what}: HIP error {rc}")


def call(name, *args):
    check(getattr(lib, name)(*args), name)
