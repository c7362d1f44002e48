#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the Python reference.

Runs ONLY in the build container (needs /root/reference, CPU only).  The reference is
imported from a symlink under /tmp with the absent third-party modules stubbed
(SURVEY.md section 8c); nothing of it is copied here -- the outputs are DATA (inputs +
expected outputs of the reference's own functions), a few hundred KB in total.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixture groups (names follow SURVEY.md 8c):
  g01_masks      generate_mask bool arrays                       (int, bit exact)
  g02_sens       SENSE coil maps f64 + anchors
  g03_fft        i2k_complex / k2i_complex
  g04_sense      SENSE forward / adjoint / SSOS (T=1 mask variant and live T=24 variant)
  g05_prox       L2Penalty / SingleCoil outputs
  g06_sigmas     get_sigmas for the shipped schedules
  g07_layers     InstanceNorm2dPlus / ConvMeanPool / ResidualBlock / RefineBlock / tiny
                 NCSNv2Deepest known-answer tests with the weights stored alongside
  g08_ald        ALDInvSegProximalRealImag + ALDUnconditionalSampler trajectories with
                 the injected noise recorded
  g09_upfirdn    upfirdn2d_native / upsample_2d / downsample_2d
  g10_biasact    CPU fused_leaky_relu
  g11_temporal   reshape_temporal_dim, FiniteDiff
  g12_sigmas_T   ALD2DTime's temporal schedule aligned onto the spatial one (nearest interpolation, -1 head)
  g16_ncsn3d     tiny NCSN3DShallow forward with its weights + a MaxPool3d(5,1,2) case
  g17_ald2dtime  ALD2DTime trajectories (diffusion1d / tv / none) with the injected noise recorded
  g13_pc         one 20-step VE predictor-corrector run on the tiny NCSN++ + every predictor/corrector update
                 function for VE / VP / subVP SDEs with an analytic score, noise recorded
  g14_ncsnpp     tiny NCSN++ (BigGAN blocks, FIR resampling, attention, progressive I/O) forward with its weights
  g15_fullnet    full-size NCSNv2Deepest (ngf=128, 128x128) forward on synthetic weights
                 produced by inverseproblemwithdiffusionmodel_amd.synthetic.synth_state_dict (weights NOT stored)
"""
import hashlib
import io
import os
import sys
import contextlib
from argparse import Namespace
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
ALIAS_ROOT = "/tmp/ipdm_oracle"


def _import_reference():
    os.makedirs(ALIAS_ROOT, exist_ok=True)
    link = os.path.join(ALIAS_ROOT, "InverseProblemWithDiffusionModel")
    if not os.path.islink(link):
        os.symlink(REF, link)
    sys.path.insert(0, ALIAS_ROOT)
    sys.path.insert(0, REPO)
    absent = ["monai", "monai.networks", "monai.networks.nets", "monai.transforms", "monai.data",
              "monai.losses", "monai.utils", "monai.metrics", "monai.inferers", "pytorch_lightning",
              "pytorch_lightning.callbacks", "pytorch_lightning.loggers", "skimage", "skimage.metrics",
              "torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.utils",
              "SimpleITK", "torchmetrics", "tensorboard", "torch.utils.tensorboard", "kornia", "cv2",
              "nibabel", "sigpy", "ml_collections"]
    for m in absent:
        if m not in sys.modules:
            sys.modules[m] = MagicMock()
    import torch.utils.cpp_extension as cpp_ext
    cpp_ext.load = lambda *a, **k: MagicMock()   # never JIT the reference's CUDA sources


_import_reference()
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from InverseProblemWithDiffusionModel.ncsn import linear_transforms as ref_lt  # noqa: E402
from InverseProblemWithDiffusionModel.ncsn.linear_transforms import undersampling_fourier as ref_uf  # noqa: E402
from InverseProblemWithDiffusionModel.ncsn.linear_transforms.finite_diff import FiniteDiff as RefFiniteDiff  # noqa: E402
from InverseProblemWithDiffusionModel.ncsn.models import proximal_op as ref_prox  # noqa: E402
from InverseProblemWithDiffusionModel.ncsn.models import ALD_optimizers as ref_ald  # noqa: E402
from InverseProblemWithDiffusionModel.ncsn.models import get_sigmas as ref_get_sigmas  # noqa: E402
from InverseProblemWithDiffusionModel.ncsn.models import ncsnv2 as ref_ncsnv2  # noqa: E402
from InverseProblemWithDiffusionModel.ncsn.models import layers as ref_layers  # noqa: E402
from InverseProblemWithDiffusionModel.ncsn.models import normalization as ref_norm  # noqa: E402
from InverseProblemWithDiffusionModel.helpers import utils as ref_utils  # noqa: E402
import importlib  # noqa: E402
ref_upfirdn = importlib.import_module("InverseProblemWithDiffusionModel.op.upfirdn2d")   # op/__init__ shadows the names
ref_fused = importlib.import_module("InverseProblemWithDiffusionModel.op.fused_act")
from InverseProblemWithDiffusionModel.models import up_or_down_sampling as ref_updown  # noqa: E402

quiet = contextlib.redirect_stdout(io.StringIO())

# mask parameter sets; the first three are the reference's (undersampling_fourier.py:68-73),
# R40 is the build's choice (no R=40 set exists in the reference)
MASK_PARAMS = {
    "R20": dict(sw=0.07, sm=0.3, sa=0.01782),
    "R16": dict(sw=0.07926, sm=0.42, sa=0.02),
    "R8": dict(sw=0.196, sm=0.5, sa=0.02),
    "R40": dict(sw=0.07, sm=0.11, sa=0.0065),
}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(arrays)} arrays)")


def npy(t):
    return t.detach().cpu().numpy()


def tiny_config(ngf=4, num_classes=10, sigma_begin=1.0, sigma_end=0.01, channels=1, image_size=32):
    return Namespace(
        device=torch.device("cpu"),
        data=Namespace(channels=channels, image_size=image_size, logit_transform=False, rescaled=False,
                       uniform_dequantization=False, gaussian_dequantization=False),
        model=Namespace(ngf=ngf, num_classes=num_classes, sigma_begin=sigma_begin, sigma_end=sigma_end,
                        sigma_dist="geometric", normalization="InstanceNorm++", nonlinearity="elu",
                        spec_norm=False),
        recons=Namespace(sigma_dist="geometric", sigma_begin=sigma_begin, sigma_end=sigma_end,
                         num_classes=num_classes),
        sampling=Namespace(n_steps_each=3, step_lr=9e-7, final_only=True, denoise=True),
    )


def t1_mask_patch(params):
    """RandomUndersamplingFourier._generate_mask restricted to the T=1 variant the reference keeps
    commented out (undersampling_fourier.py:72-73): generate_mask(1, W, ...).unsqueeze(1)."""
    def _gen(self):
        torch.random.manual_seed(self.seed)
        C, H, W = self.in_shape
        return ref_lt.generate_mask(1, W, seed=self.seed, **params).unsqueeze(1)
    return _gen


# ---------------------------------------------------------------------------------------------
def g01_masks():
    out = {}
    for seed in range(4):
        out[f"R20_T1_N128_seed{seed}"] = npy(ref_lt.generate_mask(1, 128, seed=seed, **MASK_PARAMS["R20"]))
        out[f"R40_T1_N128_seed{seed}"] = npy(ref_lt.generate_mask(1, 128, seed=seed, **MASK_PARAMS["R40"]))
    out["R16_T24_N128_seed0"] = npy(ref_lt.generate_mask(24, 128, seed=0, **MASK_PARAMS["R16"]))
    out["R8_T24_N128_seed0"] = npy(ref_lt.generate_mask(24, 128, seed=0, **MASK_PARAMS["R8"]))
    out["R8_T1_N64_seed5"] = npy(ref_lt.generate_mask(1, 64, seed=5, **MASK_PARAMS["R8"]))
    out["default_T3_N32_seed1"] = npy(ref_lt.generate_mask(3, 32, seed=1))
    out["R20_T1_N256_seed0"] = npy(ref_lt.generate_mask(1, 256, seed=0, **MASK_PARAMS["R20"]))
    for k, v in out.items():
        assert v.dtype == np.bool_, (k, v.dtype)
    print("  R40 seed0 lines:", int(out["R40_T1_N128_seed0"].sum()),
          " R20 seed0 lines:", int(out["R20_T1_N128_seed0"].sum()))
    save("g01_masks", **out)


def g02_sens():
    out = {}
    with quiet:
        for (H, W) in [(32, 32), (128, 128), (24, 64)]:
            op = ref_uf.SENSE("exp", 4, 20, 0.04, (1, H, W), seed=0)
            maps = npy(op.sens_maps)
            assert maps.dtype == np.float64
            anchors = []
            for i in range(4):
                np.random.seed(0 + i)
                anchors.append((np.random.choice(H), np.random.choice(W)))
            key = f"{H}x{W}"
            out[f"anchors_{key}"] = np.array(anchors, dtype=np.int64)
            if H * W <= 2048:
                out[f"maps_{key}"] = maps
            else:
                out[f"maps_{key}_rows8"] = maps[:, ::8, :]
                out[f"maps_{key}_sha256"] = np.frombuffer(
                    hashlib.sha256(np.ascontiguousarray(maps).tobytes()).digest(), dtype=np.uint8)
        op = ref_uf.SENSE("exp", 3, 20, 0.04, (1, 32, 32), seed=7)
        out["maps_32x32_seed7_n3"] = npy(op.sens_maps)
    save("g02_sens", **out)


def g03_fft():
    out = {}
    g = torch.Generator().manual_seed(3)
    for shape in [(2, 1, 8, 8), (1, 2, 7, 9), (1, 1, 32, 32), (1, 1, 6, 5)]:
        x = torch.complex(torch.randn(shape, generator=g), torch.randn(shape, generator=g))
        key = "x".join(map(str, shape))
        out[f"x_{key}"] = npy(x)
        out[f"i2k_{key}"] = npy(ref_lt.i2k_complex(x))
        out[f"k2i_{key}"] = npy(ref_lt.k2i_complex(x))
    save("g03_fft", **out)


def g04_g05_sense_prox():
    out4, out5 = {}, {}
    g = torch.Generator().manual_seed(4)
    H = W = 32
    orig = ref_uf.RandomUndersamplingFourier._generate_mask
    try:
        ref_uf.RandomUndersamplingFourier._generate_mask = t1_mask_patch(MASK_PARAMS["R8"])
        with quiet:
            op = ref_uf.SENSE("exp", 4, 8, 0.04, (1, H, W), seed=0)
        x = torch.complex(torch.randn(2, 1, H, W, generator=g), torch.randn(2, 1, H, W, generator=g))
        y = op(x)
        s = torch.complex(torch.randn(4, 2, 1, H, W, generator=g), torch.randn(4, 2, 1, H, W, generator=g))
        out4["mask_T1"] = npy(op.random_under_fourier.mask)
        out4["x"] = npy(x)
        out4["Ax"] = npy(y)
        out4["s"] = npy(s)
        out4["AHs"] = npy(op.conj_op(s))
        out4["ssos_s"] = npy(op.SSOS(s))
        out4["loglh_grad"] = npy(op.log_lh_grad(x, s, 0.7))
        # L2Penalty on the SENSE operator
        z = torch.complex(torch.randn(2, 1, H, W, generator=g), torch.randn(2, 1, H, W, generator=g))
        out5["z"] = npy(z)
        out5["y"] = npy(y)
        prox = ref_prox.L2Penalty(op)
        for i, (alpha, lamda) in enumerate([(0.9, 1.0), (3.0, 0.5), (9e-7, 1.0)]):
            with quiet:
                xs = prox(z, y, alpha, lamda)
            torch.set_grad_enabled(True)
            out5[f"l2_sense_{i}_alpha_lamda"] = np.array([alpha, lamda], dtype=np.float64)
            out5[f"l2_sense_{i}_x"] = npy(xs)
        # single-coil operator: L2Penalty (K = B) and SingleCoil closed form
        with quiet:
            sc = ref_uf.RandomUndersamplingFourier(8, 0.04, (1, H, W), seed=2)
        ysc = sc(x)
        out5["sc_mask"] = npy(sc.mask)
        out5["sc_y"] = npy(ysc)
        out5["sc_Ax_adj"] = npy(sc.conj_op(ysc))
        with quiet:
            out5["l2_sc_x"] = npy(ref_prox.L2Penalty(sc)(z, ysc, 0.9, 1.0))
            torch.set_grad_enabled(True)
            p = ref_prox.SingleCoil(sc)
            xs = p(z, ysc, 0.8, 2.0)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out5["singlecoil_check"] = np.array(float(p.check_solution(xs, z, ysc, 0.8, 2.0)))
        out5["singlecoil_x"] = npy(xs)
        out5["singlecoil_alpha_lamda"] = np.array([0.8, 2.0])
    finally:
        ref_uf.RandomUndersamplingFourier._generate_mask = orig
    # live (hard-wired T=24, "R=16") variant: a (24,1,H,W) stack broadcasts against (24,1,1,W)
    with quiet:
        op24 = ref_uf.SENSE("exp", 4, 16, 0.04, (1, H, W), seed=0)
    x24 = torch.complex(torch.randn(24, 1, H, W, generator=g), torch.randn(24, 1, H, W, generator=g))
    out4["mask_T24"] = npy(op24.random_under_fourier.mask)
    out4["x24"] = npy(x24)
    out4["Ax24"] = npy(op24(x24))
    save("g04_sense", **out4)
    save("g05_prox", **out5)


def g06_sigmas():
    out = {}
    for name, (s0, s1, L) in {"acdc": (348, 0.01, 2311), "cine127": (60, 0.01, 1000),
                              "cine127_1d": (40, 0.01, 400), "mnist": (50, 0.01, 232)}.items():
        cfg = tiny_config(num_classes=L, sigma_begin=s0, sigma_end=s1)
        sig = npy(ref_get_sigmas(cfg, "recons"))
        assert sig.dtype == np.float32
        out[name] = sig
    cfg = tiny_config(num_classes=17, sigma_begin=3.0, sigma_end=0.5)
    cfg.model.sigma_dist = "uniform"
    out["uniform_17"] = npy(ref_get_sigmas(cfg, "unconditioned"))
    lw = ref_ald.get_lh_weights(torch.tensor(out["mnist"]), 0.25, "linear")
    out["lh_weights_mnist_0.25"] = npy(lw)
    save("g06_sigmas", **out)


def _sd(module, prefix):
    return {prefix + "__" + k.replace(".", "__"): npy(v) for k, v in module.state_dict().items()}


def g07_layers(write=True):
    out = {}
    torch.manual_seed(7)
    act = nn.ELU()
    x = torch.randn(2, 6, 12, 10)
    # InstanceNorm2dPlus with perturbed beta so the bias path is visible
    n = ref_norm.InstanceNorm2dPlus(6)
    n.beta.data.normal_(0, 0.1)
    out["in_x"] = npy(x)
    out.update(_sd(n, "in"))
    out["in_y"] = npy(n(x))
    # ConvMeanPool 3x3 and 1x1
    cmp3 = ref_layers.ConvMeanPool(6, 5, 3)
    out.update(_sd(cmp3, "cmp3"))
    out["cmp3_y"] = npy(cmp3(x))
    # ResidualBlock: plain / plain with channel change / pooled / dilated-down / dilated-same
    norm = ref_norm.InstanceNorm2dPlus
    variants = {
        "rb_plain": dict(input_dim=6, output_dim=6, resample=None),
        "rb_widen": dict(input_dim=6, output_dim=8, resample=None),
        "rb_pool": dict(input_dim=6, output_dim=8, resample="down"),
        "rb_dil_down": dict(input_dim=6, output_dim=8, resample="down", dilation=2),
        "rb_dil_same": dict(input_dim=6, output_dim=6, resample=None, dilation=4),
    }
    for name, kw in variants.items():
        rb = ref_layers.ResidualBlock(act=act, normalization=norm, **kw)
        out.update(_sd(rb, name))
        out[name + "_y"] = npy(rb(x))
    # RefineBlock: start / two-input (second input at half resolution) / end
    xa = torch.randn(2, 6, 12, 10)
    xb = torch.randn(2, 4, 6, 5)
    out["rf_xa"], out["rf_xb"] = npy(xa), npy(xb)
    rf = ref_layers.RefineBlock([6], 6, act=act, start=True)
    out.update(_sd(rf, "rf_start"))
    out["rf_start_y"] = npy(rf([xa], xa.shape[2:]))
    rf = ref_layers.RefineBlock([6, 4], 5, act=act)
    out.update(_sd(rf, "rf_two"))
    out["rf_two_y"] = npy(rf([xa, xb], xa.shape[2:]))
    rf = ref_layers.RefineBlock([6, 4], 6, act=act, end=True)
    out.update(_sd(rf, "rf_end"))
    out["rf_end_y"] = npy(rf([xa, xb], xa.shape[2:]))
    # tiny NCSNv2Deepest, 32x32, 10 levels
    cfg = tiny_config()
    torch.manual_seed(70)
    with quiet:
        net = ref_ncsnv2.NCSNv2Deepest(cfg).eval()
    for p in net.parameters():          # make biases / betas non-trivial
        if p.ndim == 1:
            p.data.add_(0.05 * torch.randn_like(p))
    xin = torch.rand(3, 1, 32, 32)
    labels = torch.tensor([0, 4, 9])
    out.update(_sd(net, "net"))
    out["net_x"] = npy(xin)
    out["net_labels"] = npy(labels)
    with torch.no_grad():
        out["net_y"] = npy(net(xin, labels))
    if write:
        save("g07_layers", **out)
    return net, cfg


class _StandInSeg(nn.Module):
    """finite-gradient stand-in for the MONAI UNet (weight is multiplied by lh_weight = 0)."""
    def __init__(self):
        super().__init__()
        self.c = nn.Conv2d(1, 2, 1)

    def forward(self, x):
        return self.c(x)


class _NoiseTape:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.tape = []

    def __call__(self, like):
        n = torch.randn(like.shape, generator=self.g, dtype=like.dtype)
        self.tape.append(npy(n))
        return n


def g08_ald(net, cfg):
    out = {}
    H = W = 32
    ref_ald.vis_images = lambda *a, **k: None
    ref_ald.vis_multi_channel_signal = lambda *a, **k: None
    orig = ref_uf.RandomUndersamplingFourier._generate_mask
    try:
        ref_uf.RandomUndersamplingFourier._generate_mask = t1_mask_patch(MASK_PARAMS["R8"])
        with quiet:
            op = ref_uf.SENSE("exp", 4, 8, 0.04, (1, H, W), seed=0)
    finally:
        ref_uf.RandomUndersamplingFourier._generate_mask = orig
    g = torch.Generator().manual_seed(8)
    img = torch.complex(torch.rand(1, 1, H, W, generator=g), 0.3 * torch.randn(1, 1, H, W, generator=g))
    B = 2
    meas = op(img).repeat(1, B, 1, 1, 1)
    sigmas = ref_get_sigmas(cfg, "recons")
    params = dict(n_steps_each=3, step_lr=9e-7, denoise=True, final_only=True)
    out["img"] = npy(img)
    out["measurement"] = npy(meas)
    out["sigmas"] = npy(sigmas)
    for tag, lr_scaled in [("dc_visible", 2.0e6), ("script_default", 1.0)]:
        tape = _NoiseTape(80)
        real_randn_like = torch.randn_like
        torch.randn_like = tape
        try:
            sampler = ref_ald.ALDInvSegProximalRealImag(
                ref_prox.get_proximal("L2Penalty")(op), 1.0, "linear",
                (B, 1, H, W), net, sigmas, params, cfg, meas, op, seg=_StandInSeg(), device=torch.device("cpu"))
            with quiet:
                res = sampler(label=torch.zeros(B, 1, H, W, dtype=torch.long), lamda=0.1, save_dir="/tmp/ipdm_oracle/out",
                              lr_scaled=lr_scaled, seg_mode="full")[0]
        finally:
            torch.randn_like = real_randn_like
            torch.set_grad_enabled(True)
        out[f"{tag}_lr_scaled"] = np.array(lr_scaled)
        out[f"{tag}_x"] = npy(res)
        if tag == "dc_visible":
            out["noise"] = np.stack(tape.tape)      # (60, B, 1, H, W): real, imag alternating per step
    # unconditional sampler (config 1 shape family), init + noise recorded
    tape = _NoiseTape(81)
    real_randn_like, real_rand = torch.randn_like, torch.rand
    gi = torch.Generator().manual_seed(82)
    x0 = real_rand(2, 1, H, W, generator=gi)
    torch.randn_like = tape
    torch.rand = lambda *shape, **k: x0.clone()
    try:
        sampler = ref_ald.ALDUnconditionalSampler((2, 1, H, W), net, sigmas, dict(params, step_lr=2e-5), cfg,
                                                  device=torch.device("cpu"))
        with quiet:
            res = sampler()[0]
    finally:
        torch.randn_like, torch.rand = real_randn_like, real_rand
        torch.set_grad_enabled(True)
    out["uncond_x0"] = npy(x0)
    out["uncond_noise"] = np.stack(tape.tape)
    out["uncond_step_lr"] = np.array(2e-5)
    out["uncond_x"] = npy(res)
    save("g08_ald", **out)


def g18_map(net, cfg):
    """MAP baseline: the reference's SENSEMAP (= MAPOptimizer, MAP_optimizers.py:55-116) on the
    tiny score net: 50 Adam(0.5, 0.5) iterations on x from the zero-filled reconstruction"""
    from InverseProblemWithDiffusionModel.ncsn.models import MAP_optimizers as ref_map
    H = W = 32
    orig = ref_uf.RandomUndersamplingFourier._generate_mask
    try:
        ref_uf.RandomUndersamplingFourier._generate_mask = t1_mask_patch(MASK_PARAMS["R8"])
        with quiet:
            op = ref_uf.SENSE("exp", 4, 8, 0.04, (1, H, W), seed=0)
    finally:
        ref_uf.RandomUndersamplingFourier._generate_mask = orig
    g = torch.Generator().manual_seed(18)
    img = torch.complex(torch.rand(1, 1, H, W, generator=g), 0.3 * torch.randn(1, 1, H, W, generator=g))
    meas = op(img)
    out = {"img": npy(img), "measurement": npy(meas)}
    for tag, lamda, lr in [("a", 1e-2, 1e-3), ("b", 0.5, 5e-3)]:
        cfg_map = Namespace(**vars(cfg))
        cfg_map.MAP = Namespace(n_iters=50, lr=lr, complex_inner_n_steps=20)
        x_init = op.conj_op(meas).clone()
        out[f"{tag}_x_init"] = npy(x_init).copy()
        opt = ref_map.SENSEMAP(x_init, meas, net, op, lamda, cfg_map, logger=MagicMock(), device=torch.device("cpu"))
        with quiet, contextlib.redirect_stderr(io.StringIO()):
            x = opt()
        torch.set_grad_enabled(True)
        out[f"{tag}_lamda"], out[f"{tag}_lr"] = np.array(lamda), np.array(lr)
        out[f"{tag}_x"] = npy(x)
    save("g18_map", **out)


def g09_upfirdn():
    out = {}
    g = torch.Generator().manual_seed(9)
    k1 = torch.tensor([1., 3., 3., 1.])
    k = torch.outer(k1, k1)
    k = k / k.sum()
    cases = {
        "down2": dict(shape=(2, 3, 8, 8), kernel=k, up=1, down=2, pad=(1, 1)),
        "up2": dict(shape=(2, 3, 8, 8), kernel=k * 4, up=2, down=1, pad=(2, 1)),
        "down2_nonsq": dict(shape=(1, 2, 10, 6), kernel=k, up=1, down=2, pad=(1, 1)),
        "up2_nonsq": dict(shape=(1, 2, 5, 9), kernel=k * 4, up=2, down=1, pad=(2, 1)),
        "up3_down2_k5": dict(shape=(1, 2, 7, 6), kernel=torch.randn(5, 5, generator=g), up=3, down=2, pad=(3, 2)),
        "negpad_k3": dict(shape=(1, 2, 9, 9), kernel=torch.randn(3, 3, generator=g), up=1, down=1, pad=(-1, -2)),
        "up1_down3_k2x4": dict(shape=(2, 1, 11, 12), kernel=torch.randn(2, 4, generator=g), up=1, down=3, pad=(0, 3)),
        "up2_down1_k6": dict(shape=(1, 1, 6, 7), kernel=torch.randn(6, 6, generator=g), up=2, down=1, pad=(4, 3)),
    }
    for name, c in cases.items():
        x = torch.randn(c["shape"], generator=g)
        out[f"{name}_x"] = npy(x)
        out[f"{name}_k"] = npy(c["kernel"])
        out[f"{name}_udp"] = np.array([c["up"], c["down"], c["pad"][0], c["pad"][1]], dtype=np.int64)
        out[f"{name}_y"] = npy(ref_upfirdn.upfirdn2d(x, c["kernel"], c["up"], c["down"], c["pad"]))
    x = torch.randn(2, 3, 8, 12, generator=g)
    out["wrap_x"] = npy(x)
    out["wrap_up"] = npy(ref_updown.upsample_2d(x, (1, 3, 3, 1), factor=2))
    out["wrap_down"] = npy(ref_updown.downsample_2d(x, (1, 3, 3, 1), factor=2))
    save("g09_upfirdn", **out)


def g10_biasact():
    out = {}
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 5, 4, 6, generator=g)
    b = torch.randn(5, generator=g)
    out["x"], out["b"] = npy(x), npy(b)
    out["y_default"] = npy(ref_fused.fused_leaky_relu(x, b))
    out["y_scale1.5"] = npy(ref_fused.fused_leaky_relu(x, b, 0.2, 1.5))
    x2 = torch.randn(3, 7, generator=g)
    b2 = torch.randn(7, generator=g)
    out["x2"], out["b2"] = npy(x2), npy(b2)
    out["y2"] = npy(ref_fused.fused_leaky_relu(x2, b2))
    save("g10_biasact", **out)


def g11_temporal():
    out = {}
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 5, 16, 24, generator=g)
    f = ref_utils.reshape_temporal_dim(x, 8, 8, "forward")
    out["x"] = npy(x)
    out["fwd"] = npy(f)
    out["bwd"] = npy(ref_utils.reshape_temporal_dim(f, 8, 8, "backward", img_size=(16, 24)))
    v = torch.randn(1, 1, 6, 4, 4, generator=g)
    fd = RefFiniteDiff(2)
    out["fd_x"] = npy(v)
    out["fd_fwd"] = npy(fd(v))
    out["fd_adj"] = npy(fd.conj_op(v))
    out["fd_tvgrad"] = npy(fd.log_lh_grad(v, lamda=0.3))
    save("g11_temporal", **out)


class _CfgDict(dict):
    """stand-in for ml_collections.ConfigDict (absent in this image): attribute access on a dict"""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def tiny_ncsnpp_config():
    c = _CfgDict()
    c.training = _CfgDict(continuous=True, sde="vesde")
    c.sampling = _CfgDict(n_steps_each=1, noise_removal=True, probability_flow=False, snr=0.16, method="pc",
                          predictor="reverse_diffusion", corrector="langevin")
    c.data = _CfgDict(image_size=32, centered=False, num_channels=3)
    c.model = _CfgDict(name="ncsnpp", sigma_max=50.0, sigma_min=0.01, num_scales=20, beta_min=0.1, beta_max=20.,
                       dropout=0., embedding_type="fourier", scale_by_sigma=True, ema_rate=0.999,
                       normalization="GroupNorm", nonlinearity="swish", nf=8, ch_mult=(1, 2, 2), num_res_blocks=1,
                       attn_resolutions=(8,), resamp_with_conv=True, conditional=True, fir=True,
                       fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type="biggan", progressive="output_skip",
                       progressive_input="input_skip", progressive_combine="sum", attention_type="ddpm",
                       init_scale=0., fourier_scale=16, conv_size=3)
    c.device = torch.device("cpu")
    return c


def g13_g14_score_sde():
    """g14: tiny NCSN++ forward.  g13: a 20-step VE predictor-corrector run on it with the noise recorded."""
    ref_ncsnpp = importlib.import_module("InverseProblemWithDiffusionModel.models.ncsnpp")
    ref_sde_lib = importlib.import_module("InverseProblemWithDiffusionModel.sde.sde_lib")
    ref_sampling = importlib.import_module("InverseProblemWithDiffusionModel.sde.sampling")
    cfg = tiny_ncsnpp_config()
    torch.manual_seed(14)
    net = ref_ncsnpp.NCSNpp(cfg).eval()
    for p_ in net.parameters():             # init_scale = 0 leaves whole branches at 1e-10: make every path count
        if p_.requires_grad:
            p_.data = 0.15 * torch.randn_like(p_)
    out = {}
    out.update(_sd(net, "pp"))
    g = torch.Generator().manual_seed(140)
    x = torch.rand(2, 3, 32, 32, generator=g)
    sig = torch.tensor([0.7, 12.0])
    with torch.no_grad():
        out["pp_x"], out["pp_sigma"], out["pp_y"] = npy(x), npy(sig), npy(net(x, sig))
    out["pp_n_modules"] = np.array(len(net.all_modules))
    save("g14_ncsnpp", **out)

    out = {}
    sde = ref_sde_lib.VESDE(sigma_min=0.01, sigma_max=50.0, N=20)
    tape = _NoiseTape(130)
    x0 = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(131)) * 50.0
    real_randn_like = torch.randn_like
    torch.randn_like = tape
    orig_prior = sde.prior_sampling
    sde.prior_sampling = lambda shape: x0.clone()
    try:
        fn = ref_sampling.get_sampling_fn(cfg, sde, (2, 3, 32, 32), lambda v: v, 1e-5)
        with torch.no_grad():
            samples, nfe = fn(net)
    finally:
        torch.randn_like = real_randn_like
        sde.prior_sampling = orig_prior
    out["x0"], out["noise"], out["samples"], out["nfe"] = npy(x0), np.stack(tape.tape), npy(samples), np.array(nfe)
    # single update functions with an analytic score, several SDEs
    score = lambda x_, t_: -x_ / (1.0 + t_[:, None, None, None])
    t = torch.tensor([0.9, 0.3])
    xs = torch.randn(2, 1, 8, 8, generator=torch.Generator().manual_seed(132))
    out["upd_x"], out["upd_t"] = npy(xs), npy(t)
    for name, s_ in [("ve", ref_sde_lib.VESDE(0.01, 50.0, 100)), ("vp", ref_sde_lib.VPSDE(0.1, 20, 100)),
                     ("subvp", ref_sde_lib.subVPSDE(0.1, 20, 100))]:
        tape = _NoiseTape(133)
        torch.randn_like = tape
        try:
            for pname in ["euler_maruyama", "reverse_diffusion"] + (["ancestral_sampling"] if name != "subvp" else []):
                a, b = ref_sampling.get_predictor(pname)(s_, score, False).update_fn(xs, t)
                out[f"{name}_{pname}_x"], out[f"{name}_{pname}_mean"] = npy(a), npy(b)
            # (the reference's correctors raise for subVPSDE: it has no `.alphas`, sde/sampling.py:274)
            for cname in (["langevin", "ald"] if name != "subvp" else []):
                a, b = ref_sampling.get_corrector(cname)(s_, score, 0.16, 2).update_fn(xs, t)
                out[f"{name}_{cname}_x"], out[f"{name}_{cname}_mean"] = npy(a), npy(b)
        finally:
            torch.randn_like = real_randn_like
        out[f"{name}_noise"] = np.stack(tape.tape)
        mean, std = s_.marginal_prob(xs, t)
        out[f"{name}_marginal_mean"], out[f"{name}_marginal_std"] = npy(mean), npy(std)
        out[f"{name}_prior_logp"] = npy(s_.prior_logp(xs))
    save("g13_pc", **out)


def g12_g16_g17_2dtime(net2d, cfg2d):
    """temporal prior + ALD2DTime: aligned sigma_T schedule (g12), tiny NCSN3DShallow forward (g16), 2D+time
    sampler trajectories in modes diffusion1d / tv / none with injected noise (g17)."""
    ref_ncsn3d = importlib.import_module("InverseProblemWithDiffusionModel.ncsn.models.ncsn3d")
    T, H, W = 8, 32, 32
    cfgT = tiny_config(ngf=4, num_classes=6, sigma_begin=0.5, sigma_end=0.01, channels=64, image_size=T)
    cfgT.data.channels_3d = 1
    torch.manual_seed(16)
    with quiet:
        netT = ref_ncsn3d.NCSN3DShallow(cfgT).eval()
    for p_ in netT.parameters():
        if p_.ndim == 1:
            p_.data.add_(0.05 * torch.randn_like(p_))
    out = {}
    out.update(_sd(netT, "net3d"))
    g = torch.Generator().manual_seed(160)
    x = torch.rand(3, 64, T, generator=g)
    labels = torch.tensor([0, 3, 5])
    with torch.no_grad():
        out["x"], out["labels"], out["y"] = npy(x), npy(labels), npy(netT(x, labels))
    v = torch.randn(2, 3, 6, 5, 7, generator=g)
    out["mp_x"], out["mp_y"] = npy(v), npy(torch.nn.functional.max_pool3d(v, 5, 1, 2))
    save("g16_ncsn3d", **out)

    # g12: the reference aligns sigma_T onto the spatial schedule in ALD2DTime.__init__
    class _Dummy:
        pass
    out = {}
    for name, (sp, tp) in {"cine": ((60, 0.01, 1000), (40, 0.01, 400)), "tiny": ((1.0, 0.01, 10), (0.5, 0.01, 6))}.items():
        sig = ref_get_sigmas(tiny_config(num_classes=sp[2], sigma_begin=sp[0], sigma_end=sp[1]), "recons")
        sigT = ref_get_sigmas(tiny_config(num_classes=tp[2], sigma_begin=tp[0], sigma_end=tp[1]), "recons")
        d = _Dummy()
        d.config = Namespace(data=Namespace(channels=64))
        smp = ref_ald.ALD2DTime(None, d, sigT, (1, T, 1, H, W), None, sig, {}, None)
        out[f"{name}_sigmas_T"] = npy(smp.sigmas_T)
    save("g12_sigmas_T", **out)

    # g17: trajectories (tiny spatial net from g07, 10 levels; tiny temporal net, 6 levels)
    ref_ald.vis_images = lambda *a, **k: None
    ref_ald.vis_multi_channel_signal = lambda *a, **k: None
    orig = ref_uf.RandomUndersamplingFourier._generate_mask
    try:
        ref_uf.RandomUndersamplingFourier._generate_mask = t1_mask_patch(MASK_PARAMS["R8"])
        with quiet:
            op = ref_uf.SENSE("exp", 4, 8, 0.04, (1, H, W), seed=0)
    finally:
        ref_uf.RandomUndersamplingFourier._generate_mask = orig
    g = torch.Generator().manual_seed(17)
    B = 1
    img = torch.complex(torch.rand(B * T, 1, H, W, generator=g), 0.3 * torch.randn(B * T, 1, H, W, generator=g))
    meas = op(img).reshape(4, B, T, 1, H, W)
    sigmas = ref_get_sigmas(cfg2d, "recons")
    sigmas_T = ref_get_sigmas(cfgT, "recons")
    out = {"measurement": npy(meas), "sigmas": npy(sigmas), "sigmas_T": npy(sigmas_T)}
    params = dict(n_steps_each=2, step_lr=2e-5, denoise=False, final_only=True)
    for mode, lamda_T in [("diffusion1d", 3.0), ("tv", 0.01), ("none", 1.0)]:
        tape = _NoiseTape(170)
        real_randn_like = torch.randn_like
        torch.randn_like = tape
        try:
            with quiet:
                netT2 = ref_ncsn3d.NCSN3DShallow(cfgT).eval()
            netT2.load_state_dict(netT.state_dict())
            sampler = ref_ald.ALD2DTime(ref_prox.get_proximal("L2Penalty")(op), netT2, sigmas_T, (B, T, 1, H, W), net2d,
                                        sigmas, params, cfg2d, meas, op, device=torch.device("cpu"))
            with quiet:
                res = sampler(save_dir="/tmp/ipdm_oracle/out", lr_scaled=1.0e5, mode_T=mode, lamda_T=lamda_T,
                              if_random_shift=False)[0]
        finally:
            torch.randn_like = real_randn_like
            torch.set_grad_enabled(True)
        out[f"{mode}_x"] = npy(res)
        out[f"{mode}_lamda_T"] = np.array(lamda_T)
        # the noise stream is NOT stored (4.7 MB): it is `torch.randn(like.shape, generator=Generator().manual_seed(170))`
        # call after call, which the test regenerates; only its length and a checksum are kept
        out[f"{mode}_noise_calls"] = np.array(len(tape.tape))
        out[f"{mode}_noise_sum"] = np.array(float(sum(float(n.astype(np.float64).sum()) for n in tape.tape)))
    save("g17_ald2dtime", **out)

    # g19: the 2D+time MAP baseline (MAPOptimizer2DTime, MAP_optimizers.py:154-365): 12 Adam iterations, both temporal modes
    ref_map = importlib.import_module("InverseProblemWithDiffusionModel.ncsn.models.MAP_optimizers")
    ref_map.save_vol_as_gif = lambda *a, **k: None
    ref_map.vis_images = lambda *a, **k: None
    ref_map.vis_multi_channel_signal = lambda *a, **k: None
    out = {"measurement": npy(meas)}
    for mode in ["diffusion1d", "tv"]:
        with quiet:
            netT3 = ref_ncsn3d.NCSN3DShallow(cfgT).eval()
        netT3.load_state_dict(netT.state_dict())
        x_init = op.conj_op(meas.reshape(4, B * T, 1, H, W)).reshape(B, T, 1, H, W).clone()
        out[f"{mode}_x_init"] = npy(x_init).copy()
        mp = dict(lr=2e-3, opt_class=torch.optim.Adam, opt_params={"betas": (0.5, 0.5)}, device=torch.device("cpu"),
                  num_iters=12, num_plot_times=1000, win_size=8, prior_weight=0.3, spatial_step_weight=1.0,
                  temporal_step_weight=0.5, save_dir="/tmp/ipdm_oracle/out", mode_T=mode, if_random_shift=False)
        mp["num_plot_times"] = 12                                   # plot_interval = 1
        opt = ref_map.MAPOptimizer2DTime(x_init, meas, net2d, netT3, op, MagicMock(), mp)
        with quiet, contextlib.redirect_stderr(io.StringIO()):
            res = opt()
        torch.set_grad_enabled(True)
        out[f"{mode}_x"] = npy(res)
    out["params"] = np.array([2e-3, 0.3, 1.0, 0.5, 12])               # lr, prior, spatial, temporal weights, iterations
    save("g19_map2dtime", **out)


def g15_fullnet():
    """Full-size ACDC score net on the synthetic weights the benchmark uses."""
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    cfg = tiny_config(ngf=128, num_classes=2311, sigma_begin=348, sigma_end=0.01, image_size=128)
    with quiet:
        net = ref_ncsnv2.NCSNv2Deepest(cfg).eval()
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0)
    sd["sigmas"] = net.state_dict()["sigmas"]
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(15)
    x = torch.rand(2, 1, 128, 128, generator=g)
    x[1] = 40.0 * torch.randn(1, 128, 128, generator=g)      # a high-noise-level input
    labels = torch.tensor([2310, 500])
    with torch.no_grad():
        y = net(x, labels)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    save("g15_fullnet", x=npy(x), labels=npy(labels), y=npy(y),
         key_names=np.array(list(shapes.keys())),
         key_shapes=np.array([",".join(map(str, s)) for s in shapes.values()]))


if __name__ == "__main__":
    which = sys.argv[1:] or None
    torch.set_num_threads(8)
    steps = [("g01", g01_masks), ("g02", g02_sens), ("g03", g03_fft), ("g04", g04_g05_sense_prox),
             ("g06", g06_sigmas), ("g09", g09_upfirdn), ("g10", g10_biasact), ("g11", g11_temporal)]
    for tag, fn in steps:
        if which is None or tag in which:
            fn()
    if which is None or "g07" in which or "g08" in which or "g17" in which:
        net, cfg = g07_layers()
        g08_ald(net, cfg)
        g12_g16_g17_2dtime(net, cfg)
    if which is not None and "g18" in which and not ("g07" in which or "g08" in which or "g17" in which):
        net, cfg = g07_layers()
        g18_map(net, cfg)
    elif which is None or "g18" in which:
        g18_map(net, cfg)
    if which is None or "g13" in which or "g14" in which:
        g13_g14_score_sde()
    if which is None or "g15" in which:
        g15_fullnet()
