"""Assembly of the headline workload: ACDC-style multi-coil ALD reconstruction on synthetic k-space
(128x128 complex phantom, 4 'exp' coils, variable-density mask, NCSNv2Deepest ngf=128 with
seeded random-init weights, sigma 348 -> 0.01 in 2311 levels x 3 steps, L2Penalty proximal, denoise).

Used by bench.py, __graft_entry__.smoke(), scripts/ and the tests, so that they all run the same thing.
"""
from argparse import Namespace

import numpy as np
import torch

from . import ops as ops_mod

from .helpers.load_data import load_config
from .ncsn.linear_transforms.undersampling_fourier import SENSE
from .ncsn.models import get_sigmas
from .ncsn.models.ALD_optimizers import ALDInvSegProximalRealImag, SCHED_DTYPE, step_schedule
from .ncsn.models.ncsnv2 import NCSNv2Deepest
from .ncsn.models.proximal_op import get_proximal
from .synthetic import phantom_image, synth_state_dict


def conv_census(net, x, labels):
    """run one eager forward with per-launch HIP events around every convolution.
    -> list of dict(B, Cin, Cout, H, W, k, dil, flops, ms): the exact 2*MAC count and the measured duration of
    each conv launch (events are recorded on the stream the kernels run on)."""
    from . import ops
    ops.CONV_TRACE = []
    try:
        with torch.no_grad():
            net(x, labels)
        torch.cuda.synchronize()
        rec = ops.CONV_TRACE
    finally:
        ops.CONV_TRACE = None
    out = []
    for r in rec:
        ms = r.pop("e0").elapsed_time(r.pop("e1"))
        r["flops"] = 2 * r["B"] * r["Cin"] * r["Cout"] * r["k"] ** 2 * r["H"] * r["W"]
        r["ms"] = ms
        out.append(r)
    return out


def acdc_config(device, image_size=128):
    cfg = load_config("ACDC", mode="real-valued", device=device)
    cfg.data.image_size = image_size
    return cfg


def build_scorenet(cfg, seed=0):
    net = NCSNv2Deepest(cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=seed)
    net.load_state_dict(sd, strict=False)
    return net.to(cfg.device).eval()


def build_problem(device, n_samples, R=40, H=128, W=128, num_sens=4, seed=0, scorenet=None, cfg=None, lr_scaled=1.0):
    """-> Namespace(sampler, scorenet, sigmas, op, image, measurement, call_kwargs)"""
    cfg = acdc_config(device, H) if cfg is None else cfg
    scorenet = build_scorenet(cfg, seed) if scorenet is None else scorenet
    sigmas = get_sigmas(cfg, "recons")
    op = SENSE("exp", num_sens, R, 0.04, (1, H, W), seed=seed, mask_T=1)
    img = phantom_image(H, W, seed=seed).to(device)
    meas = op(img).repeat(1, n_samples, 1, 1, 1).contiguous()
    params = dict(n_steps_each=cfg.sampling.n_steps_each, step_lr=cfg.sampling.step_lr, denoise=True,
                  final_only=True)
    sampler = ALDInvSegProximalRealImag(get_proximal("L2Penalty")(op), 1.0, "linear", (n_samples, 1, H, W), scorenet,
                                        sigmas, params, cfg, meas, op, seg=None, device=device)
    return Namespace(sampler=sampler, scorenet=scorenet, sigmas=sigmas, op=op, image=img, measurement=meas, cfg=cfg,
                     params=params, call_kwargs=dict(label=None, lamda=0.1, save_dir=None, lr_scaled=lr_scaled,
                                                     seg_mode="full"))


class IterationRunner:
    """One Langevin iteration of the SENSE sampler as a replayable hipGraph, with the level chosen per call.
    This is the unit bench.py times ("step"); ALDInvSegProximalRealImag.__call__ runs the same launches."""

    def __init__(self, prob, seed=0, sample_offset=0, use_graph=True):
        s = prob.sampler
        dev = s.device
        meas = s.measurement.to(dev).to(torch.complex64).contiguous()
        x0 = s.linear_tfm.conj_op(meas)
        B, H, W = x0.shape[0], x0.shape[-2], x0.shape[-1]
        self.B, self.H, self.W = B, H, W
        self.sampler = s
        self.x = torch.cat([x0.real, x0.imag], dim=0).contiguous().float()
        steps, noise_scales = step_schedule(s.sigmas, s.params["step_lr"])
        L, n_each = len(s.sigmas), s.params["n_steps_each"]
        coef = s.proximal.coef(s.params["step_lr"] * prob.call_kwargs["lr_scaled"], 1., x0.shape)
        table = np.zeros(L * n_each, dtype=SCHED_DTYPE)
        lv = np.repeat(np.arange(L), n_each)
        table["step"], table["noise_scale"] = steps.numpy()[lv], noise_scales.numpy()[lv]
        table["coef"], table["sigma"] = coef, s.sigmas.detach().cpu().numpy()[lv]
        table["step_id"] = np.arange(L * n_each)
        self.levels = lv
        self.table_dev = torch.from_numpy(table.view(np.uint8).reshape(L * n_each, -1).copy()).to(dev)
        self.label_table = torch.arange(L, device=dev)[:, None].repeat(1, 2 * B)
        self.st = dict(x=self.x, B=B, y=meas, sc_mode=None, sens=s.linear_tfm.sens_f32(dev), mask=s.linear_tfm.mask_u8(dev),
                       work=ops_mod.sense_workspace(B, s.linear_tfm.sens_maps.shape[0], H, W, dev),
                       labels=torch.zeros(2 * B, dtype=torch.long, device=dev), noise_re=None, noise_im=None,
                       seed=seed, sample_offset=sample_offset,
                       sched_dev=torch.zeros(SCHED_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        self.n_iterations = L * n_each
        self.graph = None
        self.use_graph = use_graph

    def set_iteration(self, k):
        self.st["sched_dev"].copy_(self.table_dev[k], non_blocking=True)
        self.st["labels"].copy_(self.label_table[int(self.levels[k])], non_blocking=True)

    @torch.no_grad()
    def run(self, k):
        self.set_iteration(k)
        if not self.use_graph:
            self.sampler._iteration(self.st)
        elif self.graph is None:
            self.sampler._iteration(self.st)           # warm-up: weight packing, allocator, LDS attributes
            self.graph = self.sampler._capture(self.st)
        else:
            self.graph.replay()

    def current(self):
        return torch.complex(self.x[:self.B], self.x[self.B:])
