#!/usr/bin/env python3
"""Timings of BASELINE.json's other configurations on one GPU (not bench lines -- the bench is config 2/3's
iteration; these are recorded in DESIGN.md):
  1  MNIST-shaped unconditional ALD sampling, 1 sample, full 232 x 3 schedule
  2  ACDC 128x128 R=20 4-coil, 8 samples: Langevin iterations/s
  4  CINE 2D+time 128x128x24, R=8 masks, spatial + temporal prior: seconds per noise level (1 sample)
  5  NCSN++ 256x256 (celebahq_256_ncsnpp_continuous): score forward and one predictor-corrector step, batch 8
Usage: python scripts/bench_configs.py [1 2 4 5]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from inverseproblemwithdiffusionmodel_amd import engine                                             # noqa: E402
from inverseproblemwithdiffusionmodel_amd.helpers.load_model import reload_model                     # noqa: E402
from inverseproblemwithdiffusionmodel_amd.ncsn.models import get_sigmas                              # noqa: E402
from inverseproblemwithdiffusionmodel_amd.synthetic import phantom_image, synth_state_dict           # noqa: E402

dev = torch.device("cuda")


def timed(fn, n=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def cfg1():
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.ALD_optimizers import ALDUnconditionalSampler
    net = reload_model("Diffusion", "MNIST", device=dev)
    cfg = net.config
    sigmas = get_sigmas(cfg, "recons")
    params = dict(n_steps_each=3, step_lr=cfg.sampling.step_lr, denoise=True, final_only=True)
    smp = ALDUnconditionalSampler((1, cfg.data.channels, cfg.data.image_size, cfg.data.image_size), net, sigmas, params,
                                  cfg, device=dev)
    smp(seed=0)
    s = timed(lambda: smp(seed=1), n=1, warm=0)
    return dict(config=1, levels=len(sigmas), seconds_per_sample=s, samples_per_s=1 / s)


def cfg2():
    prob = engine.build_problem(dev, 8, R=20)
    run = engine.IterationRunner(prob)
    for k in range(3):
        run.run(k)
    s = timed(lambda: run.run(100), n=10)
    return dict(config=2, samples=8, ms_per_iteration=s * 1e3, reconstructions_per_s=8 / (s * run.n_iterations + s))


def cfg4():
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.ALD_optimizers import ALD2DTime
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.proximal_op import get_proximal
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import SENSE
    H = W = 128
    T = 24
    net2d = reload_model("Diffusion", "CINE127", device=dev)
    net3d = reload_model("Diffusion3D", "CINE127", device=dev)
    sig, sigT = get_sigmas(net2d.config, "recons"), get_sigmas(net3d.config, "recons")
    op = SENSE("exp", 4, 8, 1 / 20, (1, H, W), 0, mask_T=24)
    frames = phantom_image(H, W, seed=0).to(dev).repeat(T, 1, 1, 1)
    meas = op(frames).reshape(4, 1, T, 1, H, W)
    params = dict(n_steps_each=3, step_lr=1e-4, denoise=False, final_only=True)
    smp = ALD2DTime(get_proximal("L2Penalty")(op), net3d, sigT, (1, T, 1, H, W), net2d, sig, params, net2d.config, meas,
                    op, device=dev)
    L = len(sig)
    kw = dict(save_dir=None, lr_scaled=1.0, mode_T="diffusion1d", lamda_T=10., if_random_shift=False, seed=0)
    smp(start_level=L - 2, n_levels=1, **kw)
    n_T = int((smp.sigmas_T > 0).sum())
    s_T = timed(lambda: smp(start_level=L - 2, n_levels=1, **kw), n=2, warm=0)          # level with temporal prior
    s_S = timed(lambda: smp(start_level=0, n_levels=1, **kw), n=2, warm=0)              # spatial-only level
    total = s_T * n_T + s_S * (L - n_T)
    return dict(config=4, levels=L, levels_with_temporal_prior=n_T, s_per_level_spatial=s_S,
                s_per_level_spatial_temporal=s_T, est_seconds_per_reconstruction=total)


def cfg5():
    from inverseproblemwithdiffusionmodel_amd.configs.ve_ncsnpp import get_config
    from inverseproblemwithdiffusionmodel_amd.models import ncsnpp
    from inverseproblemwithdiffusionmodel_amd.sde import sde_lib, sampling
    cfg = get_config()
    net = ncsnpp.NCSNpp(cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0)
    net.load_state_dict(sd, strict=False)
    net = net.to(dev).eval()
    B = 8
    x = torch.randn(B, 3, 256, 256, device=dev) * 50
    t = torch.full((B,), 10.0, device=dev)
    with torch.no_grad():
        s_fwd = timed(lambda: net(x, t), n=3)
        # the same forward replayed as one hipGraph (what a sampler that captures its score evaluation pays)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y_static = net(x, t)
        s_graph = timed(g.replay, n=5)
    if os.environ.get("IPDM_CFG5_FORWARD_ONLY") == "1":
        return dict(config=5, batch=B, s_per_forward=s_fwd, s_per_forward_graph=s_graph)
    sde = sde_lib.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    pred = sampling.get_predictor("reverse_diffusion")
    corr = sampling.get_corrector("langevin")
    vt = torch.full((B,), 0.5, device=dev)

    def pc():
        with torch.no_grad():
            xc, _ = sampling.shared_corrector_update_fn(x, vt, sde, net, corr, True, cfg.sampling.snr, 1)
            sampling.shared_predictor_update_fn(xc, vt, sde, net, pred, False, True)
    s_pc = timed(pc, n=2)
    n_par = sum(p.numel() for p in net.parameters())
    return dict(config=5, batch=B, params_M=n_par / 1e6, s_per_forward=s_fwd, s_per_forward_graph=s_graph,
                images_per_s_forward=B / s_graph,
                s_per_pc_step=s_pc, est_seconds_per_batch_2000_steps=s_pc * cfg.model.num_scales)


if __name__ == "__main__":
    which = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 5]
    for c in which:
        t0 = time.time()
        r = {1: cfg1, 2: cfg2, 4: cfg4, 5: cfg5}[c]()
        r["wall_s"] = round(time.time() - t0, 1)
        print(json.dumps(r), flush=True)
