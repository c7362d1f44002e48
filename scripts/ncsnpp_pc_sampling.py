#!/usr/bin/env python3
"""Predictor-corrector sampling with the NCSN++ VE score network (BASELINE config 5: configs/ve celebahq_256 NCSN++,
`sde/sampling.py:360-416` get_pc_sampler with ReverseDiffusionPredictor + LangevinCorrector) -- the reference has no
driver script for it (score_sde's run_lib is not part of the repository); flags follow the config's `sampling` block.
Under torchrun the `--num_samples` samples are block-partitioned over the ranks: prior draws from one seeded host
generator (every rank draws the whole batch and keeps its block), Philox noise keyed by the global sample id, and the
LangevinCorrector's two batch means taken over ALL samples (sampling.set_shard: one small all-reduce per mean) -- the
result does not depend on the number of ranks.  Weights: --ckpt (a state dict) or seeded synthetic."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_samples", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--snr", type=float, default=None, help="default: config.sampling.snr")
    ap.add_argument("--n_steps_each", type=int, default=None, help="corrector steps per level (default: config)")
    ap.add_argument("--predictor", default=None)
    ap.add_argument("--corrector", default=None)
    ap.add_argument("--n_iters", type=int, default=None, help="run only the first n of the config's num_scales levels")
    ap.add_argument("--image_size", type=int, default=None, help="override config.data.image_size")
    ap.add_argument("--tiny", action="store_true", help="a 4-level, nf-16 network (tests)")
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--save_dir", default="../outputs")
    a = ap.parse_args()

    from inverseproblemwithdiffusionmodel_amd import sharding
    from inverseproblemwithdiffusionmodel_amd.configs import ve_ncsnpp
    from inverseproblemwithdiffusionmodel_amd.models import ncsnpp
    from inverseproblemwithdiffusionmodel_amd.sde import sde_lib, sampling
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    world, rank, device = sharding.init_distributed()
    cfg = ve_ncsnpp.get_config()
    cfg.device = device
    if a.tiny:
        cfg.model.nf, cfg.model.ch_mult, cfg.model.num_res_blocks, cfg.model.attn_resolutions = 16, (1, 2, 2), 1, (8,)
        cfg.data.image_size = 32
        cfg.model.num_scales = 20
    if a.image_size:
        cfg.data.image_size = a.image_size
    for k in ("snr", "n_steps_each", "predictor", "corrector"):
        if getattr(a, k) is not None:
            setattr(cfg.sampling, k, getattr(a, k))
    net = ncsnpp.NCSNpp(cfg)
    if a.ckpt:
        net.load_state_dict(torch.load(a.ckpt, map_location="cpu"))
    else:
        net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=a.seed), strict=False)
    net = net.to(device).eval()
    sde = sde_lib.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    total = a.num_samples
    lo, hi = sharding.shard_range(total, world, rank)
    n_local = max(hi - lo, 1)
    C, S = cfg.data.num_channels, cfg.data.image_size
    torch.manual_seed(a.seed)
    prior = sde.prior_sampling((total, C, S, S))                 # the whole batch on every rank, one host generator
    x_init = prior[lo:hi] if hi > lo else prior[:1]
    sampling.set_noise_source(None, seed=a.seed, sample_offset=lo)
    sampling.set_shard(total if world > 1 else None, lo)
    if world > 1 and hi == lo:
        raise SystemExit("more ranks than samples: LangevinCorrector's batch means need every rank to own >= 1 sample")
    sampler = sampling.get_sampling_fn(cfg, sde, (n_local, C, S, S), lambda x: x, 1e-5)
    t0 = time.time()
    x, nfe = sampler(net, n_iters=a.n_iters, x_init=x_init)
    torch.cuda.synchronize()
    elapsed = time.time() - t0
    x = sharding.gather_samples(x[: hi - lo].contiguous(), total, world, rank).cpu()
    if rank == 0:
        n_it = cfg.model.num_scales if a.n_iters is None else a.n_iters
        print(f"sampling time: {elapsed:.1f} s for {total} sample(s) on {world} GPU(s), {n_it} predictor-corrector levels")
        os.makedirs(a.save_dir, exist_ok=True)
        torch.save(x, os.path.join(a.save_dir, "samples.pt"))
        torch.save(prior, os.path.join(a.save_dir, "prior.pt"))
    if world > 1:
        sharding.barrier(last=True)
        torch.distributed.destroy_process_group()
