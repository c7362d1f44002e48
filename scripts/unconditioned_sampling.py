#!/usr/bin/env python3
"""Unconditional ALD sampling (counterpart of the reference's ``scripts/unconditioned_sampling.py``, BASELINE config 1:
MNIST-shaped 32x32 NCSNv2Deepest, 232 noise levels)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument("--ds_name", default="MNIST")
    parser.add_argument("--num_steps_each", type=int, default=3)
    parser.add_argument("--num_samples", type=int, default=1)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--save_dir", default="../outputs")
    args = parser.parse_args()
    from inverseproblemwithdiffusionmodel_amd.helpers.load_model import reload_model
    from inverseproblemwithdiffusionmodel_amd.ncsn.models import get_sigmas
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.ALD_optimizers import ALDUnconditionalSampler
    device = torch.device("cuda")
    net = reload_model("Diffusion", args.ds_name, device=device)
    cfg = net.config
    sigmas = get_sigmas(cfg, "recons")
    params = dict(n_steps_each=args.num_steps_each, step_lr=cfg.sampling.step_lr, denoise=True, final_only=True)
    sampler = ALDUnconditionalSampler((args.num_samples, cfg.data.channels, cfg.data.image_size, cfg.data.image_size),
                                      net, sigmas, params, cfg, device=device)
    t0 = time.time()
    out = sampler(seed=args.seed)[0]
    torch.cuda.synchronize()
    print(f"{args.num_samples} sample(s), {len(sigmas) * args.num_steps_each + 1} score evaluations in {time.time() - t0:.2f} s")
    os.makedirs(args.save_dir, exist_ok=True)
    torch.save(out, os.path.join(args.save_dir, "unconditioned_samples.pt"))
