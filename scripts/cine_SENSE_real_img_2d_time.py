#!/usr/bin/env python3
"""2D+time ALD reconstruction on synthetic k-space (counterpart of the reference's
``scripts/cine_SENSE_real_img_2d_time.py``, BASELINE config 4): spatial NCSNv2Deepest prior + temporal
NCSN3DShallow prior on 8x8xT patches, SENSE with the T=24 mask.  Same flags; prints `reconstruction time` as the
reference does.  Under torchrun the `--num_samples` posterior samples are block-partitioned over the ranks (Philox noise
keyed by the global sample id, the per-step random shift drawn from identically seeded host generators), rank 0 writes the
artefacts and the posterior mean / std: the result does not depend on the number of ranks."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument("--R", type=int, default=8)
    parser.add_argument("--center_lines_frac", type=float, default=1 / 20)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--step_lr", type=float, default=0.0001)
    parser.add_argument("--num_steps_each", type=int, default=3)
    parser.add_argument("--lr_scaled", type=float, default=1.)
    parser.add_argument("--proximal_type", default="L2Penalty")
    parser.add_argument("--num_samples", type=int, default=1)
    parser.add_argument("--sens_type", default="exp")
    parser.add_argument("--num_sens", type=int, default=4)
    parser.add_argument("--mode_T", default="diffusion1d", choices=["tv", "diffusion1d", "none", "diffusion1d-only", "tv-only"])
    parser.add_argument("--lamda_T", type=float, default=10.)
    parser.add_argument("--if_random_shift", action="store_true")
    parser.add_argument("--save_dir", default="../outputs")
    parser.add_argument("--image_size", type=int, default=128)
    parser.add_argument("--T", type=int, default=24)
    parser.add_argument("--n_levels", type=int, default=None)
    parser.add_argument("--start_level", type=int, default=0, help="first noise level (with --n_levels: a slice of the schedule)")
    a = parser.parse_args()
    from inverseproblemwithdiffusionmodel_amd.helpers.load_model import reload_model
    from inverseproblemwithdiffusionmodel_amd.ncsn.models import get_sigmas
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.ALD_optimizers import ALD2DTime
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.proximal_op import get_proximal
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import SENSE
    from inverseproblemwithdiffusionmodel_amd.synthetic import phantom_image
    from inverseproblemwithdiffusionmodel_amd import sharding
    world, rank, device = sharding.init_distributed()
    lo, hi = sharding.shard_range(a.num_samples, world, rank)
    n_local = max(hi - lo, 1)                 # a rank without samples still runs one (discarded): collectives stay aligned
    np.random.seed(a.seed)                    # if_random_shift: one shift per step for the whole batch, on every rank
    H = W = a.image_size
    scorenet = reload_model("Diffusion", "CINE127", device=device)
    scorenet_T = reload_model("Diffusion3D", "CINE127", device=device)
    sigmas = get_sigmas(scorenet.config, "recons")
    sigmas_T = get_sigmas(scorenet_T.config, "recons")
    op = SENSE(a.sens_type, a.num_sens, a.R, a.center_lines_frac, (1, H, W), a.seed, mask_T=24 if a.T == 24 else 1)
    base = phantom_image(H, W, seed=a.seed).to(device)
    beat = torch.cos(torch.arange(a.T, device=device) * (2 * torch.pi / a.T)).view(a.T, 1, 1, 1)
    frames = base * (1.0 + 0.1 * beat)                                     # (T, 1, H, W): a slowly pulsating phantom
    meas = op(frames).reshape(a.num_sens, 1, a.T, 1, H, W).repeat(1, n_local, 1, 1, 1, 1)
    params = dict(n_steps_each=a.num_steps_each, step_lr=a.step_lr, denoise=False, final_only=True)
    sampler = ALD2DTime(get_proximal(a.proximal_type)(op), scorenet_T, sigmas_T, (n_local, a.T, 1, H, W), scorenet,
                        sigmas, params, scorenet.config, meas, op, device=device)
    t0 = time.time()
    out = sampler(save_dir=a.save_dir, lr_scaled=a.lr_scaled, mode_T=a.mode_T, lamda_T=a.lamda_T,
                  if_random_shift=a.if_random_shift, seed=a.seed, sample_offset=lo, n_levels=a.n_levels, start_level=a.start_level,
                  verbose=(rank == 0))[0][: hi - lo]
    torch.cuda.synchronize()
    if rank == 0:
        print(f"reconstruction time: {time.time() - t0}")
    B, T = out.shape[:2]
    flat = out.to(device).reshape(out.shape[0], -1, H, W)                  # (n_local, T, H, W): frames as "channels"
    post = sharding.all_reduce_posterior(flat, a.num_samples) if a.num_samples > 1 else None
    out = sharding.gather_samples(out.to(device), a.num_samples, world, rank).cpu()
    if rank == 0:
        os.makedirs(a.save_dir, exist_ok=True)
        torch.save(out, os.path.join(a.save_dir, "reconstructions.pt"))
        torch.save(frames.cpu(), os.path.join(a.save_dir, "original.pt"))
        torch.save(op.random_under_fourier.mask, os.path.join(a.save_dir, "mask.pt"))
        if post is not None:
            torch.save({k: v.cpu() for k, v in post.items()}, os.path.join(a.save_dir, "posterior.pt"))
    if world > 1:
        sharding.barrier(last=True)
        torch.distributed.destroy_process_group()
