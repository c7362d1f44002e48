#!/usr/bin/env python3
"""Multi-coil ALD reconstruction on synthetic k-space -- the MI355X counterpart of the reference's
``scripts/acdc_SENSE_real_img.py`` with the same flags and output artefacts (original.pt, measurement.pt,
reconstructions.pt, ZF.pt, mask.pt, args_dict.pkl).  Data and weights are synthetic (phantom + seeded weights)
unless --ckpt points at a Lightning checkpoint of the reference; samples are sharded over the launched ranks
(torchrun) and rank 0 writes the posterior mean / std next to the reconstructions."""
import argparse
import os
import pickle
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument("--R", type=int, default=40)
    parser.add_argument("--center_lines_frac", type=float, default=1 / 4)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--seg_start_time", type=float, default=0.)      # the reference's default: guidance ramps in from level 0
    parser.add_argument("--seg_step_type", default="linear")
    parser.add_argument("--lamda", type=float, default=0.1)
    parser.add_argument("--step_lr", type=float, default=0.0000009)
    parser.add_argument("--num_steps_each", type=int, default=3)
    parser.add_argument("--lr_scaled", type=float, default=1.)
    parser.add_argument("--proximal_type", default="L2Penalty")
    parser.add_argument("--num_samples", type=int, default=1)
    parser.add_argument("--sens_type", default="exp")
    parser.add_argument("--num_sens", type=int, default=4)
    parser.add_argument("--seg_mode", choices=["full", "FG"], default="full")
    parser.add_argument("--seg_fraction", type=float, default=1.)
    parser.add_argument("--ds_idx", type=int, default=0)
    parser.add_argument("--save_dir", default="../outputs")
    # extras
    parser.add_argument("--image_size", type=int, default=128)
    parser.add_argument("--ckpt", default=None, help="Lightning .ckpt of the reference (EMA weights); default: synthetic")
    parser.add_argument("--n_levels", type=int, default=None, help="run only the first n noise levels")
    parser.add_argument("--seg_ckpt", default=None, help="Lightning TrainSeg .ckpt (MONAI UNet weights) for the guidance")
    parser.add_argument("--seg_synthetic", action="store_true",
                        help="run the guidance with seeded random UNet weights (exercises the path; not meaningful imaging)")
    args_dict = vars(parser.parse_args())
    if args_dict["seg_start_time"] < 1. and not (args_dict["seg_ckpt"] or args_dict["seg_synthetic"]):
        print("no --seg_ckpt: segmentation-likelihood guidance needs trained UNet weights -> running with seg_start_time = 1 "
              "(guidance off); pass --seg_synthetic to exercise the path with random weights")
        args_dict["seg_start_time"] = 1.

    from inverseproblemwithdiffusionmodel_amd import engine, sharding
    world, rank, device = sharding.init_distributed()
    from inverseproblemwithdiffusionmodel_amd.helpers.load_model import load_scorenet_weights

    H = args_dict["image_size"]
    total = args_dict["num_samples"]
    lo, hi = sharding.shard_range(total, world, rank)
    n_local = max(hi - lo, 1)                 # a rank without samples still runs one (discarded) to keep collectives aligned
    cfg = engine.acdc_config(device, H)
    cfg.sampling.step_lr, cfg.sampling.n_steps_each = args_dict["step_lr"], args_dict["num_steps_each"]
    scorenet = engine.build_scorenet(cfg, args_dict["seed"])
    if args_dict["ckpt"]:
        load_scorenet_weights(scorenet, args_dict["ckpt"])
    prob = engine.build_problem(device, n_local, R=args_dict["R"], H=H, W=H, num_sens=args_dict["num_sens"],
                                seed=args_dict["seed"], scorenet=scorenet, cfg=cfg, lr_scaled=args_dict["lr_scaled"])
    label = None
    if args_dict["seg_start_time"] < 1.:
        from inverseproblemwithdiffusionmodel_amd.helpers.load_model import reload_model
        from inverseproblemwithdiffusionmodel_amd.helpers.utils import undersample_seg_mask
        from inverseproblemwithdiffusionmodel_amd.ncsn.models.ALD_optimizers import ALDInvSegProximalRealImag
        seg = reload_model("Seg", "ACDC", ckpt_path=args_dict["seg_ckpt"], device=device)
        label = (prob.image.abs() > 0.5).long()                       # synthetic stand-in for the myocardium label
        label = undersample_seg_mask(label, args_dict["seg_fraction"], seed=args_dict["seed"])
        s0 = prob.sampler
        prob.sampler = ALDInvSegProximalRealImag(s0.proximal, args_dict["seg_start_time"], args_dict["seg_step_type"],
                                                 s0.x_mod_shape, s0.scorenet, s0.sigmas, s0.params, s0.config, s0.measurement,
                                                 s0.linear_tfm, seg=seg, device=device)
    save_dir = args_dict["save_dir"]
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    direct_recons = prob.op.conj_op(prob.measurement[:, :1])

    t0 = time.time()
    kw = dict(prob.call_kwargs, label=label, lamda=args_dict["lamda"], save_dir=save_dir, seg_mode=args_dict["seg_mode"],
              seed=args_dict["seed"], sample_offset=lo, n_levels=args_dict["n_levels"], verbose=(rank == 0))
    img_out = prob.sampler(**kw)[0][: hi - lo]
    torch.cuda.synchronize()
    elapsed = time.time() - t0
    post = sharding.all_reduce_posterior(img_out.to(device), total) if total > 1 else None
    img_out = sharding.gather_samples(img_out.to(device), total, world, rank).cpu()
    if rank == 0:
        resid = prob.op(img_out[:1].to(device)) - prob.measurement[:, :1]
        l2 = torch.sum(torch.abs(resid) ** 2).item()
        err = torch.sqrt(torch.mean(torch.abs(img_out[:1].to(device) - prob.image) ** 2)).item()
        print(f"reconstruction time: {elapsed:.1f} s for {total} sample(s) on {world} GPU(s)")
        print(f"data error ||A x - y||^2 = {l2:.4e}; reconstruction error (RMSE vs phantom) = {err:.4e}")
        torch.save(prob.image.cpu(), os.path.join(save_dir, "original.pt"))
        torch.save(prob.measurement[:, :1].cpu(), os.path.join(save_dir, "measurement.pt"))
        torch.save(img_out.cpu(), os.path.join(save_dir, "reconstructions.pt"))
        torch.save(direct_recons.cpu(), os.path.join(save_dir, "ZF.pt"))
        torch.save(prob.op.random_under_fourier.mask, os.path.join(save_dir, "mask.pt"))
        if post is not None:
            torch.save({k: v.cpu() for k, v in post.items()}, os.path.join(save_dir, "posterior.pt"))
        with open(os.path.join(save_dir, "args_dict.pkl"), "wb") as wf:
            pickle.dump(args_dict, wf)
    if world > 1:
        dist.destroy_process_group()
