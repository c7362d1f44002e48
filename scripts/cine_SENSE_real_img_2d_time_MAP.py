#!/usr/bin/env python3
"""2D+time MAP baseline on synthetic k-space -- counterpart of the reference's ``scripts/cine_SENSE_real_img_2d_time_MAP.py``
(same flags; Adam only -- the reference hard-wires opt_class = torch.optim.Adam too, :68): MAPOptimizer2DTime with the
spatial NCSNv2Deepest and the temporal NCSN3DShallow prior.  Artefacts as the reference writes them (no GIFs / PNG panels)."""
import argparse
import os
import pickle
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument("--ds_name", choices=["CINE64", "CINE127"], default="CINE127")
    parser.add_argument("--R", type=int, default=6)
    parser.add_argument("--center_lines_frac", type=float, default=1 / 4)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--lr", type=float, default=0.001)
    parser.add_argument("--num_iters", type=int, default=200)
    parser.add_argument("--num_plot_times", type=int, default=10)
    parser.add_argument("--prior_weight", type=float, default=1.)
    parser.add_argument("--spatial_step_weight", type=float, default=1.)
    parser.add_argument("--temporal_step_weight", type=float, default=1.)
    parser.add_argument("--num_samples", type=int, default=1)
    parser.add_argument("--sens_type", default="exp")
    parser.add_argument("--temporal_type", default="Diffusion3D")
    parser.add_argument("--num_sens", type=int, default=4)
    parser.add_argument("--mode_T", choices=["diffusion1d", "tv"], default="diffusion1d")
    parser.add_argument("--if_random_shift", action="store_true")
    parser.add_argument("--ds_idx", type=int, default=0)
    parser.add_argument("--save_dir", default="../outputs")
    parser.add_argument("--beta1", type=float, default=0.9)
    parser.add_argument("--beta2", type=float, default=0.999)
    parser.add_argument("--max_iter", type=int, default=20)          # (LBFGS option of the reference; unused: Adam only)
    args_dict = vars(parser.parse_args())

    from inverseproblemwithdiffusionmodel_amd.helpers.load_model import reload_model
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import SENSE
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.MAP_optimizers import MAPOptimizer2DTime
    from inverseproblemwithdiffusionmodel_amd.synthetic import phantom_image
    device = torch.device("cuda")
    np.random.seed(args_dict["seed"])
    ds_name = args_dict["ds_name"]
    scorenet = reload_model("Diffusion", ds_name, device=device)
    scorenet_T = reload_model(args_dict["temporal_type"], f"{ds_name}_1D", device=device)
    T, C = scorenet_T.config.data.image_size, scorenet.config.data.channels
    H = W = scorenet.config.data.image_size
    B = args_dict["num_samples"]
    op = SENSE(args_dict["sens_type"], args_dict["num_sens"], args_dict["R"], args_dict["center_lines_frac"], (C, H, W),
               args_dict["seed"], mask_T=24 if T == 24 else 1)
    base = phantom_image(H, W, seed=args_dict["seed"] + args_dict["ds_idx"]).to(device)
    beat = torch.cos(torch.arange(T, device=device) * (2 * torch.pi / T)).view(T, 1, 1, 1)
    img_complex = (base * (1.0 + 0.1 * beat)).to(torch.complex64)                 # (T, 1, H, W)
    measurement = op(img_complex).unsqueeze(1).repeat(1, B, 1, 1, 1, 1)           # (num_sens, B, T, 1, H, W)
    direct_recons = op.conj_op(measurement.reshape(measurement.shape[0], B * T, C, H, W)).reshape(B, T, C, H, W)
    params = {"lr": args_dict["lr"], "opt_class": torch.optim.Adam, "num_iters": args_dict["num_iters"],
              "num_plot_times": args_dict["num_plot_times"], "win_size": int(np.sqrt(scorenet_T.config.data.channels)),
              "prior_weight": args_dict["prior_weight"], "spatial_step_weight": args_dict["spatial_step_weight"],
              "temporal_step_weight": args_dict["temporal_step_weight"], "save_dir": args_dict["save_dir"],
              "opt_params": {"betas": (args_dict["beta1"], args_dict["beta2"])}, "mode_T": args_dict["mode_T"],
              "if_random_shift": args_dict["if_random_shift"], "device": device}
    opt = MAPOptimizer2DTime(direct_recons.clone(), measurement, scorenet, scorenet_T, op, None, params)
    t0 = time.time()
    img_out = opt()                                                               # (B, T, C, H, W)
    torch.cuda.synchronize()
    elapsed = time.time() - t0
    resid = op(img_out.to(device).reshape(B * T, C, H, W)) - measurement.reshape(measurement.shape[0], B * T, C, H, W)
    l2_error = torch.sum(torch.abs(resid) ** 2, dim=(1, 2, 3)).mean().item()
    print("-" * 100)
    print(args_dict)
    print(f"reconstruction error: {l2_error}")
    print(f"reconstruction time: {elapsed}")
    save_dir = args_dict["save_dir"]
    os.makedirs(save_dir, exist_ok=True)
    torch.save(img_complex.cpu(), os.path.join(save_dir, "original.pt"))
    torch.save(measurement.cpu(), os.path.join(save_dir, "measurement.pt"))
    torch.save(direct_recons.cpu(), os.path.join(save_dir, "ZF.pt"))
    torch.save(img_out.cpu(), os.path.join(save_dir, "reconstructions.pt"))
    torch.save(op.random_under_fourier.mask, os.path.join(save_dir, "mask.pt"))
    with open(os.path.join(save_dir, "args_dict.pkl"), "wb") as wf:
        pickle.dump(args_dict, wf)
