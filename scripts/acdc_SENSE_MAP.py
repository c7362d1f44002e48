#!/usr/bin/env python3
"""MAP baseline on synthetic k-space -- the MI355X counterpart of the reference's ``scripts/acdc_SENSE_MAP.py`` (same
flags): Adam ascent on  log p(y | x) + lamda * log p(x)  with the NCSNv2 score at noise level 1 as the prior gradient.
Writes original.pt, measurement.pt, reconstructions.pt, ZF.pt, mask.pt, args_dict.pkl."""
import argparse
import os
import pickle
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument("--R", type=int, default=8)
    parser.add_argument("--center_lines_frac", type=float, default=1 / 4)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--sens_type", default="exp")
    parser.add_argument("--num_sens", type=int, default=4)
    parser.add_argument("--ds_idx", type=int, default=0)
    parser.add_argument("--save_dir", default="../outputs")
    parser.add_argument("--lamda", type=float, default=1e-2)
    parser.add_argument("--n_iters", type=int, default=None, help="default: config.MAP.n_iters")
    args_dict = vars(parser.parse_args())

    from inverseproblemwithdiffusionmodel_amd import engine
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import SENSE
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.MAP_optimizers import SENSEMAP
    from inverseproblemwithdiffusionmodel_amd.synthetic import phantom_image

    device = torch.device("cuda")
    cfg = engine.acdc_config(device, 128)
    if args_dict["n_iters"]:
        cfg.MAP.n_iters = args_dict["n_iters"]
    scorenet = engine.build_scorenet(cfg, args_dict["seed"])
    op = SENSE(args_dict["sens_type"], args_dict["num_sens"], args_dict["R"], args_dict["center_lines_frac"], (1, 128, 128),
               args_dict["seed"])
    img = phantom_image(128, 128, seed=args_dict["seed"] + args_dict["ds_idx"]).to(device)
    measurement = op(img)
    x_init = op.conj_op(measurement).clone()
    zf = x_init.clone()
    t0 = time.time()
    x = SENSEMAP(x_init, measurement, scorenet, op, args_dict["lamda"], cfg, logger=None, device=device)()
    torch.cuda.synchronize()
    err0 = torch.sqrt(torch.mean(torch.abs(zf - img) ** 2)).item()
    err1 = torch.sqrt(torch.mean(torch.abs(x - img) ** 2)).item()
    print(f"{cfg.MAP.n_iters} MAP iterations in {time.time() - t0:.2f} s; RMSE vs phantom: zero-filled {err0:.4e} -> MAP {err1:.4e}")
    save_dir = args_dict["save_dir"]
    os.makedirs(save_dir, exist_ok=True)
    torch.save(img.cpu(), os.path.join(save_dir, "original.pt"))
    torch.save(measurement.cpu(), os.path.join(save_dir, "measurement.pt"))
    torch.save(x.cpu(), os.path.join(save_dir, "reconstructions.pt"))
    torch.save(zf.cpu(), os.path.join(save_dir, "ZF.pt"))
    torch.save(op.random_under_fourier.mask, os.path.join(save_dir, "mask.pt"))
    with open(os.path.join(save_dir, "args_dict.pkl"), "wb") as wf:
        pickle.dump(args_dict, wf)
