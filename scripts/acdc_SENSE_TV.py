#!/usr/bin/env python3
"""Total-variation regularised SENSE reconstruction -- counterpart of the reference's ``scripts/acdc_SENSE_TV.py`` (same
flags; the Lightning trainer around MAPModel is replaced by MAPModel.fit: one Adam step per epoch on the HIP kernels).
Synthetic phantom k-space; artefacts as the reference writes them."""
import argparse
import os
import pickle
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument("--R", type=int, default=5)
    parser.add_argument("--center_lines_frac", type=float, default=1 / 20)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--sens_type", default="exp")
    parser.add_argument("--num_sens", type=int, default=4)
    parser.add_argument("--ds_idx", type=int, default=0)
    parser.add_argument("--num_workers", type=int, default=0)
    parser.add_argument("--lr", type=float, default=1e-2)
    parser.add_argument("--num_epochs", type=int, default=200)
    parser.add_argument("--reg_weight", type=float, default=0.01)
    parser.add_argument("--save_dir", default="../outputs")
    parser.add_argument("--log_dir", default="SENSE")
    parser.add_argument("--image_size", type=int, default=128)
    args_dict = vars(parser.parse_args())

    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import SENSE
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.MAP_optimizers import MAPModel, TotalVariation
    from inverseproblemwithdiffusionmodel_amd.synthetic import phantom_image
    device = torch.device("cuda")
    H = args_dict["image_size"]
    try:
        op = SENSE(args_dict["sens_type"], args_dict["num_sens"], args_dict["R"], args_dict["center_lines_frac"], (1, H, H),
                   args_dict["seed"])
    except ValueError:
        # the reference's live mask generator ignores R (hard-wired T = 24 parameter set, undersampling_fourier.py:68-73):
        # an R without its own parameter set (the script's default 5) gets exactly that mask
        print(f"no mask parameter set for R={args_dict['R']}: using the reference's live T=24 mask")
        op = SENSE(args_dict["sens_type"], args_dict["num_sens"], args_dict["R"], args_dict["center_lines_frac"], (1, H, H),
                   args_dict["seed"], mask_T=24)
    img_complex = phantom_image(H, H, seed=args_dict["seed"] + args_dict["ds_idx"]).to(device)      # (1, 1, H, W)
    measurement = op(img_complex)                                                                   # (num_sens, 1, 1, H, W)
    model = MAPModel(measurement, op, TotalVariation(), args_dict["reg_weight"], device=device)
    direct_recons = op.conj_op(measurement)
    original_error = torch.sum(torch.abs(op(direct_recons) - measurement) ** 2).item()
    t0 = time.time()
    img_out = model.fit(args_dict["num_epochs"], args_dict["lr"])
    torch.cuda.synchronize()
    elapsed = time.time() - t0
    l2_error = torch.sum(torch.abs(op(img_out.to(device)) - measurement) ** 2).item()
    data_loss, reg_loss, loss = (float(v) for v in model.forward())
    print("-" * 100)
    print(args_dict)
    print(f"original error: {original_error}")
    print(f"reconstruction error: {l2_error}")
    print(f"{args_dict['num_epochs']} epochs in {elapsed:.2f} s; data loss {data_loss:.4e}, TV {reg_loss:.4e}, loss {loss:.4e}")
    save_dir = args_dict["save_dir"]
    os.makedirs(save_dir, exist_ok=True)
    torch.save(img_complex.cpu(), os.path.join(save_dir, "original.pt"))
    torch.save(measurement.cpu(), os.path.join(save_dir, "measurement.pt"))
    torch.save(direct_recons.cpu(), os.path.join(save_dir, "ZF.pt"))
    torch.save(img_out.cpu(), os.path.join(save_dir, "reconstructions.pt"))
    with open(os.path.join(save_dir, "args_dict.pkl"), "wb") as wf:
        pickle.dump(args_dict, wf)
