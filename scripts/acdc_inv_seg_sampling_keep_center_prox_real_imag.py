#!/usr/bin/env python3
"""Single-coil ALD reconstruction -- the MI355X counterpart of the reference's
``scripts/acdc_inv_seg_sampling_keep_center_prox_real_imag.py`` (and, with ``--dataset CINE64``, of
``scripts/cine_inv_sampling_keep_center_prox_real_imag.py``): RandomUndersamplingFourier + get_proximal(--proximal_type)
(L2Penalty or SingleCoil) inside ALDInvSegProximalRealImag, same flags, the fused single-coil iteration tail on the GPU.
Data are synthetic (phantom + smooth phase, ``add_phase`` semantics) unless --data_dir points at ACDC slice .npz files /
a CINE .mat; weights are seeded synthetic unless --ckpt / --seg_ckpt are given.  Writes original.pt, measurement.pt,
reconstructions.pt, ZF.pt, mask.pt, args_dict.pkl."""
import argparse
import os
import pickle
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(default_dataset="ACDC"):
    parser = argparse.ArgumentParser()
    parser.add_argument("--R", type=int, default=6)
    parser.add_argument("--center_lines_frac", type=float, default=1 / 4)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--seg_start_time", type=float, default=0.)
    parser.add_argument("--seg_step_type", default="linear")
    parser.add_argument("--lamda", type=float, default=0.1)
    parser.add_argument("--step_lr", type=float, default=0.0000009)
    parser.add_argument("--num_steps_each", type=int, default=3)
    parser.add_argument("--lr_scaled", type=float, default=1.)
    parser.add_argument("--proximal_type", default="L2Penalty")
    parser.add_argument("--num_samples", type=int, default=1)
    parser.add_argument("--ds_idx", type=int, default=0)
    parser.add_argument("--save_dir", default="../outputs")
    # extras
    parser.add_argument("--dataset", default=default_dataset, choices=["ACDC", "CINE64"])
    parser.add_argument("--image_size", type=int, default=128)
    parser.add_argument("--data_dir", default=None, help="ACDC slice .npz directory / CINE .mat directory (default: phantom)")
    parser.add_argument("--ckpt", default=None, help="Lightning .ckpt of the score network (EMA weights)")
    parser.add_argument("--seg_ckpt", default=None, help="Lightning TrainSeg .ckpt for the segmentation guidance")
    parser.add_argument("--seg_synthetic", action="store_true")
    parser.add_argument("--n_levels", type=int, default=None)
    args_dict = vars(parser.parse_args())

    from inverseproblemwithdiffusionmodel_amd import engine
    from inverseproblemwithdiffusionmodel_amd.helpers import load_data as ld
    from inverseproblemwithdiffusionmodel_amd.helpers.load_model import load_scorenet_weights, reload_model
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import RandomUndersamplingFourier
    from inverseproblemwithdiffusionmodel_amd.ncsn.models import get_sigmas
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.ALD_optimizers import ALDInvSegProximalRealImag
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.proximal_op import get_proximal
    from inverseproblemwithdiffusionmodel_amd.synthetic import phantom_image

    device = torch.device("cuda")
    H = args_dict["image_size"]
    config = engine.acdc_config(device, H)
    label = None
    if args_dict["data_dir"]:
        ds = ld.load_data(args_dict["dataset"], "val", root_dir=args_dict["data_dir"],
                          **({"if_aug": False} if args_dict["dataset"] == "ACDC" else {}))
        data = ds[args_dict["ds_idx"]]
        if args_dict["dataset"] == "ACDC":
            img, label = data[ld.IMAGE_KEY][None], data[ld.LABEL_KEY][None]
        else:
            img = data[0][None]
        H = img.shape[-1]
        config.data.image_size = H
        img_complex = ld.add_phase(img, init_shape=(5, 5), seed=args_dict["seed"])
    else:
        img_complex = phantom_image(H, H, seed=args_dict["seed"])                  # magnitude phantom * exp(i smooth phase)
        label = (img_complex.abs() > 0.5).long() if args_dict["dataset"] == "ACDC" else None
    img_complex = img_complex.to(device).to(torch.complex64)

    scorenet = engine.build_scorenet(config, args_dict["seed"])
    if args_dict["ckpt"]:
        load_scorenet_weights(scorenet, args_dict["ckpt"])
    seg = None
    seg_start = args_dict["seg_start_time"]
    if args_dict["dataset"] != "ACDC" or label is None:
        seg_start = 1.                                                             # the CINE script passes label=None
    elif seg_start < 1. and not (args_dict["seg_ckpt"] or args_dict["seg_synthetic"]):
        print("no --seg_ckpt: segmentation-likelihood guidance needs trained UNet weights -> guidance off "
              "(pass --seg_synthetic to exercise the path with random weights)")
        seg_start = 1.
    if seg_start < 1.:
        seg = reload_model("Seg", "ACDC", ckpt_path=args_dict["seg_ckpt"], device=device)
    params = dict(n_steps_each=args_dict["num_steps_each"], step_lr=args_dict["step_lr"], final_only=True, denoise=True)
    sigmas = get_sigmas(config, mode="recons")
    x_mod_shape = (args_dict["num_samples"], config.data.channels, H, H)
    linear_tfm = RandomUndersamplingFourier(args_dict["R"], args_dict["center_lines_frac"], x_mod_shape[1:], args_dict["seed"],
                                            mask_params=None if args_dict["R"] in (8, 16, 20, 40) else
                                            dict(sw=0.3, sm=0.7, sa=0.045))         # generate_mask's own defaults
    proximal = get_proximal(args_dict["proximal_type"])(linear_tfm)
    measurement = linear_tfm(img_complex).repeat(args_dict["num_samples"], 1, 1, 1)
    sampler = ALDInvSegProximalRealImag(proximal, seg_start, args_dict["seg_step_type"], x_mod_shape, scorenet, sigmas, params,
                                        config, measurement, linear_tfm, seg=seg, device=device)
    save_dir = args_dict["save_dir"]
    os.makedirs(save_dir, exist_ok=True)
    direct_recons = linear_tfm.conj_op(measurement)[:1]
    original_error = torch.sum(torch.abs(linear_tfm(direct_recons) - measurement[:1]) ** 2).item()
    t0 = time.time()
    img_out = sampler(label=label, lamda=args_dict["lamda"], save_dir=save_dir, lr_scaled=args_dict["lr_scaled"],
                      seg_mode="full", seed=args_dict["seed"], n_levels=args_dict["n_levels"], verbose=True)[0]
    torch.cuda.synchronize()
    l2_error = torch.sum(torch.abs(linear_tfm(img_out.to(device)) - measurement) ** 2, dim=(1, 2, 3)).mean().item()
    print("-" * 100)
    print(args_dict)
    print(f"sampling time: {time.time() - t0:.1f} s")
    print(f"original error: {original_error}")
    print(f"reconstruction error: {l2_error}")
    torch.save(img_complex.cpu(), os.path.join(save_dir, "original.pt"))
    torch.save(measurement[:1].cpu(), os.path.join(save_dir, "measurement.pt"))
    torch.save(img_out.cpu(), os.path.join(save_dir, "reconstructions.pt"))
    torch.save(direct_recons.cpu(), os.path.join(save_dir, "ZF.pt"))
    torch.save(linear_tfm.mask, os.path.join(save_dir, "mask.pt"))
    with open(os.path.join(save_dir, "args_dict.pkl"), "wb") as wf:
        pickle.dump(args_dict, wf)


if __name__ == '__main__':
    main("ACDC")
