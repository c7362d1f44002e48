"""Model registry and checkpoint ingestion (mirror of the reference's ``helpers/load_model.py:23-31,164-186`` and
``helpers/utils.py:161-170``).  The reference trains with Lightning and stores the EMA weights in the callback
state of the ``.ckpt``: ``ckpt["callbacks"][<EMA key>]["ema_state_dict"]`` with a ``model.`` prefix on every key;
``load_scorenet_weights`` extracts them (or takes a plain state dict) and loads them into the gfx950 modules, whose
state-dict keys are the reference's."""
import torch

from ..ncsn.models.ncsnv2 import NCSNv2, NCSNv2Deeper, NCSNv2Deepest
from ..ncsn.models.ncsn3d import NCSN3DShallow
from ..ncsn.models.seg_unet import UNet

# ncsn/configs/general_config.yml:1-6 of the reference ("Seg": the MONAI UNet arguments)
GENERAL_CONFIG = {"Seg": dict(spatial_dims=2, in_channels=1, out_channels=2, channels=[64, 128, 256, 512, 1024],
                              strides=[2, 2, 2, 2])}

TASK_NAME_TO_MODEL_CTOR = {
    "Diffusion": NCSNv2Deepest,
    "Diffusion3D": NCSN3DShallow,
    "DiffusionShallow": NCSNv2,
    "DiffusionDeeper": NCSNv2Deeper,
    "Seg": UNet,
}


def collate_state_dict(state_dict, prefix="model."):
    """strip the LightningModule attribute prefix from every key"""
    n = len(prefix)
    return {(k[n:] if k.startswith(prefix) else k): v for k, v in state_dict.items()}


def extract_ema_state_dict(ckpt):
    """Lightning checkpoint dict -> EMA weights (falls back to 'state_dict', then to the object itself)"""
    if isinstance(ckpt, dict) and "callbacks" in ckpt:
        for state in ckpt["callbacks"].values():
            if isinstance(state, dict) and "ema_state_dict" in state:
                return collate_state_dict(state["ema_state_dict"])
    if isinstance(ckpt, dict) and "state_dict" in ckpt:
        return collate_state_dict(ckpt["state_dict"])
    return collate_state_dict(ckpt)


def load_scorenet_weights(model, path_or_state, strict=True):
    ckpt = torch.load(path_or_state, map_location="cpu", weights_only=False) if isinstance(path_or_state, str) else path_or_state
    sd = extract_ema_state_dict(ckpt)
    return model.load_state_dict(sd, strict=strict)


def reload_model(task_name, ds_name, mode="real-valued", ckpt_path=None, device=None):
    """build the score network for (task, dataset); with ckpt_path its EMA weights are loaded, otherwise the seeded
    synthetic weights of synthetic.py (no checkpoints ship with the reference)"""
    from .load_data import load_config
    from ..synthetic import synth_state_dict
    assert task_name in TASK_NAME_TO_MODEL_CTOR, f"{task_name}: the score networks and the segmentation UNet are built (no Clf)"
    if task_name == "Seg":
        # helpers/load_model.py:140-141: UNet(**general_config["Seg"]); Lightning TrainSeg checkpoints carry the weights
        # under 'model.' + MONAI's names, which extract_ema_state_dict / collate_state_dict strip
        if device is None:
            device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        model = UNet(**GENERAL_CONFIG["Seg"])
        if ckpt_path is not None:
            load_scorenet_weights(model, ckpt_path)
        else:
            sd = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
            for k in sd:
                if k.endswith("adn.A.weight"):
                    sd[k] = torch.full_like(sd[k], 0.25)         # PReLU slope at its init value
            model.load_state_dict(sd, strict=False)
        return model.to(device).eval()
    ds_cfg = ds_name + "_1D" if task_name == "Diffusion3D" and not ds_name.endswith("_1D") else ds_name
    config = load_config(ds_cfg, mode, device)
    model = TASK_NAME_TO_MODEL_CTOR[task_name](config)
    if ckpt_path is not None:
        load_scorenet_weights(model, ckpt_path)
    else:
        sd = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
        model.load_state_dict(sd, strict=False)
    return model.to(config.device).eval()
