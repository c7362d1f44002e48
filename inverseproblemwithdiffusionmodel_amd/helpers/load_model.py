"""Model registry and checkpoint ingestion (mirror of the reference's ``helpers/load_model.py:23-31,164-186`` and
``helpers/utils.py:161-170``).  The reference trains with Lightning and stores the EMA weights in the callback
state of the ``.ckpt``: ``ckpt["callbacks"][<EMA key>]["ema_state_dict"]`` with a ``model.`` prefix on every key;
``load_scorenet_weights`` extracts them (or takes a plain state dict) and loads them into the gfx950 modules, whose
state-dict keys are the reference's."""
import torch

from ..ncsn.models.ncsnv2 import NCSNv2, NCSNv2Deeper, NCSNv2Deepest
from ..ncsn.models.ncsn3d import NCSN3DShallow

TASK_NAME_TO_MODEL_CTOR = {
    "Diffusion": NCSNv2Deepest,
    "Diffusion3D": NCSN3DShallow,
    "DiffusionShallow": NCSNv2,
    "DiffusionDeeper": NCSNv2Deeper,
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
    assert task_name in TASK_NAME_TO_MODEL_CTOR, f"{task_name}: only the score networks are built (no Seg / Clf)"
    ds_cfg = ds_name + "_1D" if task_name == "Diffusion3D" and not ds_name.endswith("_1D") else ds_name
    config = load_config(ds_cfg, mode, device)
    model = TASK_NAME_TO_MODEL_CTOR[task_name](config)
    if ckpt_path is not None:
        load_scorenet_weights(model, ckpt_path)
    else:
        sd = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
        model.load_state_dict(sd, strict=False)
    return model.to(config.device).eval()
