"""``load_config`` (mirror of the reference's ``helpers/load_data.py:301-321``): YAML -> nested Namespace,
``config.device`` injected, ``mode == "complex"`` switches to 2 channels.

Real-data front end (mirror of ``helpers/load_data.py:125-164`` load_cine, ``:167-226``
load_tissue_data / vol2slice / LoadDataNumpyDict, ``:241-283`` load_ACDC, ``:372-397`` add_phase): the on-disk formats
either side of the fast path -- ACDC slice ``.npz`` files (keys image, multiClassMasks, PD, T1, T2) and CINE ``.mat``
stacks (key ``imgs``, (H, W, T, N)).  The reference drives MONAI dictionary transforms; MONAI is not vendored (version
unpinned), so ``ScaleIntensityd`` / ``CropForegroundd`` / ``Resized`` / ``Resize`` are restated from their
published behaviour (min-max scaling, bounding box of ``image > 0``, ``torch.nn.functional.interpolate`` with the
transform's mode) -- PARITY UNPINNED for the exact resampled pixel values; training-time augmentation (RandRotated ...)
is out of scope.  Host-side data preparation: plain numpy / torch-CPU, nothing here runs per sampling step."""
import glob
import os
import random
from typing import Union

import numpy as np
import torch
import torch.nn.functional as F

from .utils import load_yml_file

_CFG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ncsn", "configs")
REGISTERED_DATA_CONFIG_FILENAME = {
    "MNIST": os.path.join(_CFG_DIR, "mnist.yml"),
    "CINE127": os.path.join(_CFG_DIR, "cine127.yml"),
    "CINE127_1D": os.path.join(_CFG_DIR, "cine127_1d.yml"),
    "ACDC": os.path.join(_CFG_DIR, "acdc.yml"),
}


def load_config(ds_name, mode="real-valued", device=None, **kwargs):
    assert mode in ["real-valued", "mag", "complex", "real-imag", "real-imag-random"]
    assert ds_name in REGISTERED_DATA_CONFIG_FILENAME.keys()
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    cfg = load_yml_file(REGISTERED_DATA_CONFIG_FILENAME[ds_name])
    cfg.device = device
    if mode == "complex":
        cfg.data.channels = 2
    return cfg


# ---- real-data front end ----------------------------------------------------------------------------------------------
IMAGE_KEY, LABEL_KEY = "image", "label"          # monai.utils.CommonKeys.IMAGE / LABEL


def load_tissue_data(path_to_file):
    """one ACDC .npz: image intensity, multi-class segmentation, PD, T1, T2 ([1, N_slices, Nx, Ny] volumes or (1, H, W)
    slices written by vol2slice)"""
    data = np.load(path_to_file)
    return data["image"], data["multiClassMasks"], data["PD"], data["T1"], data["T2"]


def vol2slice(root_dir, save_dir):
    """save every slice of every volume .npz as its own .npz, all arrays (1, H, W)"""
    os.makedirs(save_dir, exist_ok=True)
    for filename in glob.glob(os.path.join(root_dir, "*.npz")):
        image, multi_class, PD, T1, T2 = load_tissue_data(filename)
        base = os.path.basename(filename)
        stem = base[:base.find(".npz")]
        for k in range(image.shape[1]):
            np.savez(os.path.join(save_dir, f"{stem}_{k}.npz"), image=image[:, k, ...], multiClassMasks=multi_class[:, k, ...],
                     PD=PD[:, k, ...], T1=T1[:, k, ...], T2=T2[:, k, ...])


class LoadDataNumpyDict:
    """filename -> {image (1, H, W) float32, label (1, H, W) int64 with the chosen classes merged to 1}"""

    def __init__(self, seg_labels: list):
        self.seg_labels = seg_labels

    def __call__(self, filename):
        image, label, _, _, _ = load_tissue_data(filename)
        label_out = np.zeros_like(label, dtype=np.int64)
        for seg_label in self.seg_labels:
            label_out[label == seg_label] = 1
        return {IMAGE_KEY: torch.tensor(image).float(), LABEL_KEY: torch.tensor(label_out).long()}


def _scale_intensity(img):
    """monai ScaleIntensity defaults: (x - min) / (max - min) -> [0, 1]"""
    lo, hi = img.min(), img.max()
    return (img - lo) / (hi - lo) if float(hi - lo) != 0.0 else img - lo


def _crop_foreground(data, source_key=IMAGE_KEY):
    """monai CropForegroundd defaults: bounding box of source > 0 over the spatial axes, margin 0"""
    fg = (data[source_key] > 0).any(dim=0)
    if not bool(fg.any()):
        return data
    rows, cols = torch.where(fg.any(dim=1))[0], torch.where(fg.any(dim=0))[0]
    r0, r1, c0, c1 = int(rows[0]), int(rows[-1]) + 1, int(cols[0]), int(cols[-1]) + 1
    return {k: v[:, r0:r1, c0:c1] for k, v in data.items()}


def _resize(img, size, mode):
    """monai Resize: torch.nn.functional.interpolate on (1, C, *spatial) with align_corners=None (True where asked)"""
    x = img[None].float()
    if mode in ("nearest", "area"):
        y = F.interpolate(x, size=size, mode=mode)
    else:
        y = F.interpolate(x, size=size, mode=mode, align_corners=False)
    return y[0].to(img.dtype) if mode == "nearest" else y[0]


class ACDCSliceDataset(torch.utils.data.Dataset):
    """what load_ACDC(if_aug=False) returns: ds[i] -> {image (1, 256, 256) float32 in [0, 1], label (1, 256, 256) int64}"""

    def __init__(self, filenames, seg_labels, spatial_size=(256, 256)):
        self.filenames, self.loader, self.spatial_size = list(filenames), LoadDataNumpyDict(seg_labels), spatial_size

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, idx):
        d = self.loader(self.filenames[idx])
        d[IMAGE_KEY] = _scale_intensity(d[IMAGE_KEY])
        d = _crop_foreground(d)
        return {IMAGE_KEY: _resize(d[IMAGE_KEY], self.spatial_size, "bilinear"),
                LABEL_KEY: _resize(d[LABEL_KEY], self.spatial_size, "nearest")}


def load_ACDC(root_dir, train_test_split=(0.8, 0.1), seg_labels=(3,), mode="train", seed=0, num_workers=4, if_aug=True):
    """file split exactly as the reference (glob order shuffled by random.seed(seed), 80/10/10); seg_label 3: left MYO.
    Training-time augmentation is not built: if_aug is accepted and ignored outside training."""
    assert mode in ["train", "val", "test"]
    all_filenames = glob.glob(os.path.join(root_dir, "*.npz"))
    random.seed(seed)
    random.shuffle(all_filenames)
    a = int(len(all_filenames) * train_test_split[0])
    b = int(len(all_filenames) * sum(train_test_split))
    filenames = {"train": all_filenames[:a], "val": all_filenames[a:b], "test": all_filenames[b:]}[mode]
    return ACDCSliceDataset(filenames, list(seg_labels))


def load_cine(root_dir, mode="train", img_key="imgs", flatten=True, flatten_type="spatial",
              resize_shape: Union[int, None] = None, resize_shape_T=None, win_size=2, **kwargs):
    """CINE .mat stack (H, W, T, N) -> per-volume min-max normalised TensorDataset: 'spatial' (N*T, 1, H0, W0) frames,
    'temporal' (N', win_size^2, T') patch time series (reshape_temporal_dim)"""
    import scipy.io as sio
    from .utils import reshape_temporal_dim
    assert mode in ["train", "val", "test"]
    assert flatten_type in ["spatial", "temporal"]
    if mode == "val":
        mode = "test"
    filename = glob.glob(os.path.join(root_dir, f"*{mode}*.mat"))[0]
    ds = sio.loadmat(filename)[img_key].transpose(3, 2, 0, 1)                      # (N, T, H, W)
    lo, hi = ds.min(axis=(1, 2, 3), keepdims=True), ds.max(axis=(1, 2, 3), keepdims=True)
    ds = (ds - lo) / (hi - lo)
    if flatten:
        N, T, H, W = ds.shape
        if flatten_type == "spatial":
            ds = torch.tensor(ds.reshape(-1, H, W))
            if resize_shape is not None and not (H == resize_shape and W == resize_shape):
                ds = _resize(ds, (resize_shape, resize_shape), "area")              # frames as channels, monai Resize default
            ds = ds[:, None, ...]
        else:
            size = (T if resize_shape_T is None else resize_shape_T, H if resize_shape is None else resize_shape,
                    W if resize_shape is None else resize_shape)
            ds = _resize(torch.tensor(ds), size, "area")                            # (N, T', H', W')
            ds = reshape_temporal_dim(ds, win_size, win_size)
    if isinstance(ds, np.ndarray):
        ds = torch.tensor(ds)
    return torch.utils.data.TensorDataset(ds)


def add_phase(imgs: torch.Tensor, init_shape: Union[tuple, int] = (5, 5), seed=None, mode="spatial"):
    """smooth random phase: N(0,1) patch of `init_shape` per image / channel, resized bicubic (spatial) or trilinear
    (2D+time) with align_corners=True, imgs * exp(i phase) -> complex64.  imgs (B, C, H, W) or (T, C, H, W)."""
    assert mode in ["spatial", "2D+time"]
    if seed is not None:
        torch.manual_seed(seed)
    if mode == "spatial":
        B, C, H, W = imgs.shape
        out = torch.empty(imgs.shape, dtype=torch.complex64, device=imgs.device)
        for i in range(B):
            patch = torch.randn(C, *init_shape, device=imgs.device)
            phase = F.interpolate(patch[None], size=(H, W), mode="bicubic", align_corners=True)[0]
            out[i] = imgs[i] * torch.exp(1j * phase)
        return out
    assert len(init_shape) == 3
    T, C, H, W = imgs.shape
    patch = torch.randn(C, *init_shape, device=imgs.device)
    phase = F.interpolate(patch[None], size=(T, H, W), mode="trilinear", align_corners=True)[0]     # (C, T, H, W)
    return imgs * torch.exp(1j * phase.permute(1, 0, 2, 3))


REGISTERED_DATA_ROOT_DIR = {}          # name -> directory; the reference hard-codes cluster paths (load_data.py:33-60)


def load_data(ds_name, mode="train", root_dir=None, **kwargs):
    """dispatcher (mirror of helpers/load_data.py:62-93 for the two on-disk datasets of the fast path)"""
    root = root_dir if root_dir is not None else REGISTERED_DATA_ROOT_DIR.get(ds_name)
    if root is None:
        raise FileNotFoundError(f"no data directory registered for {ds_name!r}: pass root_dir= or set "
                                "REGISTERED_DATA_ROOT_DIR (no data ships with the reference)")
    if ds_name == "ACDC":
        return load_ACDC(root, mode=mode, **kwargs)
    if ds_name == "CINE64":
        return load_cine(root, mode=mode, **kwargs)
    if ds_name == "CINE127":
        return load_cine(root, mode=mode, resize_shape=128, **kwargs)
    raise NotImplementedError(f"dataset {ds_name!r}: only the ACDC .npz and CINE .mat front ends are built")
