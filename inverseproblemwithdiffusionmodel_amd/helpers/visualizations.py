"""Posterior-sample reporting (mirror of the reference's ``helpers/visualizations.py:58-192`` create_sample_grid_plot /
add_correlation_table and :195-300 metric_vs_hyperparam).  Reads the artefacts a driver script wrote (original.pt,
reconstructions.pt, args_dict.pkl) and produces

  * ``sample_grid_report(root_dir)``      the NUMBERS of the reference's figure: SNR, NRMSE / SSIM of the magnitudes per
                                          sample, magnitude / phase standard-deviation maps, and per sample the 2 x 2 Spearman
                                          rank-correlation table (|error| and |magnitude error| against the two std maps);
  * ``create_sample_grid_plot(root_dir)`` the figure itself (5 rows x (3 + B) columns, same panel order and titles);
  * ``metric_vs_hyperparam(root_dirs)``   metric-against-hyper-parameter curves over a set of runs.

Metrics come from the on-device kernels (helpers/metrics.py::compute_metrics_device) when a GPU is present, else from
their host definitions; the rank correlation is scipy.stats.spearmanr as in the reference (reporting, not the hot path).
matplotlib is imported lazily with the Agg backend (the figure functions raise if it is missing; the numbers do not need it)."""
import os
import pickle
from typing import Dict, List

import numpy as np
import torch
from scipy.stats import spearmanr

from .metrics import REGISTERED_METRICS, compute_metrics, compute_metrics_device, compute_snr

FIGSIZE_UNIT = 3.6
ROW_TEXTS, COL_TEXTS = ["abs error", "abs mag error"], ["mag std", "phase std"]


def compute_angle(img, if_normalize=False):
    img = img.detach().cpu().numpy() if isinstance(img, torch.Tensor) else np.asarray(img)
    angle = np.angle(img)
    if if_normalize:
        angle = angle - angle.min()
        angle = angle / angle.max()
    return angle


def load_pickle(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def correlation_table(mag_std, phase_std, abs_error, abs_mag_error) -> np.ndarray:
    """2 x 2 Spearman rank correlations: rows (abs error, abs mag error) x columns (mag std, phase std) (:33-55)"""
    corr = np.zeros((2, 2))
    for i, x_iter in enumerate([abs_error, abs_mag_error]):
        for j, y_iter in enumerate([mag_std, phase_std]):
            corr[i, j] = spearmanr(np.asarray(x_iter).ravel(), np.asarray(y_iter).ravel()).correlation
    return corr


def _load_run(root_dir, orig_filename, recons_filename, args_filename):
    img_orig = torch.load(os.path.join(root_dir, orig_filename), weights_only=False)      # (1, C, H, W)
    recons = torch.load(os.path.join(root_dir, recons_filename), weights_only=False)      # (B, C, H, W)
    args_path = os.path.join(root_dir, args_filename)
    args_dict = load_pickle(args_path) if os.path.exists(args_path) else {}
    return img_orig, recons, args_dict


def _metrics(metric_names, recons, img_orig):
    """NRMSE / SSIM / ... of the magnitudes, per sample: HIP kernels on a GPU box, host definitions otherwise"""
    if torch.cuda.is_available() and recons.dim() == 4 and recons.shape[1] == 1:
        return compute_metrics_device(metric_names, recons.cuda(), img_orig.cuda())
    return compute_metrics(metric_names, np.abs(recons.numpy()), np.abs(img_orig.numpy()))


def sample_grid_report(root_dir: str, orig_filename="original.pt", recons_filename="reconstructions.pt",
                       args_filename="args_dict.pkl", metrics=("NRMSE", "SSIM")) -> Dict:
    img_orig, recons, args_dict = _load_run(root_dir, orig_filename, recons_filename, args_filename)
    rec_np, orig_np = recons.numpy(), img_orig.numpy()
    metric_vals = {"SNR": compute_snr(rec_np)}
    metric_vals.update(_metrics(list(metrics), recons, img_orig))
    mag_std = np.abs(rec_np).std(axis=0)
    phase_std = compute_angle(rec_np).std(axis=0)
    corr = []
    for idx in range(rec_np.shape[0]):
        abs_err = np.abs(rec_np[idx, 0] - orig_np[0, 0])
        abs_mag_err = np.abs(np.abs(rec_np[idx, 0]) - np.abs(orig_np[0, 0]))
        corr.append(correlation_table(mag_std, phase_std, abs_err, abs_mag_err))
    return dict(metrics={k: np.asarray(v) for k, v in metric_vals.items()},
                metrics_mean={k: float(np.mean(v)) for k, v in metric_vals.items()},
                mag_mean=np.abs(rec_np[:, 0]).mean(axis=0), phase_mean=compute_angle(rec_np[:, 0]).mean(axis=0),
                mag_std=mag_std, phase_std=phase_std, spearman=np.stack(corr), row_labels=ROW_TEXTS, col_labels=COL_TEXTS,
                args_dict=args_dict)


def _plt():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def add_text(axis, text_dict: dict):
    cols = list(text_dict)
    vals = np.round(np.array([[float(np.ravel(text_dict[c])[0])] for c in cols]), decimals=3)
    axis.table(cellText=vals, rowLabels=cols, loc="center", cellLoc="center")
    axis.set_axis_off()


def add_correlation_table(axis, mag_std, phase_std, abs_error, abs_mag_error):
    corr = correlation_table(mag_std, phase_std, abs_error, abs_mag_error)
    axis.table(cellText=corr.round(decimals=4), rowLabels=ROW_TEXTS, colLabels=COL_TEXTS, loc="center", cellLoc="center")
    axis.set_axis_off()


def create_sample_grid_plot(root_dir: str, orig_filename="original.pt", recons_filename="reconstructions.pt",
                            args_filename="args_dict.pkl", *args, **kwargs):
    """kwargs: if_save, save_dir, metrics.  -> the figure (saved as samples.png when if_save)"""
    plt = _plt()
    rep = sample_grid_report(root_dir, orig_filename, recons_filename, args_filename, kwargs.get("metrics", ["NRMSE", "SSIM"]))
    img_orig, recons, args_dict = _load_run(root_dir, orig_filename, recons_filename, args_filename)
    rec, orig = recons.numpy(), img_orig.numpy()
    B = rec.shape[0]
    num_cols, num_rows = 3 + B, 5
    fig, axes = plt.subplots(num_rows, num_cols, figsize=(FIGSIZE_UNIT * num_cols, FIGSIZE_UNIT * num_rows))

    def show(ax, img, title):
        h = ax.imshow(img, cmap="gray")
        plt.colorbar(h, ax=ax)
        ax.set_title(title)

    for j in range(num_cols):
        keep = {0, 1}
        if j == 0:
            show(axes[0, j], np.abs(orig[0, 0]), "mag gt")
            show(axes[1, j], compute_angle(orig[0, 0]), "phase gt")
        elif j == num_cols - 2:
            show(axes[0, j], rep["mag_mean"], "mean mag")
            show(axes[1, j], rep["phase_mean"], "mean phase")
            add_text(axes[2, j], {k: [v] for k, v in rep["metrics_mean"].items()})
            keep = {0, 1, 2}
        elif j == num_cols - 1:
            show(axes[0, j], rep["mag_std"][0], "std mag")
            show(axes[1, j], rep["phase_std"][0], "std phase")
        else:
            idx = j - 1
            panels = [np.abs(rec[idx, 0]), compute_angle(rec[idx, 0]), np.abs(rec[idx, 0] - orig[0, 0]),
                      np.abs(np.abs(rec[idx, 0]) - np.abs(orig[0, 0]))]
            for i, (img, title) in enumerate(zip(panels, ["mag", "phase", "abs diff", "abs mag diff"])):
                show(axes[i, j], img, title)
            add_correlation_table(axes[-1, j], rep["mag_std"], rep["phase_std"], panels[-2], panels[-1])
            keep = {0, 1, 2, 3, 4}
        for i in range(num_rows):
            if i not in keep:
                axes[i, j].set_axis_off()
    if "lr_scaled" in args_dict and "step_lr" in args_dict:
        fig.suptitle(r"$\lambda = $" + f"{args_dict['lr_scaled']: .2E}, " + r"$\alpha$ = " + f"{args_dict['step_lr']: .2E}, ")
    fig.tight_layout()
    if kwargs.get("if_save", False):
        fig.savefig(os.path.join(kwargs.get("save_dir", root_dir), "samples.png"))
    plt.close(fig)
    return fig


def metric_vs_hyperparam(root_dirs: List[str], metrics: List[str], params: List[str], defaults: dict, if_logscale_x=(),
                         orig_filename="original.pt", recons_filename="reconstructions.pt", args_filename="args_dict.pkl",
                         *args, **kwargs):
    """one curve per (hyper-parameter, metric): runs whose OTHER hyper-parameters sit at their defaults, ordered by the
    varied one; the metric of a run is that of its first sample (as the reference, :278).  -> (curves dict, figure or None)
    kwargs: if_save, save_dir, no_plot"""
    for p in params:
        assert p in defaults, f"{p} is not valid."
    for m in metrics:
        assert m in REGISTERED_METRICS, f"Metric {m} is not supported."
    vals = {}
    for root_dir in root_dirs:
        img_orig, recons, args_dict = _load_run(root_dir, orig_filename, recons_filename, args_filename)
        vals[tuple(args_dict[p] for p in params)] = _metrics(list(metrics), recons, img_orig)
    curves = {}
    for i, p in enumerate(params):
        for m in metrics:
            pts = sorted((key[i], float(v[m][0])) for key, v in vals.items()
                         if all(key[k] == defaults[params[k]] for k in range(len(params)) if k != i))
            curves[(p, m)] = (np.array([a for a, _ in pts]), np.array([b for _, b in pts]))
    if kwargs.get("no_plot", False):
        return curves, None
    plt = _plt()
    fig, axes = plt.subplots(len(params), len(metrics), figsize=(FIGSIZE_UNIT * len(metrics), FIGSIZE_UNIT * len(params)),
                             squeeze=False)
    names = {"step_lr": r"$\alpha$", "lr_scaled": r"$\lambda$"}
    for i, p in enumerate(params):
        for j, m in enumerate(metrics):
            ax = axes[i, j]
            ax.set_xlabel(names.get(p, p))
            ax.set_ylabel(m)
            ax.plot(*curves[(p, m)])
            if p in if_logscale_x:
                ax.set_xscale("log")
    fig.tight_layout()
    if kwargs.get("if_save", False):
        assert kwargs.get("save_dir") is not None
        fig.savefig(os.path.join(kwargs["save_dir"], "metrics.png"))
    plt.close(fig)
    return curves, fig
