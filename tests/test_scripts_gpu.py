"""Script-level tests on the GPU box: every driver script of scripts/ runs as its own process with the reference's flag
names (scripts/acdc_SENSE_real_img.py:163-175 of the reference lists the artefacts: original.pt, measurement.pt,
reconstructions.pt, ZF.pt, args_dict.pkl) on a short slice of the schedule; and the sharded runs of BASELINE configs 3, 4
and 5 -- two ranks (gloo, both on cuda:0: RCCL allows one rank per card and the box has one) against one rank, bit for bit.
Children are started as fresh processes (never exec'd from a GPU-initialised one)."""
import os
import pickle
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_script(name, args, world=1, timeout=900):
    """scripts/<name> with `args` on `world` ranks; returns rank 0's stdout"""
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if world > 1:
            env.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), IPDM_DIST_BACKEND="gloo", IPDM_BENCH_DEVICE="0",
                       IPDM_DEVICE_TURNS="1")       # ranks sharing the one card take turns (sharding.py)
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "scripts", name)] + [str(a) for a in args], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=REPO))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"{name} rank {rank}: rc {p.returncode}\n{so[-1500:]}\n{se[-3000:]}"
    return outs[0][0]


def _load(d, name):
    return torch.load(os.path.join(d, name), weights_only=False)


def _args_of(d):
    with open(os.path.join(d, "args_dict.pkl"), "rb") as f:
        return pickle.load(f)


# ---- artefacts of every driver script ---------------------------------------------------------------------------------
def test_acdc_SENSE_real_img_script(tmp_path):
    d = str(tmp_path)
    out = run_script("acdc_SENSE_real_img.py", ["--R", 40, "--num_samples", 2, "--num_sens", 4, "--lr_scaled", 1.0, "--seed", 0,
                                               "--seg_start_time", 1.0, "--proximal_type", "L2Penalty", "--n_levels", 2,
                                               "--save_dir", d])
    assert "reconstruction time" in out
    rec, orig, meas, zf = (_load(d, n) for n in ("reconstructions.pt", "original.pt", "measurement.pt", "ZF.pt"))
    assert rec.shape == (2, 1, 128, 128) and rec.dtype == torch.complex64 and torch.isfinite(torch.view_as_real(rec)).all()
    assert orig.shape == (1, 1, 128, 128) and meas.shape == (4, 1, 1, 128, 128) and meas.dtype == torch.complex64
    assert zf.shape == (1, 1, 128, 128) and _load(d, "mask.pt").dtype == torch.bool
    post = _load(d, "posterior.pt")
    assert set(post) == {"mag_mean", "phase_mean", "mag_std", "phase_std", "mean"} and post["mag_mean"].shape == (1, 128, 128)
    a = _args_of(d)
    assert a["R"] == 40 and a["num_samples"] == 2 and a["proximal_type"] == "L2Penalty"


@pytest.mark.parametrize("script,extra", [("acdc_inv_seg_sampling_keep_center_prox_real_imag.py", ["--seg_start_time", 1.0]),
                                          ("cine_inv_sampling_keep_center_prox_real_imag.py", [])])
def test_single_coil_scripts(tmp_path, script, extra):
    d = str(tmp_path)
    run_script(script, ["--R", 6, "--num_samples", 2, "--n_levels", 2, "--save_dir", d] + extra)
    rec = _load(d, "reconstructions.pt")
    assert rec.dtype == torch.complex64 and rec.shape[0] == 2 and torch.isfinite(torch.view_as_real(rec)).all()
    for n in ("original.pt", "measurement.pt", "ZF.pt", "args_dict.pkl"):
        assert os.path.exists(os.path.join(d, n)), n


def test_cine_2d_time_script(tmp_path):
    d = str(tmp_path)
    out = run_script("cine_SENSE_real_img_2d_time.py", ["--R", 8, "--num_samples", 1, "--mode_T", "diffusion1d", "--lamda_T", 10.0,
                                                       "--image_size", 64, "--start_level", 996, "--n_levels", 2, "--save_dir", d])
    assert "reconstruction time" in out
    rec, orig = _load(d, "reconstructions.pt"), _load(d, "original.pt")
    assert rec.shape == (1, 24, 1, 64, 64) and rec.dtype == torch.complex64 and torch.isfinite(torch.view_as_real(rec)).all()
    assert orig.shape == (24, 1, 64, 64) and _load(d, "mask.pt").shape[0] == 24


def test_cine_2d_time_MAP_script(tmp_path):
    d = str(tmp_path)
    out = run_script("cine_SENSE_real_img_2d_time_MAP.py", ["--ds_name", "CINE127", "--R", 6, "--num_iters", 2, "--lr", 0.001,
                                                           "--mode_T", "diffusion1d", "--save_dir", d])
    assert "reconstruction error" in out and "reconstruction time" in out
    rec, zf, meas = _load(d, "reconstructions.pt"), _load(d, "ZF.pt"), _load(d, "measurement.pt")
    assert rec.shape == zf.shape and rec.dim() == 5 and rec.dtype == torch.complex64 and meas.dim() == 6
    assert torch.isfinite(torch.view_as_real(rec)).all() and not torch.equal(rec, zf)
    assert _args_of(d)["num_iters"] == 2


def test_acdc_SENSE_MAP_and_TV_scripts(tmp_path):
    d1, d2 = str(tmp_path / "map"), str(tmp_path / "tv")
    run_script("acdc_SENSE_MAP.py", ["--R", 8, "--n_iters", 3, "--lamda", 0.01, "--save_dir", d1])
    out = run_script("acdc_SENSE_TV.py", ["--R", 5, "--num_epochs", 20, "--lr", 0.01, "--reg_weight", 0.01, "--save_dir", d2])
    assert "original error" in out and "reconstruction error" in out
    for d in (d1, d2):
        rec, zf = _load(d, "reconstructions.pt"), _load(d, "ZF.pt")
        assert rec.shape == zf.shape == (1, 1, 128, 128) and rec.dtype == torch.complex64
        assert torch.isfinite(torch.view_as_real(rec)).all() and not torch.equal(rec, zf)
        assert os.path.exists(os.path.join(d, "args_dict.pkl"))


def test_unconditioned_sampling_script(tmp_path):
    d = str(tmp_path)
    out = run_script("unconditioned_sampling.py", ["--ds_name", "MNIST", "--num_samples", 2, "--num_steps_each", 1, "--save_dir", d])
    assert "score evaluations" in out
    x = _load(d, "unconditioned_samples.pt")
    assert x.shape[0] == 2 and x.dim() == 4 and torch.isfinite(x).all()


# ---- sharded runs: two ranks == one rank, bit for bit ---------------------------------------------------------------------
def test_config3_sharded_equals_single_rank(tmp_path):
    """scripts/acdc_SENSE_real_img.py (BASELINE configs 2 / 3): 4 posterior samples on one rank vs 2 + 2 on two ranks"""
    d1, d2 = str(tmp_path / "w1"), str(tmp_path / "w2")
    args = ["--R", 40, "--num_samples", 4, "--n_levels", 3, "--seed", 3, "--seg_start_time", 1.0]
    run_script("acdc_SENSE_real_img.py", args + ["--save_dir", d1], world=1)
    run_script("acdc_SENSE_real_img.py", args + ["--save_dir", d2], world=2)
    a, b = _load(d1, "reconstructions.pt"), _load(d2, "reconstructions.pt")
    assert a.shape == (4, 1, 128, 128) and torch.equal(torch.view_as_real(a), torch.view_as_real(b))
    assert not torch.equal(a[0], a[1])                                        # the samples differ: the Philox key is global
    pa, pb = _load(d1, "posterior.pt"), _load(d2, "posterior.pt")
    for k in pa:                                                               # float64 sums in a different order: rounding
        assert torch.allclose(pa[k], pb[k], rtol=1e-6, atol=1e-6), k


def test_config4_sharded_equals_single_rank(tmp_path):
    """scripts/cine_SENSE_real_img_2d_time.py (BASELINE config 4) with the temporal prior active and the per-step random
    shift (one host draw per step for the whole batch, ALD_optimizers.py:472): 2 samples on one rank vs 1 + 1 on two"""
    d1, d2 = str(tmp_path / "w1"), str(tmp_path / "w2")
    args = ["--R", 8, "--num_samples", 2, "--mode_T", "diffusion1d", "--lamda_T", 10.0, "--if_random_shift", "--image_size", 64,
            "--start_level", 997, "--n_levels", 2, "--seed", 5]
    run_script("cine_SENSE_real_img_2d_time.py", args + ["--save_dir", d1], world=1)
    run_script("cine_SENSE_real_img_2d_time.py", args + ["--save_dir", d2], world=2)
    a, b = _load(d1, "reconstructions.pt"), _load(d2, "reconstructions.pt")
    assert a.shape == (2, 24, 1, 64, 64) and torch.equal(torch.view_as_real(a), torch.view_as_real(b))
    assert not torch.equal(a[0], a[1])


def test_config5_sharded_equals_single_rank(tmp_path):
    """scripts/ncsnpp_pc_sampling.py (BASELINE config 5): the predictor-corrector sampler whose LangevinCorrector couples
    the batch through two means (sde/sampling.py:281-283): 4 samples on one rank vs 2 + 2 with the means all-reduced"""
    d1, d2 = str(tmp_path / "w1"), str(tmp_path / "w2")
    args = ["--tiny", "--num_samples", 4, "--n_iters", 6, "--seed", 2]
    out = run_script("ncsnpp_pc_sampling.py", args + ["--save_dir", d1], world=1)
    assert "sampling time" in out
    run_script("ncsnpp_pc_sampling.py", args + ["--save_dir", d2], world=2)
    a, b = _load(d1, "samples.pt"), _load(d2, "samples.pt")
    assert a.shape == (4, 3, 32, 32) and torch.isfinite(a).all() and torch.equal(a, b)
    assert not torch.equal(a[0], a[1])
