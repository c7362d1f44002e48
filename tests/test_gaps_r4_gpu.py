"""Round-4 closures of the test gaps VERDICT r3 named (one test each):
  * the reference's hook loop (ALD_optimizers.py:204-252 as rebuilt in `_call_with_hooks`): a subclass that overrides
    `adjust_grad` (identity) and `post_processing` (calling the base) must reproduce the reference's own trajectories g08 / g20;
  * a sample's bits do not depend on where the 32-bit-offset launch split (`wino_bx3_max_batch`) cuts the batch;
  * BASELINE config 2 as a SAMPLER run: R = 20, 8 samples, 128 x 128, 4 coils, two noise levels, against the CPU oracle under
    injected noise."""
import numpy as np
import pytest
import torch

from oracle import kspace, scorenet as oracle_net, ald as oracle_ald, metrics
from test_scorenet_gpu import tiny_config, pkg, tiny_net, full_net, _sense_sampler, _Tape, _SeededNoise, _LabelOffset  # noqa: F401

pytestmark = pytest.mark.gpu


def _hooked(base):
    class Hooked(base):
        """overrides both hooks the way a user extension would: the fused iteration must step aside for the generic loop"""
        n_adjust = n_post = 0

        def adjust_grad(self, grad, m_mod, **kwargs):
            type(self).n_adjust += 1
            assert grad.shape == m_mod.shape and "sigma" in kwargs and "seg_lamda" in kwargs
            return grad

        def post_processing(self, x_mod_real, x_mod_imag, **kwargs):
            type(self).n_post += 1
            return super().post_processing(x_mod_real, x_mod_imag, **kwargs)
    return Hooked


@pytest.mark.parametrize("tag", ["dc_visible", "script_default"])
def test_hook_loop_reproduces_tiny_reference_trajectory(pkg, tiny_net, golden, tag):
    g, fused = _sense_sampler(pkg, tiny_net, golden)
    Hooked = _hooked(pkg.ald.ALDInvSegProximalRealImag)
    op = fused.linear_tfm
    sampler = Hooked(pkg.prox.get_proximal("L2Penalty")(op), 1.0, "linear", (2, 1, 32, 32), tiny_net, fused.sigmas,
                     dict(n_steps_each=3, step_lr=9e-7, denoise=True, final_only=True), tiny_config(), fused.measurement, op,
                     seg=None, device=torch.device("cuda"))
    assert sampler._hooks_overridden() and not fused._hooks_overridden()
    tape = _Tape(g["noise"])
    x = sampler(label=None, lamda=0.1, save_dir=None, lr_scaled=float(g[f"{tag}_lr_scaled"]), seg_mode="full",
                noise_fn=tape)[0].numpy()
    assert tape.i == 60 and Hooked.n_post == 30 and Hooked.n_adjust == 60          # 10 levels x 3 steps, two planes
    ref = g[f"{tag}_x"]
    for b in range(x.shape[0]):
        assert metrics.nrmse(np.abs(x[b]), np.abs(ref[b])) < 1e-3
        assert abs(metrics.ssim(np.abs(x[b, 0]), np.abs(ref[b, 0])) - 1.0) < 1e-3
    np.testing.assert_allclose(x, ref, atol=1e-3)
    # ... and the fused iteration it replaces gives the same reconstruction
    y = fused(label=None, lamda=0.1, save_dir=None, lr_scaled=float(g[f"{tag}_lr_scaled"]), seg_mode="full",
              noise_fn=_Tape(g["noise"]))[0].numpy()
    np.testing.assert_allclose(x, y, atol=2e-5 * np.abs(y).max())


def test_hook_loop_reproduces_headline_reference_trajectory(pkg, golden, full_net):
    """g20 'tail_dc' (94.1 M-parameter network, 128 x 128, R = 40, lr_scaled 2e6) through the hook loop"""
    cfg, net = full_net
    g = golden("g20_fullsize_ald")
    lv0, lr_scaled, seed, n_calls, n_sum = g["tail_dc_meta"]
    lv0 = int(lv0)
    B, H, W = 2, 128, 128
    op = pkg.uf.SENSE("exp", 4, 40, 0.04, (1, H, W), seed=0)
    sig = torch.from_numpy(kspace.get_sigmas(348, 0.01, 2311))[lv0:lv0 + 12].clone().cuda()
    meas = torch.from_numpy(g["measurement_1"]).repeat(1, B, 1, 1, 1).cuda()
    Hooked = _hooked(pkg.ald.ALDInvSegProximalRealImag)
    sampler = Hooked(pkg.prox.get_proximal("L2Penalty")(op), 1.0, "linear", (B, 1, H, W), _LabelOffset(net, lv0), sig,
                     dict(n_steps_each=3, step_lr=9e-7, denoise=True, final_only=True), cfg, meas, op, seg=None,
                     device=torch.device("cuda"))
    noise = _SeededNoise(seed)
    x = sampler(label=None, lamda=0.1, save_dir=None, lr_scaled=float(lr_scaled), seg_mode="full", noise_fn=noise)[0].numpy()
    assert noise.calls == int(n_calls) == 72 and Hooked.n_post == 36 and Hooked.n_adjust == 72
    ref = g["tail_dc_x"]
    scale = np.abs(ref).max()
    for b in range(B):
        assert metrics.nrmse(np.abs(x[b]), np.abs(ref[b])) < 1e-3
        assert abs(metrics.ssim(np.abs(x[b, 0]), np.abs(ref[b, 0])) - 1.0) < 1e-3
    np.testing.assert_allclose(x, ref, atol=1e-3 * scale)
    x0 = op.conj_op(meas).cpu().numpy()
    assert np.linalg.norm((x - x0) - (ref - x0)) <= 2e-3 * np.linalg.norm(ref - x0)


def test_launch_split_keeps_every_samples_bits(monkeypatch):
    """batches beyond the Winograd kernels' 32-bit buffer offsets run as several launches (ops.wino_bx3_max_batch; the 210-image
    batch of config 3 on one GPU).  With the reach patched down so that a batch of 5 splits as 2 + 2 + 1, every form -- plain,
    residual + activated copy, pooled, statistics partials, split-K, dilated -- returns the bits of the unsplit launch, and the
    maxima vectors of the split launches are the unsplit ones"""
    from inverseproblemwithdiffusionmodel_amd import ops
    gen = torch.Generator().manual_seed(77)
    cases = [(5, 64, 64, 32, 64, 1, False), (5, 64, 128, 32, 32, 1, True), (5, 256, 256, 16, 16, 1, False),
             (5, 64, 64, 16, 16, 2, False), (5, 32, 64, 40, 36, 1, False)]
    for (B, Cin, Cout, H, W, dil, pool) in cases:
        x = (torch.randn(B, Cin, H, W, generator=gen) * torch.tensor([1.0, 30.0, 1e-2, 5.0, 0.3]).view(B, 1, 1, 1)).cuda()
        w = (torch.randn(Cout, Cin, 3, 3, generator=gen) / (9 * Cin) ** 0.5).cuda()
        b = torch.randn(Cout, generator=gen).cuda()
        oh, ow = (H // 2, W // 2) if pool else (H, W)
        res = torch.randn(B, Cout, oh, ow, generator=gen).cuda()
        U = ops.conv_wino_split_weight(w)
        kw = dict(dilation=dil, pool2=pool, act_out=ops.ACT_ELU, want_stats=not pool and dil == 1, want_amax=True,
                  in_amax=ops.absmax_per_image(x) if U.fmt == "hx2" else None)
        whole = ops.conv2d_wino_bx3(x, U, b, res, **kw)
        part_whole = getattr(whole[0], "_ipdm_partials", (None,))[0]
        with monkeypatch.context() as m:
            m.setattr(ops, "wino_bx3_max_batch", lambda *a, **k: 2)
            split = ops.conv2d_wino_bx3(x, U, b, res, **kw)
            part_split = getattr(split[0], "_ipdm_partials", (None,))[0]
        for t0, t1 in zip(whole, split):
            assert torch.equal(t0, t1), (Cin, H, dil, pool)
            assert torch.equal(ops.amax_value(ops.amax_of(t0)), ops.amax_value(ops.amax_of(t1)))
        assert (part_whole is None) == (part_split is None)
        if part_whole is not None:
            assert torch.equal(part_whole, part_split)
        # ... and a sample computed alone carries the same bits as inside the batch
        alone = ops.conv2d_wino_bx3(x[3:4].contiguous(), U, b, res[3:4].contiguous(),
                                    **dict(kw, in_amax=None if kw["in_amax"] is None else kw["in_amax"][3:4].contiguous()))
        assert torch.equal(alone[0][0], whole[0][3]) and torch.equal(alone[1][0], whole[1][3])


def test_config2_sampler_r20_b8_vs_cpu_oracle(pkg):
    """BASELINE config 2: ACDC 128 x 128 complex, SENSE R = 20, 4 coils, NCSNv2Deepest, 8 samples on one MI355X -- the sampler
    itself (fused iteration, hipGraph replay) for two noise levels x 3 steps + denoise against the CPU oracle sampler under the
    same injected noise; a narrow network (ngf 32) keeps the oracle at seconds.  NRMSE / SSIM 1e-3 per sample (north_star)."""
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict, phantom_image
    B, H, W = 8, 128, 128
    cfg = tiny_config(ngf=32, num_classes=2, sigma_begin=0.05, sigma_end=0.01)
    cfg.data.image_size = H
    net = pkg.ncsnv2.NCSNv2Deepest(cfg)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=2), strict=False)
    net = net.cuda().eval()
    op = pkg.uf.SENSE("exp", 4, 20, 0.04, (1, H, W), seed=0)
    mask = op.random_under_fourier.mask.numpy()
    assert 4 <= int(mask.sum()) <= 12                                        # R = 20: ~6 of 128 lines
    img = torch.cat([phantom_image(H, W, seed=s) for s in range(B)], dim=0).numpy().astype(np.complex64)      # (B, 1, H, W)
    meas_np = kspace.sense_forward(img, op.sens_maps.numpy(), mask)
    meas = torch.from_numpy(meas_np).cuda()
    sigmas = torch.from_numpy(kspace.get_sigmas(0.05, 0.01, 2)).cuda()
    params = dict(n_steps_each=3, step_lr=9e-7, denoise=True, final_only=True)
    sampler = pkg.ald.ALDInvSegProximalRealImag(pkg.prox.get_proximal("L2Penalty")(op), 1.0, "linear", (B, 1, H, W), net, sigmas,
                                                params, cfg, meas, op, seg=None, device=torch.device("cuda"))
    gen = torch.Generator().manual_seed(21)
    tape = [torch.randn(B, 1, H, W, generator=gen) for _ in range(12)]
    it = iter(tape)
    x = sampler(label=None, lamda=0.1, save_dir=None, lr_scaled=2e6, seg_mode="full", noise_fn=lambda like: next(it),
                use_graph=True)[0].numpy()
    sd_cpu = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    it2 = iter(tape)
    with torch.no_grad():
        ref = oracle_ald.ald_sense_real_imag(lambda xx, lab: oracle_net.ncsnv2_deepest(xx, lab, sd_cpu), sigmas.cpu().numpy(),
                                             meas_np, op.sens_maps.numpy(), mask, 9e-7, 3, 2e6, True, lambda like: next(it2))
    assert x.shape == ref.shape == (B, 1, H, W) and np.isfinite(x).all()
    x0 = op.conj_op(meas).cpu().numpy()
    assert np.linalg.norm(ref - x0) > 0.02 * np.linalg.norm(x0)                # data consistency + score moved the image
    for b in range(B):
        assert metrics.nrmse(np.abs(x[b]), np.abs(ref[b])) < 1e-3
        assert abs(metrics.ssim(np.abs(x[b, 0]), np.abs(ref[b, 0])) - 1.0) < 1e-3
    assert np.linalg.norm((x - x0) - (ref - x0)) <= 2e-3 * np.linalg.norm(ref - x0)
