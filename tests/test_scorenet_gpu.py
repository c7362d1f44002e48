"""Score network + samplers on the GPU (HIP kernels through the C ABI) against the golden vectors the
reference produced and against the CPU oracle.  fp32 tolerances stated per test."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import state_dict_from_golden
from oracle import kspace, scorenet as oracle_net, ald as oracle_ald, metrics

pytestmark = pytest.mark.gpu


def tiny_config(ngf=4, num_classes=10, sigma_begin=1.0, sigma_end=0.01, device="cuda"):
    return Namespace(
        device=torch.device(device),
        data=Namespace(channels=1, image_size=32, logit_transform=False, rescaled=False,
                       uniform_dequantization=False, gaussian_dequantization=False),
        model=Namespace(ngf=ngf, num_classes=num_classes, sigma_begin=sigma_begin, sigma_end=sigma_end,
                        sigma_dist="geometric", normalization="InstanceNorm++", nonlinearity="elu", spec_norm=False),
        recons=Namespace(sigma_dist="geometric", sigma_begin=sigma_begin, sigma_end=sigma_end,
                         num_classes=num_classes),
        sampling=Namespace(n_steps_each=3, step_lr=9e-7, final_only=True, denoise=True))


@pytest.fixture(scope="module")
def pkg():
    assert torch.cuda.is_available()
    from inverseproblemwithdiffusionmodel_amd.ncsn.models import layers, ncsnv2, normalization, ALD_optimizers, proximal_op
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms import undersampling_fourier
    return Namespace(layers=layers, ncsnv2=ncsnv2, normalization=normalization, ald=ALD_optimizers,
                     prox=proximal_op, uf=undersampling_fourier)


def _load(module, g, prefix):
    sd = state_dict_from_golden(g, prefix)
    missing, unexpected = module.load_state_dict(sd, strict=True), None
    return module.cuda()


def test_instance_norm_module(pkg, golden):
    g = golden("g07_layers")
    m = _load(pkg.normalization.InstanceNorm2dPlus(6), g, "in")
    y = m(torch.from_numpy(g["in_x"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(y, g["in_y"], atol=3e-6)


@pytest.mark.parametrize("name,kw", [
    ("rb_plain", dict(input_dim=6, output_dim=6, resample=None)),
    ("rb_widen", dict(input_dim=6, output_dim=8, resample=None)),
    ("rb_pool", dict(input_dim=6, output_dim=8, resample="down")),
    ("rb_dil_down", dict(input_dim=6, output_dim=8, resample="down", dilation=2)),
    ("rb_dil_same", dict(input_dim=6, output_dim=6, resample=None, dilation=4)),
])
def test_residual_block(pkg, golden, name, kw):
    g = golden("g07_layers")
    act = pkg.layers._Act("elu")
    m = _load(pkg.layers.ResidualBlock(act=act, normalization=pkg.normalization.InstanceNorm2dPlus, **kw), g, name)
    y = m(torch.from_numpy(g["in_x"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(y, g[name + "_y"], atol=1e-5)


def test_refine_blocks(pkg, golden):
    g = golden("g07_layers")
    act = pkg.layers._Act("elu")
    xa, xb = torch.from_numpy(g["rf_xa"]).cuda(), torch.from_numpy(g["rf_xb"]).cuda()
    m = _load(pkg.layers.RefineBlock([6], 6, act=act, start=True), g, "rf_start")
    np.testing.assert_allclose(m([xa], xa.shape[2:]).cpu().numpy(), g["rf_start_y"], atol=2e-5)
    m = _load(pkg.layers.RefineBlock([6, 4], 5, act=act), g, "rf_two")
    np.testing.assert_allclose(m([xa, xb], xa.shape[2:]).cpu().numpy(), g["rf_two_y"], atol=2e-5)
    m = _load(pkg.layers.RefineBlock([6, 4], 6, act=act, end=True), g, "rf_end")
    np.testing.assert_allclose(m([xa, xb], xa.shape[2:]).cpu().numpy(), g["rf_end_y"], atol=2e-5)


@pytest.fixture(scope="module")
def tiny_net(pkg, golden):
    g = golden("g07_layers")
    net = pkg.ncsnv2.NCSNv2Deepest(tiny_config())
    net.load_state_dict(state_dict_from_golden(g, "net"), strict=True)
    return net.cuda().eval()


def test_tiny_ncsnv2_deepest(tiny_net, golden):
    g = golden("g07_layers")
    y = tiny_net(torch.from_numpy(g["net_x"]).cuda(), torch.from_numpy(g["net_labels"]).cuda()).cpu().numpy()
    ref = g["net_y"]
    assert np.abs(y - ref).max() <= 1e-4 * np.abs(ref).max()


def test_full_size_ncsnv2_deepest(pkg, golden):
    """ACDC network (ngf 128, 94.1 M parameters) at 128x128 vs the reference's own forward."""
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    g = golden("g15_fullnet")
    cfg = tiny_config(ngf=128, num_classes=2311, sigma_begin=348, sigma_end=0.01)
    net = pkg.ncsnv2.NCSNv2Deepest(cfg)
    keys = list(net.state_dict().keys())
    assert keys == list(g["key_names"])                       # same 230 state-dict keys, same order
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0)
    net.load_state_dict(sd, strict=False)
    net = net.cuda().eval()
    y = net(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["labels"]).cuda()).cpu().numpy()
    ref = g["y"]
    assert np.abs(y - ref).max() <= 2e-4 * np.abs(ref).max()
    assert metrics.nrmse(y, ref) < 1e-4
    # size-independent properties at the production size:
    # (1) per-image normalisation => a sample's score does not depend on what else is in the batch, bit for bit
    #     (persistent tile walk, Winograd / direct dispatch and LDS-DMA staging are all batch-independent)
    from inverseproblemwithdiffusionmodel_amd import ops
    xg, lg = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["labels"]).cuda()
    x3 = torch.cat([xg, xg[:1] * 0.5 + 3.0], dim=0)
    l3 = torch.cat([lg, lg[:1]], dim=0)
    y3 = net(x3, l3)
    assert torch.equal(y3[:xg.shape[0]].cpu(), torch.from_numpy(y))
    # (2) the two independently written kernel families (split-bf16 vs exact-fp32 MFMA, each with its own Winograd)
    #     agree far inside the parity tolerance
    impl = ops.CONV_IMPL
    try:
        ops.CONV_IMPL = "f32" if impl == "bx3" else "bx3"
        y_other = net(xg, lg).cpu().numpy()
    finally:
        ops.CONV_IMPL = impl
    assert metrics.nrmse(y_other, y) < 2e-5
    assert np.abs(y_other - ref).max() <= 2e-4 * np.abs(ref).max()


def test_packed_weight_cache_invalidation(pkg):
    """packed / Winograd-domain weight caches.  Entries are per (kernel family, layout), so toggling
    ops.CONV_IMPL can never return a blob of the other family; load_state_dict() and in-place (`copy_`) updates refresh
    them; a raw `.data` write (the reference's EMA swap) needs invalidate()."""
    from inverseproblemwithdiffusionmodel_amd import ops
    conv = pkg.layers.Conv2d(64, 64, 3).cuda()
    x = torch.randn(2, 64, 32, 32, device="cuda")
    ref = lambda: torch.nn.functional.conv2d(x.double(), conv.weight.data.double(), conv.bias.data.double(), padding=1)
    close = lambda y: float((y.double() - ref()).abs().max()) < 2e-5 * max(1.0, float(ref().abs().max()))
    assert close(conv(x))
    impl = ops.CONV_IMPL
    try:
        ops.CONV_IMPL = "f32" if impl == "bx3" else "bx3"          # other family: its own cache entry
        assert close(conv(x))
    finally:
        ops.CONV_IMPL = impl
    assert close(conv(x))
    with torch.no_grad():
        conv.weight.mul_(-0.5)                                      # in-place on the parameter: version bump
    assert close(conv(x))
    sd = {k: v.clone() * 3.0 for k, v in conv.state_dict().items()}
    conv.load_state_dict(sd)                                        # hook clears the cache
    assert close(conv(x))
    conv.weight.data.copy_(torch.randn_like(conv.weight.data) * 0.05)   # EMA-style write through .data: not seen ...
    stale = conv(x)
    conv.invalidate()                                                # ... until invalidated
    assert close(conv(x)) and not close(stale)


class _Tape:
    def __init__(self, tape):
        self.tape, self.i = tape, 0

    def __call__(self, like):
        n = torch.from_numpy(self.tape[self.i])
        self.i += 1
        return n


def _sense_sampler(pkg, tiny_net, golden, B=2):
    g = golden("g08_ald")
    op = pkg.uf.SENSE("exp", 4, 8, 0.04, (1, 32, 32), seed=0)
    sigmas = torch.from_numpy(g["sigmas"]).cuda()
    params = dict(n_steps_each=3, step_lr=9e-7, denoise=True, final_only=True)
    meas = torch.from_numpy(g["measurement"])[:, :B].cuda()
    sampler = pkg.ald.ALDInvSegProximalRealImag(pkg.prox.get_proximal("L2Penalty")(op), 1.0, "linear",
                                                (B, 1, 32, 32), tiny_net, sigmas, params, tiny_config(), meas, op,
                                                seg=None, device=torch.device("cuda"))
    return g, sampler


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("tag", ["dc_visible", "script_default"])
def test_ald_sense_trajectory_golden(pkg, tiny_net, golden, tag, use_graph):
    """the reference's own ALDInvSegProximalRealImag trajectory (60 noisy steps + denoise) under the same
    injected noise: NRMSE (reference definition) and SSIM of the magnitudes within 1e-3."""
    g, sampler = _sense_sampler(pkg, tiny_net, golden)
    tape = _Tape(g["noise"])
    x = sampler(label=None, lamda=0.1, save_dir=None, lr_scaled=float(g[f"{tag}_lr_scaled"]), seg_mode="full",
                noise_fn=tape, use_graph=use_graph)[0].numpy()
    assert tape.i == 60
    ref = g[f"{tag}_x"]
    assert x.shape == ref.shape and x.dtype == np.complex64
    for b in range(x.shape[0]):
        assert metrics.nrmse(np.abs(x[b]), np.abs(ref[b])) < 1e-3
        assert abs(metrics.ssim(np.abs(x[b, 0]), np.abs(ref[b, 0])) - 1.0) < 1e-3
    np.testing.assert_allclose(x, ref, atol=1e-3)


def test_ald_unconditional_golden(pkg, tiny_net, golden):
    g = golden("g08_ald")
    sigmas = torch.from_numpy(g["sigmas"]).cuda()
    params = dict(n_steps_each=3, step_lr=float(g["uncond_step_lr"]), denoise=True, final_only=True)
    sampler = pkg.ald.ALDUnconditionalSampler((2, 1, 32, 32), tiny_net, sigmas, params, tiny_config(),
                                              device=torch.device("cuda"))
    sampler.init_x_mod = lambda: torch.from_numpy(g["uncond_x0"]).cuda()
    x = sampler(noise_fn=_Tape(g["uncond_noise"]))[0].numpy()              # score forward replayed as a hipGraph
    np.testing.assert_allclose(x, g["uncond_x"], atol=1e-3)
    assert metrics.nrmse(x, g["uncond_x"]) < 1e-3
    x_eager = sampler(noise_fn=_Tape(g["uncond_noise"]), use_graph=False)[0].numpy()
    assert np.array_equal(x, x_eager)                                      # same launches, same bits


def test_philox_run_is_shard_invariant(pkg, tiny_net, golden):
    """4 samples on one rank == samples [0,2) and [2,4) on two ranks (global-sample-id Philox keys)."""
    g, s4 = _sense_sampler(pkg, tiny_net, golden, B=2)
    # build a 4-sample measurement by repeating the 2-sample one
    s4.measurement = torch.cat([s4.measurement, s4.measurement], dim=1)
    full = s4(lr_scaled=2e6, seed=123, sample_offset=0, n_levels=4)[0]
    _, s2 = _sense_sampler(pkg, tiny_net, golden, B=2)
    a = s2(lr_scaled=2e6, seed=123, sample_offset=0, n_levels=4)[0]
    b = s2(lr_scaled=2e6, seed=123, sample_offset=2, n_levels=4)[0]
    assert torch.equal(full[:2], a) and torch.equal(full[2:], b)
    assert not torch.equal(a, b)


class _SeededNoise:
    """the generator make_golden_r2.py used: randn(like.shape) call after call from Generator().manual_seed(seed)"""

    def __init__(self, seed):
        self.g, self.calls, self.total = torch.Generator().manual_seed(int(seed)), 0, 0.0

    def __call__(self, like):
        n = torch.randn(like.shape, generator=self.g, dtype=torch.float32)
        self.calls += 1
        self.total += float(n.double().sum())
        return n


class _LabelOffset(torch.nn.Module):
    """the window onto the 2311-level network the fixture used: net(x, labels + offset)"""

    def __init__(self, net, offset):
        super().__init__()
        self.net, self.offset = net, offset

    def forward(self, x, labels):
        return self.net(x, labels + self.offset)


@pytest.fixture(scope="module")
def full_net(pkg):
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    cfg = tiny_config(ngf=128, num_classes=2311, sigma_begin=348, sigma_end=0.01)
    cfg.data.image_size = 128
    net = pkg.ncsnv2.NCSNv2Deepest(cfg)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0), strict=False)
    return cfg, net.cuda().eval()


@pytest.mark.parametrize("tag,use_graph", [("tail_dc", True), ("tail_dc", False), ("tail_default", True),
                                           ("mid_default", True), ("mid_default", False)])
def test_headline_trajectory_vs_reference(pkg, golden, full_net, tag, use_graph):
    """THE headline configuration (128x128 complex, R = 40, 4 coils, NCSNv2Deepest ngf 128 = 94.1 M parameters, L2Penalty)
    against the REFERENCE's own ALDInvSegProximalRealImag runs (tests/golden/g20_fullsize_ald.npz): 12 noise levels x 3
    Langevin + proximal iterations + the denoising step under the same seeded noise stream, at the tail of the schedule
    (lr_scaled 2e6 and the script default 1) and in mid-schedule (sigma ~ 2.4: noise and score steps are O(0.1..1) of the
    image).  Everything that only runs at full size is in the loop: dilated Winograd at 16 px, split-K, the persistent
    LDS-DMA kernel, the batch-dependent dispatch, the hipGraph replay.  Tolerance: north_star's NRMSE / SSIM 1e-3 per
    sample, plus elementwise bounds on the reconstruction AND on the update x - x_init (so that a wrong network could not
    hide behind a small step)."""
    cfg, net = full_net
    g = golden("g20_fullsize_ald")
    lv0, lr_scaled, seed, n_calls, n_sum = g[f"{tag}_meta"]
    lv0 = int(lv0)
    B, H, W = 2, 128, 128
    op = pkg.uf.SENSE("exp", 4, 40, 0.04, (1, H, W), seed=0)
    assert np.array_equal(op.random_under_fourier.mask.numpy(), g["mask"])                 # bit-exact R=40 mask
    from oracle import kspace as okspace
    sig_all = torch.from_numpy(okspace.get_sigmas(348, 0.01, 2311))
    sig = sig_all[lv0:lv0 + 12].clone().cuda()
    meas = torch.from_numpy(g["measurement_1"]).repeat(1, B, 1, 1, 1).cuda()
    params = dict(n_steps_each=3, step_lr=9e-7, denoise=True, final_only=True)
    sampler = pkg.ald.ALDInvSegProximalRealImag(pkg.prox.get_proximal("L2Penalty")(op), 1.0, "linear", (B, 1, H, W),
                                                _LabelOffset(net, lv0), sig, params, cfg, meas, op, seg=None,
                                                device=torch.device("cuda"))
    noise = _SeededNoise(seed)
    x = sampler(label=None, lamda=0.1, save_dir=None, lr_scaled=float(lr_scaled), seg_mode="full", noise_fn=noise,
                use_graph=use_graph)[0].numpy()
    assert noise.calls == int(n_calls) == 72 and abs(noise.total - float(n_sum)) < 1e-2     # the reference's stream
    ref = g[f"{tag}_x"]
    assert x.shape == ref.shape == (B, 1, H, W) and np.isfinite(x).all()
    x0 = op.conj_op(meas).cpu().numpy()
    upd, upd_ref = x - x0, ref - x0
    scale = np.abs(ref).max()
    for b in range(B):
        assert metrics.nrmse(np.abs(x[b]), np.abs(ref[b])) < 1e-3
        assert abs(metrics.ssim(np.abs(x[b, 0]), np.abs(ref[b, 0])) - 1.0) < 1e-3
    np.testing.assert_allclose(x, ref, atol=1e-3 * scale)
    # the update itself: relative l2 error of (x - x_init) against the reference's
    assert np.linalg.norm(upd - upd_ref) <= 2e-3 * np.linalg.norm(upd_ref)
    assert np.linalg.norm(upd_ref) > 0.05 * np.linalg.norm(x0)                              # the run moved the image


G29_SNAPS = {"head": (), "long": (36, 100, 200)}


def _g29_run(pkg, full_net, g, tag, use_graph=True, impl=None):
    """the product sampler on one g29 window: the full 2311-level schedule, start_level / n_levels select the window
    (true step sizes), the fixture's seeded noise stream; -> (final x, {iteration: state}, x_init)"""
    from inverseproblemwithdiffusionmodel_amd import ops
    from oracle import kspace as okspace
    cfg, net = full_net
    lv0, n_lv, seed, n_calls, n_sum = g[f"{tag}_meta"]
    lv0, n_lv = int(lv0), int(n_lv)
    B, H, W = 2, 128, 128
    op = pkg.uf.SENSE("exp", 4, 40, 0.04, (1, H, W), seed=0)
    assert np.array_equal(op.random_under_fourier.mask.numpy(), g["mask"])
    sig_all = torch.from_numpy(okspace.get_sigmas(348, 0.01, 2311)).cuda()
    meas = torch.from_numpy(g["measurement_1"]).repeat(1, B, 1, 1, 1).cuda()
    params = dict(n_steps_each=3, step_lr=9e-7, denoise=False, final_only=True)
    sampler = pkg.ald.ALDInvSegProximalRealImag(pkg.prox.get_proximal("L2Penalty")(op), 1.0, "linear", (B, 1, H, W), net,
                                                sig_all, params, cfg, meas, op, seg=None, device=torch.device("cuda"))
    noise = _SeededNoise(seed)
    old = ops.CONV_IMPL
    try:
        if impl is not None:
            ops.CONV_IMPL = impl
        x = sampler(label=None, lamda=0.1, save_dir=None, lr_scaled=1.0, seg_mode="full", noise_fn=noise,
                    use_graph=use_graph, start_level=lv0, n_levels=n_lv, snapshot_its=G29_SNAPS[tag])[0].numpy()
    finally:
        ops.CONV_IMPL = old
    assert noise.calls == int(n_calls) == 6 * n_lv and abs(noise.total - float(n_sum)) < 5e-2      # the reference's stream
    snaps = {k: v.numpy() for k, v in sampler._snapshots.items()}
    return x, snaps, op.conj_op(meas).cpu().numpy()


@pytest.mark.parametrize("tag,use_graph", [("head", True), ("head", False), ("long", True), ("long", False)])
def test_headline_head_and_long_window_vs_reference(pkg, golden, full_net, tag, use_graph):
    """The headline configuration against the REFERENCE's own runs at the schedule's TRUE step sizes
    (tests/golden/g29_headline_long.npz, make_golden_r3.py): `head` = noise levels 0..11 (sigma ~ 348, step
    9e-7 (348/0.01)^2 ~ 1.09e3, noise scale ~ 47: the state grows to |x| ~ 1e3 -- where rounding differences are amplified
    most), `long` = 300 CONSECUTIVE iterations from level 1100 (sigma 2.4 -> 1.5; step 0.05 -> 0.02).  Tolerance:
    north_star's NRMSE / SSIM 1e-3 per sample at the end of the window and at every recorded intermediate state."""
    g = golden("g29_headline_long")
    x, snaps, x0 = _g29_run(pkg, full_net, g, tag, use_graph)
    ref = g[f"{tag}_x"]
    assert x.shape == ref.shape == (2, 1, 128, 128) and np.isfinite(x).all()
    checks = [(x, ref)] + [(snaps[k], g[f"{tag}_x_it{k}"]) for k in G29_SNAPS[tag]]
    for a, r in checks:
        for b in range(2):
            assert metrics.nrmse(np.abs(a[b]), np.abs(r[b])) < 1e-3
            assert abs(metrics.ssim(np.abs(a[b, 0]), np.abs(r[b, 0]), data_range=float(np.abs(r[b]).max())) - 1.0) < 1e-3
    assert np.linalg.norm(ref - x0) > 0.5 * np.linalg.norm(x0)                                 # the run moved the image
    np.testing.assert_allclose(x, ref, atol=2e-3 * np.abs(ref).max())


def test_error_growth_table(pkg, golden, full_net):
    """SURVEY.md section 7's sensitivity study as numbers: NRMSE of the complex state against the reference (fp32 torch CPU)
    after 36 / 100 / 200 / 300 iterations of the `long` window, for the default kernel family, the exact-fp32-MFMA family,
    and the float64 CPU oracle (g29f_sensitivity_f64.npz: NOT the reference -- it measures how far fp32 rounding alone
    moves the trajectory).  Written to gpurun_out/r03_error_growth.json; DESIGN.md section 2 carries the table."""
    import json, os
    from inverseproblemwithdiffusionmodel_amd import ops
    g, g64 = golden("g29_headline_long"), golden("g29f_sensitivity_f64")

    def err(a, r):
        return float(np.linalg.norm(a - r) / np.linalg.norm(r))

    table = {}
    its = list(G29_SNAPS["long"]) + [300]
    refs = {k: g[f"long_x_it{k}"] for k in G29_SNAPS["long"]}
    refs[300] = g["long_x"]
    for name, impl in [(ops.CONV_IMPL, None), ("f32", "f32")] + ([("bx3", "bx3")] if ops.CONV_IMPL != "bx3" else []):
        x, snaps, _ = _g29_run(pkg, full_net, g, "long", True, impl)
        snaps[300] = x
        table[name + "_vs_reference"] = {k: err(snaps[k], refs[k]) for k in its}
    f64 = {k: g64[f"long_x_it{k}"] for k in G29_SNAPS["long"]}
    f64[300] = g64["long_x"]
    table["float64_oracle_vs_reference"] = {k: err(f64[k], refs[k]) for k in its}
    xh, _, _ = _g29_run(pkg, full_net, g, "head", True)
    table["head_36_iterations"] = {ops.CONV_IMPL + "_vs_reference": err(xh, g["head_x"]),
                                   "float64_oracle_vs_reference": err(g64["head_x"], g["head_x"])}
    print(json.dumps(table, indent=1))
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/r03_error_growth.json", "w") as f:
            json.dump(table, f, indent=1)
    except OSError:
        pass
    # fp32 rounding alone (float64 oracle vs the fp32 reference) sets the scale; the HIP path must not be an order worse
    for name, row in table.items():
        if name.endswith("_vs_reference") and not name.startswith("float64"):
            for k in its:
                assert row[k] < max(1e-3, 10 * table["float64_oracle_vs_reference"][k]), (name, k, row[k])


@pytest.mark.parametrize("name,cls,size", [("v2_32", "NCSNv2", 32), ("v2_28", "NCSNv2", 28), ("deeper_32", "NCSNv2Deeper", 32)])
def test_ncsnv2_variants_golden(pkg, golden, name, cls, size):
    """NCSNv2 (incl. the 28-pixel branch of ncsnv2.py:50-56 = BASELINE config 1's literal size) and NCSNv2Deeper vs the
    reference's forwards; weights = synth_state_dict(seed=231) on the reference's own key/shape list"""
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    g = golden("g23_variants")
    cfg = tiny_config(ngf=4, num_classes=10)
    cfg.data.image_size = size
    net = getattr(pkg.ncsnv2, cls)(cfg)
    keys = [f"{k}:{','.join(map(str, v.shape))}" for k, v in net.state_dict().items()]
    assert keys == list(g[f"{name}_keys"])
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=231), strict=False)
    net = net.cuda().eval()
    y = net(torch.from_numpy(g[f"{name}_x"]).cuda(), torch.from_numpy(g[f"{name}_labels"]).cuda()).cpu().numpy()
    ref = g[f"{name}_y"]
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("name,cls", [("sde64", "NCSNv2"), ("sde128", "NCSNv2_128"), ("sde256", "NCSNv2_256")])
def test_score_sde_ncsnv2_adapters(golden, name, cls):
    """score_sde flavour (models/ncsnv2.py:43-416): ConfigDict schema, the reference classes' state-dict keys/shapes and
    float sigma table; forward pinned to the architecture the classes state (their ncsn-side twins on the same weights),
    because the reference's own forward raises at the first dilated block at every size (fixture flag, see
    make_golden_r2.py g23)."""
    from inverseproblemwithdiffusionmodel_amd.configs import ConfigDict
    from inverseproblemwithdiffusionmodel_amd.models import ncsnv2 as sde_ncsnv2, utils as mutils
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    g = golden("g23_variants")
    assert bool(g[f"{name}_reference_forward_raises"])
    size, centered, nf, scales, smax, smin = g[f"{name}_cfg"]
    c = ConfigDict()
    c.data = ConfigDict(image_size=int(size), centered=bool(centered), channels=1, num_channels=1)
    c.model = ConfigDict(nf=int(nf), num_scales=int(scales), sigma_max=float(smax), sigma_min=float(smin),
                         normalization="InstanceNorm++", nonlinearity="elu")
    c.device = torch.device("cuda")
    net = getattr(sde_ncsnv2, cls)(c)
    assert mutils.get_model({"sde64": "ncsnv2_64", "sde128": "ncsnv2_128", "sde256": "ncsnv2_256"}[name]) is getattr(sde_ncsnv2, cls)
    keys = [f"{k}:{','.join(map(str, v.shape))}" for k, v in net.state_dict().items()]
    assert sorted(keys) == sorted(g[f"{name}_keys"])
    np.testing.assert_allclose(net.sigmas.cpu().numpy(), g[f"{name}_sigmas"], rtol=1e-6)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=234), strict=False)
    net = net.cuda().eval()
    y = net(torch.from_numpy(g[f"{name}_x"]).cuda(), torch.from_numpy(g[f"{name}_labels"]).cuda()).float().cpu().numpy()
    ref = g[f"{name}_y"]
    assert np.abs(y - ref).max() <= 1e-4 * np.abs(ref).max()
    # get_network picks the class by image size (models/ncsnv2.py:31-40)
    for sz, want in [(64, "NCSNv2"), (128, "NCSNv2_128"), (256, "NCSNv2_256")]:
        c.data.image_size = sz
        assert sde_ncsnv2.get_network(c).func is getattr(sde_ncsnv2, want)


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("tag,prox_name", [("l2_dc", "L2Penalty"), ("l2_default", "L2Penalty"),
                                           ("closed_dc", "SingleCoil"), ("closed_default", "SingleCoil")])
def test_singlecoil_trajectory_golden(pkg, tiny_net, golden, tag, prox_name, use_graph):
    """single-coil sampler (scripts/acdc_inv_seg_sampling_keep_center_prox_real_imag.py:79-89: RandomUndersamplingFourier +
    get_proximal(...)) vs the reference's own 30-step + denoise trajectories (g26), fused single-coil iteration tail"""
    g = golden("g26_singlecoil")
    sc = pkg.uf.RandomUndersamplingFourier(8, 0.04, (1, 32, 32), seed=2)
    assert np.array_equal(sc.mask.numpy(), g["mask"])
    lr_scaled, n_calls, n_sum = g[f"{tag}_meta"]
    sigmas = torch.from_numpy(g["sigmas"]).cuda()
    params = dict(n_steps_each=3, step_lr=9e-7, denoise=True, final_only=True)
    meas = torch.from_numpy(g["measurement"]).cuda()
    sampler = pkg.ald.ALDInvSegProximalRealImag(pkg.prox.get_proximal(prox_name)(sc), 1.0, "linear", (2, 1, 32, 32),
                                                tiny_net, sigmas, params, tiny_config(), meas, sc, seg=None,
                                                device=torch.device("cuda"))
    noise = _SeededNoise(260)
    x = sampler(label=None, lamda=0.1, save_dir=None, lr_scaled=float(lr_scaled), seg_mode="full", noise_fn=noise,
                use_graph=use_graph)[0].numpy()
    assert noise.calls == int(n_calls) and abs(noise.total - float(n_sum)) < 1e-3
    ref = g[f"{tag}_x"]
    for b in range(2):
        assert metrics.nrmse(np.abs(x[b]), np.abs(ref[b])) < 1e-3
        assert abs(metrics.ssim(np.abs(x[b, 0]), np.abs(ref[b, 0])) - 1.0) < 1e-3
    np.testing.assert_allclose(x, ref, atol=1e-3)
    if tag.endswith("_dc"):
        assert np.abs(ref - g[tag.replace("_dc", "_default") + "_x"]).max() > 1e-2         # data consistency is visible


def test_sampler_rejects_unbuilt_combinations(pkg, tiny_net, golden):
    g = golden("g26_singlecoil")
    sc = pkg.uf.RandomUndersamplingFourier(8, 0.04, (1, 32, 32), seed=2)
    mk = lambda prox: pkg.ald.ALDInvSegProximalRealImag(
        prox, 1.0, "linear", (2, 1, 32, 32), tiny_net, torch.from_numpy(g["sigmas"]).cuda(),
        dict(n_steps_each=3, step_lr=9e-7, denoise=True, final_only=True), tiny_config(),
        torch.from_numpy(g["measurement"]).cuda(), sc, seg=None, device=torch.device("cuda"))
    with pytest.raises(TypeError):                 # the reference's 4-argument proximal call cannot take Constrained
        mk(pkg.prox.get_proximal("Constrained")(sc))(lr_scaled=1.0)


def test_full_size_trajectory_bf16x3_vs_fp32_mfma():
    """The headline workload (128x128, R = 40, 4 coils, ngf-128 network, L2Penalty) over the LAST 12 noise levels
    (36 Langevin + proximal iterations and the denoising step, Philox noise): the split-bf16 kernel family and the
    exact-fp32-MFMA family give the same reconstructions to the metric north_star names -- NRMSE / SSIM to 1e-3 --
    with two orders of magnitude to spare."""
    from inverseproblemwithdiffusionmodel_amd import engine, ops
    dev = torch.device("cuda")
    prob = engine.build_problem(dev, 2)
    L = len(prob.sigmas)
    kw = dict(prob.call_kwargs, seed=7, start_level=L - 12)

    def run(impl):
        old = ops.CONV_IMPL
        try:
            ops.CONV_IMPL = impl
            return prob.sampler(**kw)[0].numpy()
        finally:
            ops.CONV_IMPL = old

    b = run("f32")
    for impl in ("hx2", "bx3"):                          # two fp16 pieces / three bf16 pieces vs the exact-fp32 MFMA kernels
        a = run(impl)
        assert np.isfinite(a).all() and a.shape == (2, 1, 128, 128)
        for i in range(2):
            ma, mb = np.abs(a[i]), np.abs(b[i])
            assert metrics.nrmse(ma, mb) < 1e-5, impl
            assert abs(metrics.ssim(ma[0], mb[0]) - 1.0) < 1e-5, impl
        assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max(), impl


@pytest.mark.parametrize("tag", ["a", "b"])
def test_map_sense_golden_gpu(pkg, tiny_net, golden, tag):
    """MAP baseline SENSEMAP (50 Adam iterations) on the GPU kernels vs the reference's own run"""
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.MAP_optimizers import SENSEMAP
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import SENSE
    g = golden("g18_map")
    op = SENSE("exp", 4, 8, 0.04, (1, 32, 32), seed=0)
    cfg = tiny_config()
    cfg.MAP = Namespace(n_iters=50, lr=float(g[f"{tag}_lr"]), complex_inner_n_steps=20)
    x_init = torch.from_numpy(g[f"{tag}_x_init"].copy()).cuda()
    opt = SENSEMAP(x_init, torch.from_numpy(g["measurement"]).cuda(), tiny_net, op, float(g[f"{tag}_lamda"]), cfg,
                   logger=None, device=torch.device("cuda"))
    x = opt().cpu().numpy()
    ref = g[f"{tag}_x"]
    np.testing.assert_allclose(x, ref, atol=0.02 * 50 * float(g[f"{tag}_lr"]))      # see test_oracle_golden.py
    assert metrics.nrmse(np.abs(x), np.abs(ref)) < 1e-3
    assert np.array_equal(x_init.cpu().numpy(), x)                # updated in place, as the reference's parameter
