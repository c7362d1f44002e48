"""Segmentation-likelihood guidance on the GPU: the glue kernels against torch, the UNet's
launch-chain forward and input-gradient against the torch-CPU restatement + autograd (oracle/seg_unet.py), and the guided
SENSE sampler against the CPU oracle sampler with the same guidance term.  PARITY UNPINNED against MONAI itself (not
installed, not vendored by the reference: SURVEY.md 8c); what is pinned is the architecture as published and the reference's
own compute_seg_grad / adjust_grad formulas (ncsn/models/__init__.py:197-215, ALD_optimizers.py:272-286)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import state_dict_from_golden
from oracle import seg_unet as oseg, scorenet as oracle_net, ald as oracle_ald, kspace, metrics
from test_scorenet_gpu import tiny_config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from inverseproblemwithdiffusionmodel_amd import ops
    return ops


def test_seg_glue_kernels_vs_torch(ops):
    gen = torch.Generator().manual_seed(50)
    x = torch.randn(3, 5, 6, 10, generator=gen)
    z = ops.zero_insert2(x.cuda()).cpu()
    want = torch.zeros(3, 5, 12, 20)
    want[..., ::2, ::2] = x
    assert torch.equal(z, want)
    assert torch.equal(ops.subsample2(z.cuda()).cpu(), x)
    # InstanceNorm + PReLU forward / backward vs autograd (float64)
    a = torch.tensor([0.17])
    xd = (x.double() * 3 + 1).requires_grad_(True)
    y = F.prelu(F.instance_norm(xd, eps=1e-5), a.double())
    gy = torch.randn(x.shape, generator=gen)
    y.backward(gy.double())
    xhat, yk, rstd = ops.in_prelu_fwd((x * 3 + 1).cuda(), a.cuda())
    assert float((yk.cpu().double() - y.detach()).abs().max()) < 2e-6
    gx = ops.in_prelu_bwd(gy.cuda(), xhat, rstd, a.cuda()).cpu().double()
    assert float((gx - xd.grad).abs().max()) < 5e-6 * float(xd.grad.abs().max())
    # log-likelihood gradient at the logits
    logits = torch.randn(2, 3, 4, 5, generator=gen) * 4
    lab = torch.randint(0, 3, (2, 1, 4, 5), generator=gen)
    ld = logits.double().requires_grad_(True)
    torch.log(torch.gather(torch.softmax(ld, 1), 1, lab)).sum().backward()
    g = ops.seg_loglh_grad(logits.cuda(), lab.cuda()).cpu().double()
    assert float((g - ld.grad).abs().max()) < 2e-6
    # schedule-scaled accumulate
    yv, xv = torch.randn(2, 7, generator=gen), torch.randn(2, 7, generator=gen)
    sched = np.zeros(1, dtype=[("step", "f4"), ("ns", "f4"), ("coef", "f4"), ("sigma", "f4"), ("id", "i8"), ("seg", "f4"), ("rsv", "f4")])
    sched["seg"] = 0.375
    got = ops.axpy_sched(yv.cuda().clone(), xv.cuda(), dev_sched=torch.from_numpy(sched.view(np.uint8)).cuda()).cpu()
    assert torch.equal(got, yv + xv * 0.375)
    m = torch.tensor([1, 0, 1, 1, 0, 0, 1])
    got = ops.axpy_sched(yv.cuda().clone(), xv.cuda(), scale=2.0, mask=m.cuda()).cpu()
    assert torch.equal(got, yv + xv * m.float() * 2.0)


def _pair(channels, strides, seed):
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.seg_unet import UNet
    torch.manual_seed(seed)
    ref = oseg.UNet(channels=channels, strides=strides).eval()
    for n_, p_ in ref.named_parameters():                    # make biases / slopes non-trivial
        if p_.ndim == 1:
            p_.data.add_(0.1 * torch.randn_like(p_))
    net = UNet(channels=channels, strides=strides)
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())        # MONAI's parameter names, same order
    net.load_state_dict(ref.state_dict(), strict=True)
    return ref, net.cuda().eval()


@pytest.mark.parametrize("channels,strides,size,B", [((8, 16, 32), (2, 2), 32, 3), ((16, 32, 64, 128, 256), (2, 2, 2, 2), 64, 2)])
def test_unet_forward_and_loglh_grad_vs_autograd(channels, strides, size, B):
    ref, net = _pair(channels, strides, 51)
    gen = torch.Generator().manual_seed(52)
    x = torch.rand(B, 1, size, size, generator=gen) * 2 - 1
    lab = (torch.rand(B, 1, size, size, generator=gen) > 0.7).long()
    with torch.no_grad():
        want = ref(x)
    got = net(x.cuda()).cpu()
    assert float((got - want).abs().max()) < 2e-4 * float(want.abs().max())
    for mode in ("full", "FG"):
        g_ref = oseg.compute_seg_grad(ref, x, lab, mode)
        g = net.loglh_grad(x.cuda(), lab.cuda(), mode).cpu()
        assert g.shape == x.shape
        assert float((g - g_ref).abs().max()) < 1e-3 * float(g_ref.abs().max()), mode
        assert metrics.nrmse(g.numpy(), g_ref.numpy()) < 1e-3


def test_full_size_unet_loglh_grad():
    """the reference's Seg configuration (channels 64..1024, 10.5 M parameters) at 128x128 vs torch autograd on the CPU"""
    ref, net = _pair((64, 128, 256, 512, 1024), (2, 2, 2, 2), 53)
    gen = torch.Generator().manual_seed(54)
    x = torch.rand(2, 1, 128, 128, generator=gen) * 2 - 1
    lab = torch.zeros(2, 1, 128, 128, dtype=torch.long)
    lab[:, :, 40:80, 50:90] = 1
    g_ref = oseg.compute_seg_grad(ref, x, lab, "full")
    g = net.loglh_grad(x.cuda(), lab.cuda(), "full").cpu()
    assert metrics.nrmse(g.numpy(), g_ref.numpy()) < 1e-3
    assert float((g - g_ref).abs().max()) < 2e-3 * float(g_ref.abs().max())


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("seg_mode", ["full", "FG"])
def test_guided_sense_trajectory_vs_oracle(golden, seg_mode, use_graph):
    """ALDInvSegProximalRealImag with seg_start_time = 0 (the reference script's default: guidance ramps in from the first
    level): 10 levels x 3 steps + denoise on the tiny score net of g07 with a small UNet, same injected noise, against
    the CPU oracle sampler that adds compute_seg_grad / sigma * lh_weight to both score planes"""
    from inverseproblemwithdiffusionmodel_amd.ncsn.models import ncsnv2, ALD_optimizers, proximal_op
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import SENSE
    g7, g8 = golden("g07_layers"), golden("g08_ald")
    net2d = ncsnv2.NCSNv2Deepest(tiny_config())
    net2d.load_state_dict(state_dict_from_golden(g7, "net"), strict=True)
    net2d = net2d.cuda().eval()
    ref_seg, seg = _pair((8, 16, 32), (2, 2), 55)
    op = SENSE("exp", 4, 8, 0.04, (1, 32, 32), seed=0)
    B = 2
    sigmas = torch.from_numpy(g8["sigmas"]).cuda()
    meas = torch.from_numpy(g8["measurement"])[:, :B].cuda()
    params = dict(n_steps_each=3, step_lr=9e-7, denoise=True, final_only=True)
    label = torch.zeros(B, 1, 32, 32, dtype=torch.long)
    label[:, :, 8:20, 10:24] = 1
    sampler = ALD_optimizers.ALDInvSegProximalRealImag(proximal_op.get_proximal("L2Penalty")(op), 0.0, "linear", (B, 1, 32, 32),
                                                       net2d, sigmas, params, tiny_config(), meas, op, seg=seg,
                                                       device=torch.device("cuda"))
    assert float(sampler.lh_weights[-1]) == 1.0 and float(sampler.lh_weights[0]) == 0.0
    tape = [torch.from_numpy(t) for t in g8["noise"]]
    it = iter(tape)
    x = sampler(label=label, lamda=0.1, save_dir=None, lr_scaled=2.0e6, seg_mode=seg_mode, noise_fn=lambda like: next(it),
                use_graph=use_graph)[0].numpy()
    sd_cpu = state_dict_from_golden(g7, "net")
    it2 = iter(tape)
    lhw = ALD_optimizers.get_lh_weights(torch.from_numpy(g8["sigmas"]), 0.0, "linear")
    torch.set_num_threads(4)          # hundreds of tiny CPU ops per step: a many-core default pool only adds hand-off latency
    with torch.no_grad():
        want = oracle_ald.ald_sense_real_imag(
            lambda x_, lab: oracle_net.ncsnv2_deepest(x_, lab, sd_cpu), g8["sigmas"], g8["measurement"][:, :B],
            kspace.sens_maps(4, 32, 32, 0), op.random_under_fourier.mask.numpy(), 9e-7, 3, 2.0e6, True, lambda like: next(it2),
            seg_grad_fn=lambda x_: oseg.compute_seg_grad(ref_seg, x_, label, seg_mode), lh_weights=lhw)
    # the guidance term matters in this run (same run without it differs visibly) ...
    assert np.abs(want - g8["dc_visible_x"]).max() > 1e-3
    # ... and the HIP chain reproduces it
    for b in range(B):
        assert metrics.nrmse(np.abs(x[b]), np.abs(want[b])) < 1e-3
        assert abs(metrics.ssim(np.abs(x[b, 0]), np.abs(want[b, 0])) - 1.0) < 1e-3
    np.testing.assert_allclose(x, want, atol=2e-3)


def test_sampler_needs_fused_seg_network(golden):
    from inverseproblemwithdiffusionmodel_amd.ncsn.models import ncsnv2, ALD_optimizers, proximal_op
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import SENSE
    g8 = golden("g08_ald")
    op = SENSE("exp", 4, 8, 0.04, (1, 32, 32), seed=0)
    s = ALD_optimizers.ALDInvSegProximalRealImag(proximal_op.get_proximal("L2Penalty")(op), 0.5, "linear", (2, 1, 32, 32),
                                                 None, torch.from_numpy(g8["sigmas"]).cuda(), {}, tiny_config(), None, op,
                                                 seg=torch.nn.Conv2d(1, 2, 1), device=torch.device("cuda"))
    with pytest.raises(NotImplementedError):
        s._check_fast_path({})
