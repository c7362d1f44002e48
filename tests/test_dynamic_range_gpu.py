"""The f16x2 family's DYNAMIC RANGE without a pass over any tensor.

A convolution of the default family splits fp32 operands into two fp16 pieces; each image of its input is scaled into fp16's
range by an exact power of two derived from an upper bound of that image's max |x|.  The bounds come from the producers:
convolution / resize epilogues accumulate the exact maxima of what they store (atomic max: exact, order-independent),
InstanceNorm++ hands over a bound computed from its coefficients, pooling / activations / gathers pass their input's bound on.
Checked here:
  * every producer's maxima are EXACT (or a true upper bound, for the normalisation) -- all kernel forms;
  * kernel level: static contract vs dynamic range at input scales 1e-3 .. 1e-5 and 1e5 against float64;
  * network level: NCSNv2Deepest (tiny and the full-size 94 M-parameter one) and NCSN3DShallow with the un-normalised stream
    rescaled to ~1e-4 and ~1e5, against the float64 CPU oracle on the same weights, at the golden tests' tolerance;
  * the hot path never measures a tensor (ops.AMAX_MEASURED stays put) and fills its slots once per evaluation.
The reference computes in fp32 (ncsn/models/layers.py:28-60 nn.Conv2d): 'any fp32 input' is the contract to meet."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import scorenet as oracle_net
from test_scorenet_gpu import tiny_config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from inverseproblemwithdiffusionmodel_amd import ops
    return ops


def _amax(t):
    return t.abs().amax(dim=tuple(range(1, t.dim()))).float()


# (B, Cin, Cout, H, W, dilation, kind): every epilogue that stores a convolution result
FORMS = [
    (3, 64, 64, 32, 64, 1, "wino"),            # wide persistent kernel, 16 x 4 tile block
    (2, 32, 64, 40, 36, 1, "wino"),            # ... ragged edges
    (3, 64, 64, 16, 16, 1, "wino"),            # small-image kernel
    (40, 32, 256, 16, 16, 1, "wino"),          # 8 x 8 persistent form
    (3, 256, 256, 16, 16, 1, "wino"),          # Winograd split-K + reduce pass
    (2, 64, 128, 16, 16, 2, "wino"),           # polyphase (dilated 16 x 16)
    (1, 16, 64, 24, 16, 2, "wino"),            # register-staged dilated kernel
    (2, 64, 128, 32, 32, 1, "pool"),           # pooled epilogue (ConvMeanPool)
    (2, 64, 64, 32, 64, 1, "stats"),           # statistics epilogue
    (2, 32, 48, 20, 24, 1, "direct"),          # direct kernel, ragged channels
    (2, 64, 64, 16, 16, 4, "direct"),          # direct, dilated 16-pixel tiles
    (1, 256, 64, 8, 16, 1, "direct"),          # direct split-K + reduce pass
    (2, 64, 32, 24, 40, 1, "direct1"),         # 1 x 1
]


@pytest.mark.parametrize("fmt", ["hx2", "bx3"])
@pytest.mark.parametrize("B,Cin,Cout,H,W,dil,kind", FORMS)
def test_producer_maxima_are_exact(ops, B, Cin, Cout, H, W, dil, kind, fmt):
    """out / out_act of a convolution called with want_amax carry max |stored value| per image, bit for bit"""
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(B, Cin, H, W, generator=gen)
    x[0] *= 37.0                                              # images of one batch differ
    k = 1 if kind == "direct1" else 3
    w = (torch.randn(Cout, Cin, k, k, generator=gen) / (k * k * Cin) ** 0.5).cuda()
    b = torch.randn(Cout, generator=gen).cuda()
    pool = kind == "pool"
    oh, ow = (H // 2, W // 2) if pool else (H, W)
    res = torch.randn(B, Cout, oh, ow, generator=gen).cuda()
    xg = x.cuda()
    if kind in ("wino", "pool", "stats"):
        assert ops.conv_wino_bx3_supported(Cin, Cout, H, W, dil)
        U = ops.conv_wino_bx3_weight(w, fmt=fmt)
        out, act = ops.conv2d_wino_bx3(xg, U, b, res, act_out=ops.ACT_ELU, dilation=dil, pool2=pool, want_stats=kind == "stats",
                                       in_amax=True if fmt == "hx2" else None, want_amax=True)
        only = ops.conv2d_wino_bx3(xg, U, b, res, act_out=ops.ACT_ELU, raw=False, dilation=dil, pool2=pool, want_amax=True)[1]
    else:
        wq = ops.conv_bx3_weight(w, fmt=fmt)
        out, act = ops.conv_bx3(xg, wq, b, residual=res, dilation=dil, act_out=ops.ACT_ELU, want_amax=True)
        only = ops.conv_bx3(xg, wq, b, residual=res, dilation=dil, act_out=ops.ACT_ELU, raw=False, want_amax=True)[1]
    for t in (out, act, only):
        am = ops.amax_of(t)
        assert am is not None and torch.equal(ops.amax_value(am), _amax(t)), (kind, fmt)
    assert ops.amax_of(ops.maxpool5(act)) is ops.amax_of(act)             # pooling passes the bound on
    out.add_(1.0)                                                        # written in place: the maxima are gone
    assert ops.amax_of(out) is None


def test_producer_maxima_3d_resize_and_norm_bound(ops):
    gen = torch.Generator().manual_seed(42)
    x3 = torch.randn(3, 16, 8, 8, 24, generator=gen)
    x3[1] *= 1e-3
    w3 = (torch.randn(32, 16, 3, 3, 3, generator=gen) / (27 * 16) ** 0.5).cuda()
    for fmt in ("hx2", "bx3"):
        out, act = ops.conv_bx3(x3.cuda(), ops.conv_bx3_weight(w3, fmt=fmt), dilation=2, act_out=ops.ACT_ELU, want_amax=True,
                                in_amax=True if fmt == "hx2" else None)
        assert torch.equal(ops.amax_value(ops.amax_of(out)), _amax(out))
        assert torch.equal(ops.amax_value(ops.amax_of(act)), _amax(act))
    assert ops.amax_of(ops.maxpool3d5(act)) is ops.amax_of(act)
    assert ops.amax_of(ops.temporal_taps(act, 0)) is ops.amax_of(act)
    # resize + accumulate + activation: the three 2-D kernels (LDS strips, float4 gather, generic) and the 3-D one
    for (C, ih, iw, oh, ow) in [(8, 16, 16, 32, 32), (8, 32, 32, 16, 16), (5, 9, 7, 11, 13)]:
        src = torch.randn(3, C, ih, iw, generator=gen).cuda()
        base = (torch.randn(3, C, oh, ow, generator=gen) * torch.tensor([1.0, 50.0, 1e-3]).view(3, 1, 1, 1)).cuda()
        got = ops.bilinear(src, (oh, ow), out=base, accumulate=True, act=ops.ACT_ELU, want_amax=True)
        assert torch.equal(ops.amax_value(ops.amax_of(got)), _amax(got))
    v = torch.randn(2, 4, 4, 6, 8, generator=gen).cuda()
    got = ops.trilinear(v, (8, 8, 12), act=ops.ACT_ELU, want_amax=True)
    assert torch.equal(ops.amax_value(ops.amax_of(got)), _amax(got))
    # InstanceNorm++: a BOUND from the coefficients (never below the true maximum, within sqrt(HW) of it)
    if ops.dynamic_range():
        xn = torch.randn(3, 12, 16, 16, generator=gen).cuda() * 1e4
        al, ga, be = (torch.randn(12, generator=gen).cuda() for _ in range(3))
        y = ops.affine_act(xn, ops.instnorm_plus_coef(xn, al, ga, be), ops.ACT_ELU)
        bound = ops.amax_value(ops.amax_of(y))
        assert bound is not None and (bound >= _amax(y)).all() and (bound <= 16.0 * 4 * _amax(y) + 64).all()


@pytest.mark.parametrize("scale", [1.0e-3, 1.0e-4, 1.0e-5, 1.0e5])
def test_static_vs_dynamic_range_against_float64(ops, scale):
    """kernel level (K = 1152, the census shape of VERDICT r3's emulation): the dynamic range holds 1e-5 of the output range at
    every input magnitude; the static contract degrades below |x| ~ 2^-3 (absolute 2^-25 operand floor) and overflows above
    65504.  The static numbers are recorded, not required."""
    gen = torch.Generator().manual_seed(50)
    rec = {}
    for name, (B, Cin, Cout, H, W, dil, wino) in dict(wide=(2, 128, 128, 64, 64, 1, True), small=(2, 128, 128, 16, 16, 1, True),
                                                       poly=(2, 128, 128, 16, 16, 2, True), direct=(2, 128, 64, 24, 24, 1, False),
                                                       splitk=(2, 256, 256, 16, 16, 1, True)).items():
        x = F.elu(torch.randn(B, Cin, H, W, generator=gen)) * scale
        w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (9 * Cin) ** 0.5
        want = F.conv2d(x.double(), w.double(), padding=dil, dilation=dil)
        rng = float(want.abs().max())
        xg, wg = x.cuda(), w.cuda()
        if wino:
            U = ops.conv_wino_hx2_weight(wg)
            dyn = ops.conv2d_wino_bx3(xg, U, dilation=dil, in_amax=ops.absmax_per_image(xg))
            sta = ops.conv2d_wino_bx3(xg, U, dilation=dil)
        else:
            wq = ops.conv_hx2_weight(wg)
            dyn = ops.conv_bx3(xg, wq, dilation=dil, in_amax=ops.absmax_per_image(xg))
            sta = ops.conv_bx3(xg, wq, dilation=dil)
        e_dyn = float((dyn.cpu().double() - want).abs().max()) / rng
        e_sta = float((sta.cpu().double() - want).abs().max()) / rng if torch.isfinite(sta).all() else float("inf")
        rec[name] = dict(dynamic=e_dyn, static=e_sta)
        assert e_dyn <= 1e-5, (name, scale, e_dyn)
        # a bound 100x too large (what a coefficient bound may be) still holds the tolerance
        loose = ops.absmax_per_image(xg) * 100.0                 # (every way scaled: still a maxima vector)
        d2 = ops.conv2d_wino_bx3(xg, U, dilation=dil, in_amax=loose) if wino else ops.conv_bx3(xg, wq, dilation=dil, in_amax=loose)
        assert float((d2.cpu().double() - want).abs().max()) / rng <= 1e-5, (name, scale)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"r04_dynamic_range_scale_{scale:g}.json"), "w") as f:
            json.dump(rec, f, indent=1)


def _rescale_unnormalised_stream(sd, s):
    """multiply what the un-normalised stream of an NCSNv2 / NCSN3D carries by s: begin_conv and every ResidualBlock's conv2 scale
    with s (weights and biases), shortcut convolutions are linear in an already scaled input (their biases scale).  The
    normalised branches are scale-free, so every layer output -- what the RefineNet RCU / CRP / MSF / shortcut convolutions read --
    sits at s times its previous magnitude."""
    out = {}
    for k, v in sd.items():
        v = v.clone()
        blk = k.split(".")
        if k.startswith("begin_conv."):
            v *= s
        elif blk[0].startswith("res") and "conv2" in blk:
            v *= s
        elif blk[0].startswith("res") and "shortcut" in blk and blk[-1] == "bias":
            v *= s
        elif k.startswith("conv_temporal_down.") or k.startswith("conv_temporal_up."):
            pass
        out[k] = v
    return out


def _check_net(ops, net, sd, x, labels, oracle, tol):
    """HIP forward vs the float64 oracle on the same weights; the hot path must not measure any tensor"""
    net.load_state_dict(sd, strict=False)
    before = ops.AMAX_MEASURED
    y = net(x.cuda(), labels.cuda())
    assert ops.AMAX_MEASURED == before, "a convolution of the hot path measured its input (extra pass)"
    assert torch.isfinite(y).all()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in net.state_dict().items()}
    sd64 = {k: v.cpu() for k, v in sd64.items()}
    with torch.no_grad():
        ref = oracle(x.double(), labels, sd64)
    err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err <= tol, err
    return err


@pytest.mark.parametrize("s", [1.0, 1.0e-4, 1.0e5])
def test_tiny_network_with_rescaled_stream_vs_float64(ops, s):
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.ncsnv2 import NCSNv2Deepest
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    cfg = tiny_config(ngf=16, num_classes=12, sigma_begin=2.0)
    net = NCSNv2Deepest(cfg).cuda().eval()
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=4)
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(3, 1, 32, 32, generator=gen) * torch.tensor([1.0, 300.0, 1e-2]).view(3, 1, 1, 1)
    labels = torch.tensor([0, 5, 11])
    _check_net(ops, net, _rescale_unnormalised_stream(sd, s), x, labels, oracle_net.ncsnv2_deepest, 1e-4)


@pytest.mark.parametrize("s", [1.0e-4, 1.0e5])
def test_full_size_network_with_rescaled_stream_vs_float64(ops, s):
    """the headline network (ngf 128, 94.1 M parameters, 128 x 128): same tolerance as the reference-pinned g15 test"""
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.ncsnv2 import NCSNv2Deepest
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    cfg = tiny_config(ngf=128, num_classes=2311, sigma_begin=348, sigma_end=0.01)
    net = NCSNv2Deepest(cfg).cuda().eval()
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(1, 1, 128, 128, generator=gen) * 40.0
    labels = torch.tensor([700])
    err_dyn = _check_net(ops, net, _rescale_unnormalised_stream(sd, s), x, labels, oracle_net.ncsnv2_deepest, 2e-4)
    # what the static contract (round 3's default) does on the same weights: recorded, required only to be no better
    rec = dict(scale=s, dynamic=err_dyn)
    try:
        ops.HX2_DYNAMIC = False
        y = net(x.cuda(), labels.cuda())
        sd64 = {k: (v.double().cpu() if v.is_floating_point() else v.cpu()) for k, v in net.state_dict().items()}
        with torch.no_grad():
            ref = oracle_net.ncsnv2_deepest(x.double(), labels, sd64)
        rec["static"] = float((y.cpu().double() - ref).abs().max() / ref.abs().max()) if torch.isfinite(y).all() else float("inf")
    finally:
        ops.HX2_DYNAMIC = True
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"r04_dynamic_range_fullnet_{s:g}.json"), "w") as f:
            json.dump(rec, f)


def test_one_fill_per_evaluation_and_graph_replay(ops):
    """the maxima slots of one evaluation come out of one zeroed block allocated INSIDE the evaluation, so a captured forward
    re-zeroes them on every replay: replaying a graph on new inputs gives the eager result, bit for bit"""
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.ncsnv2 import NCSNv2Deepest
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    cfg = tiny_config(ngf=16, num_classes=12, sigma_begin=2.0)
    net = NCSNv2Deepest(cfg).cuda().eval()
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=4), strict=False)
    gen = torch.Generator().manual_seed(10)
    xs = [torch.randn(2, 1, 32, 32, generator=gen).cuda() * a for a in (1.0, 500.0, 1e-3)]
    labels = torch.tensor([1, 7]).cuda()
    eager = [net(x, labels).clone() for x in xs]
    stat = xs[1].clone()                                       # capture on the LARGE input: stale maxima would then be too big
    net(stat, labels)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = net(stat, labels)
    for x, want in zip((xs[2], xs[0], xs[1]), (eager[2], eager[0], eager[1])):
        stat.copy_(x)
        g.replay()
        assert torch.equal(y, want)


def test_temporal_network_never_measures(ops):
    """NCSN3DShallow (config 4): every convolution gets its maxima from a producer, including the temporal (1,1,4) convolutions
    through their tap gather; rescaled stream vs the float64 oracle"""
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.ncsn3d import NCSN3DShallow
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    from oracle import scorenet3d
    cfg = tiny_config(ngf=16, num_classes=12, sigma_begin=2.0)
    cfg.data.channels, cfg.data.channels_3d = 64, 1
    net = NCSN3DShallow(cfg).cuda().eval()
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=6)
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(2, 64, 24, generator=gen) * torch.tensor([1.0, 200.0]).view(2, 1, 1)
    labels = torch.tensor([3, 9])
    for s in (1.0, 1.0e-4, 1.0e5):
        _check_net(ops, net, _rescale_unnormalised_stream(sd, s), x, labels, scorenet3d.ncsn3d_shallow, 1e-4)


def test_raw_pointer_writes_invalidate_attached_maxima(ops):
    """the kernels write through raw pointers, which torch's version counters do not see by themselves: every op that stores
    into an existing tensor bumps the counter (ops._written), so maxima attached to ANY alias of that storage go stale.  The
    sampler state is the case that matters: it is updated in place by the fused Langevin / proximal kernels every iteration and
    read by the segmentation network's first convolution through persistent views."""
    x = torch.randn(4, 1, 16, 16, device="cuda")
    lo, hi = x[:2], x[2:]                                          # persistent views, as the samplers keep them
    am = ops.in_amax_for(lo, "hx2", always=True)                   # measured and attached
    assert ops.amax_of(lo) is am and torch.equal(ops.amax_value(am), lo.abs().amax(dim=(1, 2, 3)))
    ops.langevin_step(hi, torch.ones_like(hi), step=100.0)         # writes the OTHER half of the storage: same counter
    assert ops.amax_of(lo) is None
    am2 = ops.in_amax_for(lo, "hx2", always=True)
    assert ops.amax_of(lo) is am2
    ops.axpy_sched(lo, torch.ones_like(lo), scale=1e3)             # in place on the tagged view itself
    assert ops.amax_of(lo) is None and float(lo.abs().max()) > 500
    y = torch.randn(2, 3, 8, 8, device="cuda")
    ops.tag_amax(y, ops.absmax_per_image(y))
    ops.add(y, y, out=y)
    assert ops.amax_of(y) is None
    z = torch.randn(2, 3, 8, 8, device="cuda")
    ops.tag_amax(z, ops.absmax_per_image(z))
    ops.act(z, ops.ACT_ELU, out=z)                                 # an activation in place keeps a valid BOUND
    assert ops.amax_of(z) is not None and (ops.amax_value(ops.amax_of(z)) >= z.abs().amax(dim=(1, 2, 3))).all()
