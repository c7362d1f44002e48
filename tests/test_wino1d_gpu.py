"""The 1-D Winograd convolution kernel (csrc/conv_wino1d.hip: F(2,3) along x, the filter's rows as K, f16x2 operands) against
float64 and against the 2-D Winograd kernel it stands in for: every epilogue option the score networks use (bias, residual,
residual into the activated copy only, ELU / copy activations, output scale, per-image maxima, dynamic input range, statistics
partials), ragged pixel blocks, and the layer dispatch (`IPDM_WINO1D`)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from inverseproblemwithdiffusionmodel_amd import ops as o
    return o


def _case(B, Cin, Cout, H, W, seed=7, wscale=None):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=gen).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * (wscale or 1.0 / (3 * Cin ** 0.5))).cuda()
    b = torch.randn(Cout, generator=gen).cuda()
    r = torch.randn(B, Cout, H, W, generator=gen).cuda()
    return x, w, b, r


SHAPES = [(1, 32, 128, 8, 32), (2, 64, 128, 16, 64), (3, 128, 256, 40, 36), (2, 32, 128, 10, 44), (1, 64, 384, 24, 96),
          (2, 128, 128, 128, 128)]


@pytest.mark.parametrize("B,Cin,Cout,H,W", SHAPES)
def test_wino1d_against_float64(ops, B, Cin, Cout, H, W):
    """fp32-faithful: max error <= 1e-6 of the output range on every shape, ragged 8 x 32 blocks included (H = 10, 40; W = 36,
    44); the activated copy is ELU of the same values"""
    x, w, b, r = _case(B, Cin, Cout, H, W)
    assert ops._lib.lib.ipdm_conv2d_wino1d_supported(Cin, Cout, H, W) == 1
    U = ops.conv_wino1d_weight(w)
    assert U.kk == 12 and U.fmt == "hx2"
    y, ya = ops.conv2d_wino_bx3(x, U, b, r, act_out=ops.ACT_ELU)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double()
    scale = ref.abs().max()
    assert (y.double() - ref).abs().max() <= 1e-6 * scale
    assert (ya.double() - F.elu(ref)).abs().max() <= 1e-6 * scale
    y1 = ops.conv2d_wino_bx3(x, U, b, r)                              # one output: another instantiation, the same bits
    assert torch.equal(y1, y)
    none, ya1 = ops.conv2d_wino_bx3(x, U, b, r, act_out=ops.ACT_ELU, raw=False)
    assert none is None and torch.equal(ya1, ya)
    y0 = ops.conv2d_wino_bx3(x, U)                                    # no bias, no residual
    ref0 = F.conv2d(x.double(), w.double(), padding=1)
    assert (y0.double() - ref0).abs().max() <= 1e-6 * ref0.abs().max()


def test_wino1d_matches_2d_kernel_to_rounding(ops):
    """the two Winograd forms agree to a few fp32 roundings of the range (they are different summation orders of one sum)"""
    x, w, b, r = _case(2, 128, 128, 32, 64, seed=9)
    y1 = ops.conv2d_wino_bx3(x, ops.conv_wino1d_weight(w), b, r)
    y2 = ops.conv2d_wino_bx3(x, ops.conv_wino_hx2_weight(w), b, r)
    assert (y1 - y2).abs().max() <= 1e-6 * y2.abs().max()


def test_wino1d_residual_into_activated_copy_only(ops):
    """res_second (CRPBlock: path = conv(pool(path)); x = path + x): out = conv + bias, out_act = conv + bias + residual"""
    x, w, b, r = _case(2, 64, 128, 16, 64, seed=11)
    U = ops.conv_wino1d_weight(w)
    path, xs = ops.conv2d_wino_bx3(x, U, b, r, act_out=ops.ACT_COPY, res_second=True)
    plain = ops.conv2d_wino_bx3(x, U, b)
    assert torch.equal(path, plain)
    assert torch.equal(xs, plain + r)
    _, e = ops.conv2d_wino_bx3(x, U, b, r, act_out=ops.ACT_ELU, res_second=True)
    assert (e - F.elu(plain + r)).abs().max() <= 2e-6 * (plain + r).abs().max()


def test_wino1d_output_scale_and_per_image_bias(ops):
    x, w, b, r = _case(3, 32, 128, 16, 32, seed=12)
    U = ops.conv_wino1d_weight(w)
    rows = torch.randn(3, 128, generator=torch.Generator().manual_seed(1)).cuda()
    y = ops.conv2d_wino_bx3(x, U, rows, r, out_scale=0.70710678)
    ref = (F.conv2d(x.double(), w.double(), padding=1) + rows.double()[:, :, None, None] + r.double()) * 0.70710678
    assert (y.double() - ref).abs().max() <= 1e-6 * ref.abs().max()


def test_wino1d_unsupported_shapes_and_activations(ops):
    lib = ops._lib.lib
    assert lib.ipdm_conv2d_wino1d_supported(128, 64, 64, 64) == 0          # 128 output channels per workgroup
    assert lib.ipdm_conv2d_wino1d_supported(48, 128, 64, 64) == 0          # K chunks in pairs
    assert lib.ipdm_conv2d_wino1d_supported(128, 128, 16, 16) == 0         # rows of 32 pixels or more
    assert lib.ipdm_conv2d_wino1d_supported(128, 128, 64, 34) == 0         # 16-byte row pieces
    assert lib.ipdm_conv2d_wino1d_stats_partials(128, 128, 16, 16) == 0
    x, w, b, r = _case(1, 32, 128, 8, 32)
    U = ops.conv_wino1d_weight(w)
    with pytest.raises(ops._lib.IpdmUnsupported):
        ops.conv2d_wino_bx3(x, U, b, act_out=ops.ACT_SWISH)               # the epilogue's activations: ELU and copy
    with pytest.raises(ValueError):
        ops.conv2d_wino_bx3(x, U, b, dilation=2)


@pytest.mark.parametrize("B,Cin,Cout,H,W,res", [(2, 64, 128, 64, 32, True), (1, 32, 128, 34, 44, False), (2, 128, 256, 16, 64, True)])
def test_wino1d_pooled_epilogue(ops, B, Cin, Cout, H, W, res):
    """ConvMeanPool in one launch (layers.py:291-313): 2x2 mean of the convolution in the reference's summation order, pooled-size
    residual, activated copy, maxima and statistics of the pooled result; ragged blocks (H = 34, W = 44)"""
    gen = torch.Generator().manual_seed(43)
    x = torch.randn(B, Cin, H, W, generator=gen).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.1).cuda()
    b = torch.randn(Cout, generator=gen).cuda()
    r = (torch.randn(B, Cout, H // 2, W // 2, generator=gen) * 2 + 3).cuda() if res else None
    U = ops.conv_wino1d_weight(w)
    with ops.amax_scope():
        y, ya = ops.conv2d_wino_bx3(x, U, b, r, act_out=ops.ACT_ELU, pool2=True, want_stats=True, want_amax=True)
        conv = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        ref = (conv[..., ::2, ::2] + conv[..., 1::2, ::2] + conv[..., ::2, 1::2] + conv[..., 1::2, 1::2]) / 4
        if res:
            ref = ref + r.double()
        assert tuple(y.shape) == (B, Cout, H // 2, W // 2)
        assert (y.double() - ref).abs().max() <= 1e-6 * ref.abs().max()
        assert (ya.double() - F.elu(ref)).abs().max() <= 1e-6 * ref.abs().max()
        assert torch.equal(ops.amax_value(ops.amax_of(y)), y.abs().amax(dim=(1, 2, 3)))
        assert torch.equal(ops.amax_value(ops.amax_of(ya)), ya.abs().amax(dim=(1, 2, 3)))
    part, _ = y._ipdm_partials
    assert float(part[..., 0].sum(dim=2).min()) == (H // 2) * (W // 2) == float(part[..., 0].sum(dim=2).max())
    ones = torch.ones(Cout).cuda()
    c_part = ops.instnorm_plus_coef(y, ones, ones, None)
    c_full = ops.instnorm_plus_coef(y, ones, ones, None)                   # partials consumed: a pass over the tensor
    assert (c_part - c_full).abs().max() <= 2e-5 * c_full.abs().max()
    y1 = ops.conv2d_wino_bx3(x, U, b, r, pool2=True)                       # plain pooled launch: the same bits
    assert torch.equal(y1, y)
    y2 = ops.conv2d_wino_bx3(x, ops.conv_wino_hx2_weight(w), b, r, pool2=True)      # the 2-D kernel's pooled epilogue
    assert (y1 - y2).abs().max() <= 1e-6 * y2.abs().max()


@pytest.mark.parametrize("scale", [1e-5, 1.0, 3e4, 1e8])
def test_wino1d_dynamic_range_and_maxima(ops, scale):
    """per-image dynamic input scale: any fp32 magnitude is in range, images of one batch at different magnitudes included;
    the epilogue's maxima vectors hold exactly max |stored value| per image"""
    x, w, b, r = _case(3, 64, 128, 16, 64, seed=13)
    x = x * torch.tensor([scale, 1.0, scale * 3.0]).cuda()[:, None, None, None]
    U = ops.conv_wino1d_weight(w)
    with ops.amax_scope():
        y, ya = ops.conv2d_wino_bx3(x, U, b, None, act_out=ops.ACT_ELU, in_amax=True, want_amax=True)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        for i in range(3):
            assert (y[i].double() - ref[i]).abs().max() <= 1e-6 * ref[i].abs().max()
        am_y, am_a = ops.amax_of(y), ops.amax_of(ya)
        assert am_y is not None and am_a is not None
        assert torch.equal(ops.amax_value(am_y), y.abs().amax(dim=(1, 2, 3)))
        assert torch.equal(ops.amax_value(am_a), ya.abs().amax(dim=(1, 2, 3)))


@pytest.mark.parametrize("B,Cin,Cout,H,W,res", [(2, 128, 128, 64, 64, True), (3, 32, 128, 40, 36, False), (2, 64, 256, 24, 96, True)])
def test_wino1d_statistics_epilogue(ops, B, Cin, Cout, H, W, res):
    """the statistics epilogue: InstanceNorm++ coefficients from its partials equal those from a pass over the stored tensor
    (ragged blocks, residual with a mean far from zero); the result is bit-identical with and without the epilogue"""
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(B, Cin, H, W, generator=gen).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.1).cuda()
    b = torch.randn(Cout, generator=gen).cuda()
    r = (torch.randn(B, Cout, H, W, generator=gen) * 3 + 5).cuda() if res else None
    U = ops.conv_wino1d_weight(w)
    P = int(ops._lib.lib.ipdm_conv2d_wino1d_stats_partials(Cin, Cout, H, W))
    assert P == ((W + 31) // 32) * ((H + 7) // 8)
    y0 = ops.conv2d_wino_bx3(x, U, b, r)
    y1, y1a = ops.conv2d_wino_bx3(x, U, b, r, act_out=ops.ACT_ELU, want_stats=True)
    assert torch.equal(y0, y1) and hasattr(y1, "_ipdm_partials")
    part, tag = y1._ipdm_partials
    assert tuple(part.shape) == (B, Cout, P, 3) and tag == (y1._version, y1.data_ptr())
    assert float(part[..., 0].sum(dim=2).min()) == H * W == float(part[..., 0].sum(dim=2).max())
    alpha, gamma, beta = (torch.randn(Cout, generator=gen).cuda() for _ in range(3))
    c_part = ops.instnorm_plus_coef(y1, alpha, gamma, beta)
    c_full = ops.instnorm_plus_coef(y0, alpha, gamma, beta)
    assert (c_part - c_full).abs().max() <= 2e-5 * c_full.abs().max()
    yd = y0.double()
    mean = yd.mean(dim=(2, 3))
    rstd = 1.0 / torch.sqrt(yd.var(dim=(2, 3), unbiased=False) + 1e-5)
    assert (c_part[..., 0].double() - mean).abs().max() <= 1e-6 * mean.abs().max()
    assert ((c_part[..., 1] / gamma).double() / rstd - 1).abs().max() <= 1e-5
    y2 = ops.conv2d_wino_bx3(x, U, b, r, want_stats=True)                  # one output + statistics: its own instantiation
    assert torch.equal(y2, y0)
    c2 = ops.instnorm_plus_coef(y2, alpha, gamma, beta)
    assert torch.equal(c2, c_part)


def test_wino1d_statistics_large_mean(ops):
    """planes with |mean| >> spread (1000 +- 1): the one-pass sums are taken about a shift from the data"""
    gen = torch.Generator().manual_seed(42)
    B, Cin, Cout, H, W = 2, 32, 128, 40, 36
    x = torch.randn(B, Cin, H, W, generator=gen).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05).cuda()
    r = (torch.randn(B, Cout, H, W, generator=gen) + 1000.0).cuda()
    y = ops.conv2d_wino_bx3(x, ops.conv_wino1d_weight(w), None, r, want_stats=True)
    ones = torch.ones(Cout).cuda()
    c_part = ops.instnorm_plus_coef(y, ones, ones, None)
    yd = y.double()
    rstd = 1.0 / torch.sqrt(yd.var(dim=(2, 3), unbiased=False) + 1e-5)
    assert (c_part[..., 0].double() - yd.mean(dim=(2, 3))).abs().max() < 1e-3
    assert (c_part[..., 1].double() / rstd - 1).abs().max() < 2e-4


def test_wino1d_batch_invariance_and_determinism(ops):
    """a sample's bits do not depend on the batch it shares a launch with (sharding.py's invariance), nor on the run"""
    x, w, b, r = _case(5, 64, 128, 16, 64, seed=15)
    U = ops.conv_wino1d_weight(w)
    y = ops.conv2d_wino_bx3(x, U, b, r)
    assert torch.equal(y, ops.conv2d_wino_bx3(x, U, b, r))
    for i in (0, 3):
        assert torch.equal(y[i:i + 1], ops.conv2d_wino_bx3(x[i:i + 1].contiguous(), U, b, r[i:i + 1].contiguous()))


def test_wino1d_layer_dispatch(ops, monkeypatch):
    """ncsn Conv2d: with the switch on, eligible shapes run the 1-D kernel (its blob is cached next to the 2-D one), others and
    the pooled form keep the 2-D kernel; results agree to rounding either way"""
    from inverseproblemwithdiffusionmodel_amd.ncsn.models import layers
    if ops.CONV_IMPL != "hx2":
        pytest.skip("the 1-D kernel belongs to the f16x2 family")
    conv = layers.Conv2d(64, 128, 3).cuda()
    x = torch.randn(2, 64, 32, 32, generator=torch.Generator().manual_seed(3)).cuda()
    monkeypatch.setattr(ops, "WINO1D", False)
    y2 = conv(x)
    monkeypatch.setattr(ops, "WINO1D", True)
    ops.CONV_TRACE = []
    try:
        y1 = conv(x)
        small = conv(x[:, :, :16, :16].contiguous())                      # 16-pixel rows: not the 1-D kernel's
        kinds = [bool(t.get("wino1d")) for t in ops.CONV_TRACE]
    finally:
        ops.CONV_TRACE = None
    assert kinds == [True, False]
    assert (y1 - y2).abs().max() <= 1e-6 * y2.abs().max()
    assert tuple(small.shape) == (2, 128, 16, 16)


@pytest.mark.parametrize("B,Cin,Cout,H,W,pool", [(2, 64, 128, 16, 64, False), (3, 32, 128, 40, 36, False), (2, 128, 128, 64, 64, False),
                                                 (2, 64, 128, 34, 44, True)])
def test_wino1d_fused_input_norm_and_elu(ops, B, Cin, Cout, H, W, pool):
    """the input as (raw x, InstanceNorm++ coefficients): ELU((x - mu) * scale + shift) applied in the producer, the padding kept
    at zero (ragged blocks: rows and columns beyond the image) -- against the separate affine + activation pass followed by the
    same kernel, and against float64; statistics, residual, activated copy and the dynamic range (the coefficient bound) ride
    along"""
    gen = torch.Generator().manual_seed(51)
    x = (torch.randn(B, Cin, H, W, generator=gen) * 3 + 1).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.1).cuda()
    b = torch.randn(Cout, generator=gen).cuda()
    oh, ow = (H // 2, W // 2) if pool else (H, W)
    r = torch.randn(B, Cout, oh, ow, generator=gen).cuda()
    alpha, gamma, beta = (torch.randn(Cin, generator=gen).cuda() for _ in range(3))
    U = ops.conv_wino1d_weight(w)
    with ops.amax_scope():
        coef = ops.instnorm_plus_coef(x, alpha, gamma, beta)
        bound = getattr(coef, "_ipdm_amax_bound", None)
        a = ops.affine_act(x, coef, ops.ACT_ELU)
        y_sep, ya_sep = ops.conv2d_wino_bx3(a, U, b, r, act_out=ops.ACT_ELU, pool2=pool, in_amax=bound)
        y, ya = ops.conv2d_wino_bx3(x, U, b, r, act_out=ops.ACT_ELU, pool2=pool, want_stats=True, in_amax=bound, coef=coef,
                                    act=ops.ACT_ELU)
    ad = F.elu((x.double() - coef[..., 0, None, None].double()) * coef[..., 1, None, None].double() + coef[..., 2, None, None].double())
    conv = F.conv2d(ad, w.double(), b.double(), padding=1)
    ref = (conv[..., ::2, ::2] + conv[..., 1::2, ::2] + conv[..., ::2, 1::2] + conv[..., 1::2, 1::2]) / 4 if pool else conv
    ref = ref + r.double()
    assert (y.double() - ref).abs().max() <= 2e-6 * ref.abs().max()
    assert (y - y_sep).abs().max() <= 2e-6 * y_sep.abs().max()          # (the fused affine contracts one multiply-add)
    assert (ya - ya_sep).abs().max() <= 2e-6 * y_sep.abs().max()
    assert hasattr(y, "_ipdm_partials")
    with pytest.raises(ValueError):
        ops.conv2d_wino_bx3(x, ops.conv_wino_hx2_weight(w), b, coef=coef, act=ops.ACT_ELU)      # the 2-D kernel has no fused input
    with pytest.raises(ValueError):
        ops.conv2d_wino_bx3(x, U, b, coef=coef, act=ops.ACT_RELU)


def test_residual_block_fused_norms_match_separate_passes(ops, monkeypatch):
    """ResidualBlock (plain and down-sampling): with the normalisations applied inside the convolutions the block's result equals
    the separate-pass form to rounding"""
    from inverseproblemwithdiffusionmodel_amd.ncsn.models import layers
    if ops.CONV_IMPL != "hx2":
        pytest.skip("f16x2 family")
    gen = torch.Generator().manual_seed(52)
    x = torch.randn(2, 128, 32, 64, generator=gen).cuda()
    for resample, cout in ((None, 128), ("down", 256)):
        blk = layers.ResidualBlock(128, cout, resample=resample, act=torch.nn.ELU()).cuda()
        for p_ in blk.parameters():
            p_.data = (0.2 * torch.randn(p_.shape, generator=gen)).cuda() + (1.0 if p_.dim() == 1 else 0.0)
        with torch.no_grad():
            monkeypatch.setattr(ops, "WINO1D_FIN", False)
            with ops.amax_scope():
                y0 = blk(x)
            monkeypatch.setattr(ops, "WINO1D_FIN", True)
            ops.CONV_TRACE = []
            try:
                with ops.amax_scope():
                    y1 = blk(x)
                n = len(ops.CONV_TRACE)
            finally:
                ops.CONV_TRACE = None
        assert n >= 2
        assert (y1 - y0).abs().max() <= 5e-6 * y0.abs().max()


@pytest.mark.parametrize("B,Cin,Cout,D,H,W", [(2, 32, 128, 5, 8, 12), (1, 64, 128, 4, 8, 24), (2, 32, 128, 3, 8, 8), (1, 32, 256, 6, 16, 12),
                                              (1, 32, 128, 1, 8, 12), (2, 32, 128, 2, 10, 16), (1, 32, 128, 3, 8, 40)])
def test_wino1d_volume_form(ops, B, Cin, Cout, D, H, W):
    """3x3x3 'same' convolution on the 1-D Winograd kernel: depth taps as K chunks (zero planes outside the volume), two depth
    slices per row block on planes of 12 pixels or less (odd depths: the last block holds one), every epilogue option of the
    temporal network"""
    gen = torch.Generator().manual_seed(61)
    x = torch.randn(B, Cin, D, H, W, generator=gen).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (5 * Cin ** 0.5)).cuda()
    b = torch.randn(Cout, generator=gen).cuda()
    r = torch.randn(B, Cout, D, H, W, generator=gen).cuda()
    assert ops._lib.lib.ipdm_conv3d_wino1d_supported(Cin, Cout, D, H, W) == 1
    U = ops.conv_wino1d_weight3d(w)
    with ops.amax_scope():
        y, ya = ops.conv3d(x, U, b, residual=r, act_out=ops.ACT_ELU, in_amax=True, want_amax=True)
        ref = F.conv3d(x.double(), w.double(), b.double(), padding=1) + r.double()
        assert (y.double() - ref).abs().max() <= 1e-6 * ref.abs().max()
        assert (ya.double() - F.elu(ref)).abs().max() <= 1e-6 * ref.abs().max()
        assert torch.equal(ops.amax_value(ops.amax_of(y)), y.abs().amax(dim=(1, 2, 3, 4)))
        assert torch.equal(ops.amax_value(ops.amax_of(ya)), ya.abs().amax(dim=(1, 2, 3, 4)))
    y1 = ops.conv3d(x, U, b, residual=r, in_amax=True)                    # one output: another instantiation, the same bits
    assert torch.equal(y1, y)
    path, xs = ops.conv3d(x, U, b, residual=r, act_out=ops.ACT_COPY, res_second=True)
    plain = ops.conv3d(x, U, b)
    assert torch.equal(path, plain) and torch.equal(xs, plain + r)
    yd = ops.conv3d(x, ops.conv_bx3_weight(w, fmt="hx2"), b, residual=r)             # the direct kernel it stands in for
    assert (y1 - yd).abs().max() <= 1e-6 * yd.abs().max()


def test_wino1d_linearity_and_shift_at_full_size(ops):
    """size-independent properties at the headline layer's full size (B = 28, 128 -> 128 @128^2, where no float64 reference
    finishes in seconds): linearity conv(a x1 + b x2) = a conv(x1) + b conv(x2), and equivariance under a shift by one 8 x 32 pixel
    block (interior pixels: another workgroup, another position in its tile list, the same sum)"""
    gen = torch.Generator(device="cuda").manual_seed(81)
    B, C, H, W = 28, 128, 128, 128
    x1 = torch.randn(B, C, H, W, device="cuda", generator=gen)
    x2 = torch.randn(B, C, H, W, device="cuda", generator=gen)
    w = torch.randn(C, C, 3, 3, device="cuda", generator=gen) / (3 * C ** 0.5)
    U = ops.conv_wino1d_weight(w)
    y1, y2 = ops.conv2d_wino_bx3(x1, U), ops.conv2d_wino_bx3(x2, U)
    y12 = ops.conv2d_wino_bx3(0.5 * x1 - 2.0 * x2, U)                  # (power-of-two coefficients: the combination is exact)
    ref = 0.5 * y1 - 2.0 * y2
    assert (y12 - ref).abs().max() <= 2e-6 * ref.abs().max()
    xs = torch.roll(x1, shifts=(8, 32), dims=(2, 3))
    ys = ops.conv2d_wino_bx3(xs, U)
    assert torch.equal(ys[:, :, 10:-2, 34:-2], torch.roll(y1, shifts=(8, 32), dims=(2, 3))[:, :, 10:-2, 34:-2])
    assert torch.isfinite(y12).all()
