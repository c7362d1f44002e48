"""Parity of the individual HIP kernels (through the C ABI, via ops.py) against the CPU oracle and
the golden fixtures.  Tolerances are fp32 round-off (stated per test); integer/index logic is exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import state_dict_from_golden
from oracle import kspace, resample, scorenet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from inverseproblemwithdiffusionmodel_amd import ops as _ops
    return _ops


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def test_cpu_tensor_is_rejected(ops):
    with pytest.raises(RuntimeError):
        ops.act(torch.zeros(4), ops.ACT_ELU)


# ---- upfirdn2d / bias-act ---------------------------------------------------------------------
@pytest.mark.parametrize("name", ["down2", "up2", "down2_nonsq", "up2_nonsq", "up3_down2_k5", "negpad_k3",
                                  "up1_down3_k2x4", "up2_down1_k6"])
def test_upfirdn2d_golden(ops, golden, name):
    g = golden("g09_upfirdn")
    x, k = g[f"{name}_x"], g[f"{name}_k"]
    up, down, p0, p1 = (int(v) for v in g[f"{name}_udp"])
    N, C, H, W = x.shape
    y = ops.upfirdn2d_raw(dev(x).reshape(N * C, H, W, 1), dev(k), up, up, down, down, p0, p1, p0, p1)
    y = y.reshape(N, C, y.shape[1], y.shape[2]).cpu().numpy()
    assert y.shape == g[f"{name}_y"].shape
    np.testing.assert_allclose(y, g[f"{name}_y"], atol=2e-6)          # fp32 round-off of <= 36 taps


@pytest.mark.parametrize("shape,mode", [((2, 128, 128, 128), "down"), ((2, 128, 128, 128), "up"),
                                        ((1, 3, 256, 256), "down"), ((3, 256, 4, 4), "up"), ((1, 5, 37, 53), "down"),
                                        ((1, 5, 37, 53), "up"), ((1, 2, 70, 130), "same")])
def test_upfirdn2d_bench_shapes(ops, shape, mode):
    rng = np.random.default_rng(1)
    x = rng.standard_normal(shape).astype(np.float32)
    k = resample.setup_kernel([1, 3, 3, 1])
    if mode == "down":
        args = (1, 1, 2, 2, 1, 1, 1, 1)
    elif mode == "up":
        k = k * 4
        args = (2, 2, 1, 1, 2, 1, 2, 1)
    else:
        args = (1, 1, 1, 1, 2, 1, 2, 1)
    ref = resample.upfirdn2d(x, k, *args)
    N, C, H, W = shape
    y = ops.upfirdn2d_raw(dev(x).reshape(N * C, H, W, 1), dev(k), *args)
    np.testing.assert_allclose(y.reshape(ref.shape).cpu().numpy(), ref, atol=3e-6)


@pytest.mark.parametrize("shape,mode,ksize,pad1", [
    ((3, 5, 12, 20), "down", (4, 4), 1),      # width 20 -> 10 outputs would break float4 rows: tiled fallback
    ((3, 5, 12, 24), "down", (4, 4), 1),      # 12 output columns = 3 float4 column groups
    ((2, 7, 9, 8), "down", (4, 4), 1),        # odd height (rows beyond the image are zero padding)
    ((2, 3, 16, 16), "down", (3, 2), 1),      # smaller tap grid, zero-extended at the high end
    ((2, 3, 16, 16), "down", (4, 4), 3),      # larger pad1: one more output row / column of zero padding
    ((2, 3, 16, 16), "down", (4, 4), -1),     # negative pad1 crops
    ((3, 5, 6, 4), "up", (4, 4), 1),          # 8 output columns = exactly one column group
    ((3, 5, 7, 12), "up", (4, 4), 1),         # 24 output columns = 3 groups, odd input height
    ((2, 3, 8, 8), "up", (2, 3), 1),
    ((2, 3, 8, 6), "up", (4, 4), 1),          # 12 output columns: the second float4 of the last group is outside
    ((40, 16, 64, 64), "down", (4, 4), 1),    # long strips (several row groups per thread, ring rotation)
    ((40, 16, 32, 32), "up", (4, 4), 1),
])
def test_upfirdn2d_stream_kernel(ops, shape, mode, ksize, pad1):
    """the register-streaming FIR kernel (down2 pad0 1 / up2 pad0 2 on float4-shaped rows) and its fallbacks vs the oracle,
    with a NON-symmetric random tap grid so that a flipped or transposed tap would show"""
    rng = np.random.default_rng(11)
    x = rng.standard_normal(shape).astype(np.float32)
    k = rng.standard_normal(ksize).astype(np.float32)
    args = (1, 1, 2, 2, 1, pad1, 1, pad1) if mode == "down" else (2, 2, 1, 1, 2, pad1, 2, pad1)
    ref = resample.upfirdn2d(x, k, *args)
    N, C, H, W = shape
    y = ops.upfirdn2d_raw(dev(x).reshape(N * C, H, W, 1), dev(k), *args)
    np.testing.assert_allclose(y.reshape(ref.shape).cpu().numpy(), ref, atol=1e-5)
    # linearity + shift structure (size-independent properties): a one-hot input reproduces the taps
    if mode == "down" and ksize == (4, 4) and pad1 == 1 and H >= 8:
        e = np.zeros((1, 1, H, W), np.float32)
        e[0, 0, 4, 4] = 1.0
        ye = ops.upfirdn2d_raw(dev(e).reshape(1, H, W, 1), dev(k), *args).reshape(H // 2, W // 2).cpu().numpy()
        # out[oy, ox] = sum in[2oy + ky - 1, 2ox + kx - 1] * k[3 - ky, 3 - kx]  ->  taps land at oy in {1, 2}, ox in {1, 2}
        want = np.zeros_like(ye)
        for oy in range(H // 2):
            for ox in range(W // 2):
                ky, kx = 4 - 2 * oy + 1, 4 - 2 * ox + 1
                if 0 <= ky < 4 and 0 <= kx < 4:
                    want[oy, ox] = k[3 - ky, 3 - kx]
        np.testing.assert_allclose(ye, want, atol=1e-7)


def test_upfirdn2d_minor_dim(ops):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 9, 8, 4)).astype(np.float32)          # [major, h, w, minor]
    k = rng.standard_normal((3, 3)).astype(np.float32)
    ref = resample.upfirdn2d(np.transpose(x, (0, 3, 1, 2)), k, 2, 2, 1, 1, 1, 1, 1, 1)
    y = ops.upfirdn2d_raw(dev(x), dev(k), 2, 2, 1, 1, 1, 1, 1, 1).cpu().numpy()
    np.testing.assert_allclose(np.transpose(y, (0, 3, 1, 2)), ref, atol=3e-6)


def test_fused_bias_act(ops, golden):
    g = golden("g10_biasact")
    y = ops.fused_bias_act_raw(dev(g["x"]), dev(g["b"]), None, 3, 0, 0.2, 2 ** 0.5).cpu().numpy()
    np.testing.assert_allclose(y, g["y_default"], atol=1e-6)
    y = ops.fused_bias_act_raw(dev(g["x2"]), dev(g["b2"]), None, 3, 0, 0.2, 2 ** 0.5).cpu().numpy()
    np.testing.assert_allclose(y, g["y2"], atol=1e-6)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 6, 8, 8)).astype(np.float32)
    b = rng.standard_normal(6).astype(np.float32)
    ref = rng.standard_normal(x.shape).astype(np.float32)
    for act, grad in [(1, 0), (1, 1), (1, 2), (3, 0), (3, 1), (3, 2)]:
        want = resample.bias_act(x, b, ref, act, grad, 0.3, 1.7)
        got = ops.fused_bias_act_raw(dev(x), dev(b), dev(ref), act, grad, 0.3, 1.7).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=1e-6, err_msg=f"act={act} grad={grad}")
    assert ops.fused_bias_act_raw(dev(np.zeros((0, 3), np.float32)), dev(b[:3]), None, 3, 0, 0.2, 1.0).numel() == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.float64])
@pytest.mark.parametrize("name", ["down2", "up2", "down2_nonsq", "up2_nonsq", "up3_down2_k5", "negpad_k3",
                                  "up1_down3_k2x4", "up2_down1_k6"])
def test_upfirdn2d_golden_half_and_double(ops, golden, name, dtype):
    """the other storage types of the reference's dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF, op/upfirdn2d_kernel.cu:311):
    the g09 vectors cast to half / double.  Half: inputs and taps rounded to half first (what a half tensor holds), result
    within one half rounding of the exact FIR of those rounded operands; double: 1e-12."""
    g = golden("g09_upfirdn")
    x, k = torch.from_numpy(g[f"{name}_x"]).to(dtype), torch.from_numpy(g[f"{name}_k"]).to(dtype)
    up, down, p0, p1 = (int(v) for v in g[f"{name}_udp"])
    N, C, H, W = x.shape
    y = ops.upfirdn2d_raw(x.cuda().reshape(N * C, H, W, 1), k.cuda(), up, up, down, down, p0, p1, p0, p1)
    assert y.dtype == dtype
    y = y.reshape(N, C, y.shape[1], y.shape[2]).cpu()
    want = resample.upfirdn2d(x.double().numpy(), k.double().numpy(), up, up, down, down, p0, p1, p0, p1)
    assert tuple(y.shape) == want.shape == g[f"{name}_y"].shape
    if dtype == torch.float16:
        np.testing.assert_allclose(y.double().numpy(), want, atol=2.0 ** -11 * np.abs(want).max() + 1e-6)
        np.testing.assert_allclose(y.float().numpy(), g[f"{name}_y"], atol=4e-3 * np.abs(g[f"{name}_y"]).max())   # vs the fp32 golden
    else:
        np.testing.assert_allclose(y.numpy(), want, atol=1e-12)
    from inverseproblemwithdiffusionmodel_amd.op import upfirdn2d as op_upfirdn2d          # the reference-signature wrapper
    y2 = op_upfirdn2d(x.cuda(), k.cuda(), up=up, down=down, pad=(p0, p1))
    assert y2.dtype == dtype and torch.equal(y2.cpu(), y)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float64])
def test_fused_bias_act_half_and_double(ops, golden, dtype):
    """g10 (the reference's CPU fused_leaky_relu) cast to half / double + every (act, grad) code against the oracle"""
    g = golden("g10_biasact")
    tol = 2e-3 if dtype == torch.float16 else 2e-6          # (the oracle evaluates in float32)
    for xk, bk, yk in (("x", "b", "y_default"), ("x2", "b2", "y2")):
        x, b = torch.from_numpy(g[xk]).to(dtype), torch.from_numpy(g[bk]).to(dtype)
        y = ops.fused_bias_act_raw(x.cuda(), b.cuda(), None, 3, 0, 0.2, 2 ** 0.5)
        assert y.dtype == dtype
        want = resample.bias_act(x.double().numpy(), b.double().numpy(), None, 3, 0, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(y.cpu().double().numpy(), want, atol=tol * max(1.0, np.abs(want).max()))
        if dtype == torch.float16:
            np.testing.assert_allclose(y.cpu().float().numpy(), g[yk], atol=4e-3 * np.abs(g[yk]).max())
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((2, 6, 8, 8))).to(dtype)
    b = torch.from_numpy(rng.standard_normal(6)).to(dtype)
    ref = torch.from_numpy(rng.standard_normal(tuple(x.shape))).to(dtype)
    for act, grad in [(1, 0), (1, 1), (1, 2), (3, 0), (3, 1), (3, 2)]:
        want = resample.bias_act(x.double().numpy(), b.double().numpy(), ref.double().numpy(), act, grad, 0.3, 1.7)
        got = ops.fused_bias_act_raw(x.cuda(), b.cuda(), ref.cuda(), act, grad, 0.3, 1.7).cpu().double().numpy()
        np.testing.assert_allclose(got, want, atol=tol * max(1.0, np.abs(want).max()), err_msg=f"act={act} grad={grad}")
    with pytest.raises(TypeError):
        ops.fused_bias_act_raw(torch.zeros(2, 3, dtype=torch.bfloat16).cuda(), None, None, 3, 0, 0.2, 1.0)


# ---- k-space --------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", ["2x1x8x8", "1x2x7x9", "1x1x32x32", "1x1x6x5"])
def test_fft2c_golden(ops, golden, shape):
    g = golden("g03_fft")
    x = g[f"x_{shape}"]
    np.testing.assert_allclose(ops.fft2c(dev(x)).cpu().numpy(), g[f"i2k_{shape}"], atol=3e-6)
    np.testing.assert_allclose(ops.fft2c(dev(x), inverse=True).cpu().numpy(), g[f"k2i_{shape}"], atol=3e-6)


@pytest.mark.parametrize("H,W", [(128, 128), (64, 128), (16, 4), (256, 256), (96, 40)])
def test_fft2c_sizes(ops, H, W):
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((3, H, W)) + 1j * rng.standard_normal((3, H, W))).astype(np.complex64)
    k = ops.fft2c(dev(x))
    ref = kspace.fft2c(x)
    tol = 2e-5 * np.sqrt(H * W) / 32            # direct DFT path accumulates N terms
    np.testing.assert_allclose(k.cpu().numpy(), ref, atol=max(tol, 5e-6))
    np.testing.assert_allclose(ops.fft2c(k, inverse=True).cpu().numpy(), x, atol=max(tol, 5e-6))   # round trip
    # Parseval (orthonormal transform)
    assert abs(float((k.abs() ** 2).sum()) / float((np.abs(x) ** 2).sum()) - 1) < 1e-4


def _sense_setup(golden):
    g = golden("g04_sense")
    maps = kspace.sens_maps(4, 32, 32, 0)
    return g, maps, dev(maps.astype(np.float32)), dev(g["mask_T1"].reshape(1, 32).astype(np.uint8))


def test_sense_ops_golden(ops, golden):
    g, maps, sens, mask = _sense_setup(golden)
    y = ops.sense_forward(dev(g["x"]), sens, mask).cpu().numpy()
    np.testing.assert_allclose(y, g["Ax"], atol=5e-6)
    np.testing.assert_allclose(ops.sense_adjoint(dev(g["s"]), sens).cpu().numpy(), g["AHs"], atol=5e-6)
    np.testing.assert_allclose(ops.sense_ssos(dev(g["s"])).cpu().numpy(), g["ssos_s"], atol=5e-6)
    # the hard-wired T=24 mask of the live reference: image b uses mask row b
    m24 = dev(g["mask_T24"].reshape(24, 32).astype(np.uint8))
    np.testing.assert_allclose(ops.sense_forward(dev(g["x24"]), sens, m24).cpu().numpy(), g["Ax24"], atol=5e-6)


def test_sense_adjointness_128(ops):
    rng = np.random.default_rng(5)
    H = W = 128
    maps = kspace.sens_maps(4, H, W, 0).astype(np.float32)
    mask = kspace.generate_mask(1, W, seed=0, **kspace.MASK_PARAMS["R40"]).astype(np.uint8)
    x = (rng.standard_normal((2, 1, H, W)) + 1j * rng.standard_normal((2, 1, H, W))).astype(np.complex64)
    s = (rng.standard_normal((4, 2, 1, H, W)) + 1j * rng.standard_normal((4, 2, 1, H, W))).astype(np.complex64)
    Ax = ops.sense_forward(dev(x), dev(maps), dev(mask)).cpu().numpy().astype(np.complex128)
    AHs = ops.sense_adjoint(dev(s), dev(maps), dev(mask), apply_mask=True).cpu().numpy().astype(np.complex128)
    lhs, rhs = np.vdot(s.astype(np.complex128), Ax), np.vdot(AHs, x.astype(np.complex128))
    assert abs(lhs - rhs) < 1e-4 * abs(lhs)
    np.testing.assert_allclose(Ax, kspace.sense_forward(x, maps.astype(np.float64), mask[None]), atol=2e-5)


def test_l2prox_golden(ops, golden):
    g, maps, sens, mask = _sense_setup(golden)
    p = golden("g05_prox")
    z = p["z"]
    for i in range(3):
        alpha, lamda = p[f"l2_sense_{i}_alpha_lamda"]
        coef = 0.05 * (alpha / lamda) / (4 * 32)
        o_re, o_im = ops.sense_l2prox(dev(z.real), dev(z.imag), dev(p["y"]), sens, mask, coef)
        got = o_re.cpu().numpy() + 1j * o_im.cpu().numpy()
        np.testing.assert_allclose(got, p[f"l2_sense_{i}_x"], atol=3e-6)


def test_singlecoil_ops_golden(ops, golden):
    """RandomUndersamplingFourier __call__ / conj_op, its L2Penalty branch (K = B) and the SingleCoil closed form against
    the reference's own outputs (g05 sc_*), through the product classes and through the raw kernels"""
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import RandomUndersamplingFourier
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.proximal_op import get_proximal
    g4, p = golden("g04_sense"), golden("g05_prox")
    sc = RandomUndersamplingFourier(8, 0.04, (1, 32, 32), seed=2)
    assert np.array_equal(sc.mask.numpy(), p["sc_mask"])                                  # bit-exact mask
    x, z = dev(g4["x"]), dev(p["z"])
    y = sc(x)
    np.testing.assert_allclose(y.cpu().numpy(), p["sc_y"], atol=5e-6)
    np.testing.assert_allclose(sc.conj_op(dev(p["sc_y"])).cpu().numpy(), p["sc_Ax_adj"], atol=5e-6)
    got = get_proximal("L2Penalty")(sc)(z, dev(p["sc_y"]), 0.9, 1.0).cpu().numpy()
    np.testing.assert_allclose(got, p["l2_sc_x"], atol=3e-6)
    assert np.abs(p["l2_sc_x"] - p["z"]).max() > 1e-3                                    # the update is visible
    a, l = p["singlecoil_alpha_lamda"]
    prox = get_proximal("SingleCoil")(sc)
    xs = prox(z, dev(p["sc_y"]), float(a), float(l))
    np.testing.assert_allclose(xs.cpu().numpy(), p["singlecoil_x"], atol=5e-6)
    assert float(prox.check_solution(xs, z, dev(p["sc_y"]), float(a), float(l))) < 1e-8   # reference: p["singlecoil_check"]
    # raw kernel, planar, in place (out aliases z), vs the oracle
    m8 = dev(p["sc_mask"].reshape(1, 32).astype(np.uint8))
    zr, zi = dev(p["z"].real), dev(p["z"].imag)
    ops.singlecoil_prox(zr, zi, dev(p["sc_y"]), m8, float(a / l), ops.SC_CLOSED_FORM, out_re=zr, out_im=zi)
    want = kspace.single_coil(p["z"], p["sc_y"], a, l, p["sc_mask"])
    np.testing.assert_allclose(zr.cpu().numpy() + 1j * zi.cpu().numpy(), want, atol=5e-6)
    # projection (Constrained): F^-1[lamda S + (1-lamda) M F X + (1-M) F X] for an arbitrary S
    S = g4["s"][0]
    lam = 0.3
    k = kspace.fft2c(p["z"])
    m = p["sc_mask"]
    want = kspace.ifft2c(lam * S + (1 - lam) * m * k + (1 - m) * k)
    got = get_proximal("Constrained")(sc)(z, dev(S), lam).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=5e-6)
    with pytest.raises(RuntimeError):
        sc(torch.zeros(1, 1, 32, 32, dtype=torch.complex64))                             # no CPU path


def test_ald_singlecoil_step_matches_oracle(ops):
    """fused Langevin + single-coil data consistency at 128x128 (both proximals) vs the oracle; Philox path == injected"""
    rng = np.random.default_rng(16)
    H = W = 128
    B = 3
    mask = kspace.generate_mask(1, W, seed=0, **kspace.MASK_PARAMS["R8"])
    img = (rng.random((1, 1, H, W)) * np.exp(1j * rng.standard_normal((1, 1, H, W)))).astype(np.complex64)
    y = np.repeat((mask * kspace.fft2c(img)).astype(np.complex64), B, axis=0)
    x = (rng.standard_normal((B, 1, H, W)) + 1j * rng.standard_normal((B, 1, H, W))).astype(np.complex64)
    g = rng.standard_normal((2, B, 1, H, W)).astype(np.float32)
    n = rng.standard_normal((2, B, 1, H, W)).astype(np.float32)
    step, ns = np.float32(0.21), np.float32(np.sqrt(2 * 0.21))
    z = ((x.real + step * g[0] + n[0] * ns) + 1j * (x.imag + step * g[1] + n[1] * ns)).astype(np.complex64)
    m8 = dev(mask.astype(np.uint8))
    for mode, alpha, want in [(ops.SC_L2PENALTY, 30.0, kspace.l2_penalty_single(z, y, 30.0, 1.0, mask)),
                              (ops.SC_CLOSED_FORM, 0.7, kspace.single_coil(z, y, 0.7, 1.0, mask))]:
        coef = 0.05 * alpha / B if mode == ops.SC_L2PENALTY else alpha
        x_re, x_im = dev(x.real), dev(x.imag)
        ops.ald_singlecoil_step(x_re, x_im, dev(g[0]), dev(g[1]), dev(y), m8, mode, step=float(step),
                                noise_scale=float(ns), coef=coef, noise_re=dev(n[0]), noise_im=dev(n[1]))
        got = x_re.cpu().numpy() + 1j * x_im.cpu().numpy()
        assert np.abs(want - z).max() > 1e-2
        np.testing.assert_allclose(got, want, atol=1e-5)
    # Philox noise: the fused kernel draws what ipdm_philox_normal_f32 reports for (seed, sample, step, plane)
    x_re, x_im = dev(x.real), dev(x.imag)
    ops.ald_singlecoil_step(x_re, x_im, dev(g[0]), dev(g[1]), dev(y), m8, ops.SC_L2PENALTY, step=float(step),
                            noise_scale=float(ns), coef=0.0, seed=9, sample_offset=4, step_id=77)
    nr = ops.philox_normal((B, H * W), "cuda", seed=9, sample_offset=4, step_id=77, plane=0).cpu().numpy().reshape(B, 1, H, W)
    np.testing.assert_allclose(x_re.cpu().numpy(), x.real + step * g[0] + nr * ns, atol=2e-6)
    with pytest.raises(ValueError):
        ops.ald_singlecoil_step(x_re[:, :, :, ::2], x_im, dev(g[0]), dev(g[1]), dev(y), m8, 0)   # strided in-place operand


@pytest.mark.parametrize("H,W", [(256, 256), (128, 256), (512, 64)])
def test_large_image_kspace_ops_vs_oracle(ops, H, W):
    """images beyond one CU's LDS (the reference's real ACDC slices are 256x256, helpers/load_data.py:274): row / column
    pass FFT, SENSE forward / adjoint / SSOS, the L2Penalty proximal, the fused Langevin + proximal step (device schedule,
    injected and Philox noise) and the single-coil operators, all against the CPU oracle"""
    rng = np.random.default_rng(25)
    B, n = 3, 4
    x = (rng.standard_normal((B, 1, H, W)) + 1j * rng.standard_normal((B, 1, H, W))).astype(np.complex64)
    k = ops.fft2c(dev(x))
    np.testing.assert_allclose(k.cpu().numpy(), kspace.fft2c(x), atol=2e-5)
    np.testing.assert_allclose(ops.fft2c(k, inverse=True).cpu().numpy(), x, atol=2e-5)
    xin = dev(x)
    assert torch.equal(ops.fft2c(xin), k)                                  # same launch sequence, same bits
    maps = kspace.sens_maps(n, H, W, 0)
    mask = kspace.generate_mask(1, W, seed=0, **kspace.MASK_PARAMS["R8" if W < 128 else "R20"])
    sens, m8 = dev(maps.astype(np.float32)), dev(mask.astype(np.uint8))
    Ax = ops.sense_forward(dev(x), sens, m8).cpu().numpy()
    np.testing.assert_allclose(Ax, kspace.sense_forward(x, maps, mask[None]), atol=3e-5)
    s = (rng.standard_normal((n, B, 1, H, W)) + 1j * rng.standard_normal((n, B, 1, H, W))).astype(np.complex64)
    np.testing.assert_allclose(ops.sense_adjoint(dev(s), sens).cpu().numpy(), kspace.sense_adjoint(s, maps), atol=3e-5)
    np.testing.assert_allclose(ops.sense_ssos(dev(s)).cpu().numpy(), kspace.sense_ssos(s, maps), atol=3e-5)
    AHs = ops.sense_adjoint(dev(s), sens, m8, apply_mask=True).cpu().numpy().astype(np.complex128)
    lhs, rhs = np.vdot(s.astype(np.complex128), Ax.astype(np.complex128)), np.vdot(AHs, x.astype(np.complex128))
    assert abs(lhs - rhs) < 1e-4 * abs(lhs)                                # <s, A x> = <A^H s, x>
    # L2Penalty closed form and the fused step
    img = (rng.random((1, 1, H, W)) * np.exp(1j * rng.standard_normal((1, 1, H, W)))).astype(np.complex64)
    y = np.repeat(kspace.sense_forward(img, maps, mask[None]), B, axis=1)
    g = rng.standard_normal((2, B, 1, H, W)).astype(np.float32)
    nz = rng.standard_normal((2, B, 1, H, W)).astype(np.float32)
    step, ns, alpha = np.float32(0.37), np.float32(np.sqrt(2 * 0.37)), 60.0
    coef = 0.05 * alpha / (n * W)
    z = ((x.real + step * g[0] + nz[0] * ns) + 1j * (x.imag + step * g[1] + nz[1] * ns)).astype(np.complex64)
    want = kspace.l2_penalty_sense(z, y, alpha, 1.0, maps, mask[None])
    assert np.abs(want - z).max() > 1e-3
    o_re, o_im = ops.sense_l2prox(dev(z.real), dev(z.imag), dev(y), sens, m8, coef)
    np.testing.assert_allclose(o_re.cpu().numpy() + 1j * o_im.cpu().numpy(), want, atol=2e-5)
    work = ops.sense_workspace(B, n, H, W, "cuda")
    assert work.numel() * 4 == n * B * H * W * 8
    x_re, x_im = dev(x.real), dev(x.imag)
    sched = np.zeros(1, dtype=[("step", "f4"), ("ns", "f4"), ("coef", "f4"), ("sigma", "f4"), ("id", "i8"), ("seg", "f4"), ("rsv", "f4")])
    sched["step"], sched["ns"], sched["coef"], sched["id"] = step, ns, coef, 5
    ops.ald_sense_step(x_re, x_im, dev(g[0]), dev(g[1]), dev(y), sens, m8, work, noise_re=dev(nz[0]), noise_im=dev(nz[1]),
                       dev_sched=dev(sched.view(np.uint8)))
    np.testing.assert_allclose(x_re.cpu().numpy() + 1j * x_im.cpu().numpy(), want, atol=2e-5)
    # Philox noise, keyed as the 128x128 kernel keys it: (seed, sample_offset + b, step id, plane, quad)
    x_re, x_im = dev(x.real), dev(x.imag)
    ops.ald_sense_step(x_re, x_im, dev(g[0]), dev(g[1]), dev(y), sens, m8, work, step=float(step), noise_scale=float(ns),
                       coef=0.0, seed=9, sample_offset=4, step_id=77)
    n0 = ops.philox_normal((B, H * W), "cuda", seed=9, sample_offset=4, step_id=77, plane=0).cpu().numpy().reshape(B, 1, H, W)
    n1 = ops.philox_normal((B, H * W), "cuda", seed=9, sample_offset=4, step_id=77, plane=1).cpu().numpy().reshape(B, 1, H, W)
    np.testing.assert_allclose(x_re.cpu().numpy(), x.real + step * g[0] + n0 * ns, atol=3e-6)
    np.testing.assert_allclose(x_im.cpu().numpy(), x.imag + step * g[1] + n1 * ns, atol=3e-6)
    # single-coil operators
    ysc = np.repeat((mask * kspace.fft2c(img)).astype(np.complex64), B, axis=0)
    for mode, a, ref in [(ops.SC_L2PENALTY, 30.0, kspace.l2_penalty_single(z, ysc, 30.0, 1.0, mask)),
                         (ops.SC_CLOSED_FORM, 0.7, kspace.single_coil(z, ysc, 0.7, 1.0, mask))]:
        c = 0.05 * a / B if mode == ops.SC_L2PENALTY else a
        o_re, o_im = ops.singlecoil_prox(dev(z.real), dev(z.imag), dev(ysc), m8, c, mode)
        np.testing.assert_allclose(o_re.cpu().numpy() + 1j * o_im.cpu().numpy(), ref, atol=3e-5)
        x_re, x_im = dev(x.real), dev(x.imag)
        ops.ald_singlecoil_step(x_re, x_im, dev(g[0]), dev(g[1]), dev(ysc), m8, mode, step=float(step), noise_scale=float(ns),
                                coef=c, noise_re=dev(nz[0]), noise_im=dev(nz[1]))
        np.testing.assert_allclose(x_re.cpu().numpy() + 1j * x_im.cpu().numpy(), ref, atol=3e-5)
    lam = 0.3
    kz = kspace.fft2c(z)
    want = kspace.ifft2c(lam * ysc + (1 - lam) * mask * kz + (1 - mask) * kz)
    o_re, o_im = ops.singlecoil_prox(dev(z.real), dev(z.imag), dev(ysc), m8, lam, ops.SC_PROJECTION)
    np.testing.assert_allclose(o_re.cpu().numpy() + 1j * o_im.cpu().numpy(), want, atol=3e-5)


def test_sense_sampler_256(ops):
    """the SENSE sampler end to end at the reference's real ACDC size (256x256, 4 coils): tiny score net, 3 levels, graph
    and eager bit-equal, and equal to the CPU oracle under the same injected noise"""
    from argparse import Namespace
    from inverseproblemwithdiffusionmodel_amd import engine
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.ncsnv2 import NCSNv2Deepest
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    from oracle import scorenet as oracle_net, ald as oracle_ald, metrics
    devc = torch.device("cuda")
    H = W = 256
    cfg = engine.acdc_config(devc, H)
    cfg.model.ngf, cfg.model.num_classes, cfg.model.sigma_begin = 8, 12, 2.0
    cfg.recons.num_classes, cfg.recons.sigma_begin = 12, 2.0
    net = NCSNv2Deepest(cfg)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=1), strict=False)
    net = net.to(devc).eval()
    prob = engine.build_problem(devc, 2, R=20, H=H, W=W, scorenet=net, cfg=cfg, lr_scaled=2e6)
    gen = torch.Generator().manual_seed(5)
    tape = [torch.randn(2, 1, H, W, generator=gen) for _ in range(18)]
    runs = []
    for use_graph in (True, False):
        it = iter(tape)
        runs.append(prob.sampler(**prob.call_kwargs, noise_fn=lambda like: next(it), n_levels=3, use_graph=use_graph)[0])
    assert torch.equal(runs[0], runs[1])
    sd_cpu = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    it2 = iter(tape)
    with torch.no_grad():
        x_cpu = oracle_ald.ald_sense_real_imag(
            lambda x, lab: oracle_net.ncsnv2_deepest(x, lab, sd_cpu), prob.sigmas.cpu().numpy(),
            prob.measurement.cpu().numpy(), prob.op.sens_maps.numpy(), prob.op.random_under_fourier.mask.numpy(),
            cfg.sampling.step_lr, 3, 2e6, False, lambda like: next(it2), n_levels=3)
    x_gpu = runs[0].numpy()
    assert metrics.nrmse(np.abs(x_gpu), np.abs(x_cpu)) < 1e-3
    np.testing.assert_allclose(x_gpu, x_cpu, atol=1e-3 * np.abs(x_cpu).max())


def test_inplace_wrappers_reject_bad_operands(ops):
    """in-place kernels must not silently update a contiguous copy nor read a wrong dtype as raw memory"""
    x = torch.zeros(2, 8, 8, device="cuda")
    with pytest.raises(ValueError):
        ops.langevin_step(x.transpose(1, 2), torch.zeros(2, 8, 8, device="cuda"), 0.1, 0.1, noise=torch.zeros(2, 8, 8, device="cuda"))
    with pytest.raises(TypeError):
        ops.langevin_step(x, torch.zeros(2, 8, 8, device="cuda", dtype=torch.float64), 0.1, 0.1)
    with pytest.raises(ValueError):
        ops.langevin_step(x, torch.zeros(2, 8, 4, device="cuda"), 0.1, 0.1)
    with pytest.raises(TypeError):
        ops.adam_ascent(x, x.double(), x.clone(), x.clone(), 1e-3, 1)


def test_ald_sense_step_coil_parallel_is_bit_identical(ops, tmp_path):
    """the coil-parallel pair of kernels (one workgroup per (sample, coil) + combine) against the one-workgroup-per-sample
    kernel (IPDM_SENSE_COILS=0, read once per process -> a child process): same bits, Philox noise included"""
    import subprocess, sys, os
    code = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from inverseproblemwithdiffusionmodel_amd import ops
g = torch.Generator().manual_seed(77)
B, n, H, W = 3, 4, 128, 128
x = torch.randn(2, B, H, W, generator=g).cuda(); gr = torch.randn(2, B, H, W, generator=g).cuda()
y = torch.complex(torch.randn(n, B, H, W, generator=g), torch.randn(n, B, H, W, generator=g)).cuda()
sens = torch.randn(n, H, W, generator=g).cuda()
mask = (torch.rand(1, W, generator=g) < 0.3).to(torch.uint8).cuda()
work = ops.sense_workspace(B, n, H, W, "cuda")
ops.ald_sense_step(x[0], x[1], gr[0], gr[1], y, sens, mask, work, step=0.3, noise_scale=0.7, coef=0.011, seed=5,
                   sample_offset=9, step_id=1234)
torch.save(x.cpu(), sys.argv[2])
"""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, val in (("coils", "1"), ("serial", "0")):
        out = str(tmp_path / f"{tag}.pt")
        r = subprocess.run([sys.executable, "-c", code, repo, out], env=dict(os.environ, IPDM_SENSE_COILS=val),
                           capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(out))
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_ald_sense_step_matches_oracle(ops):
    rng = np.random.default_rng(6)
    H = W = 128
    B = 3
    maps = kspace.sens_maps(4, H, W, 0)
    mask = kspace.generate_mask(1, W, seed=0, **kspace.MASK_PARAMS["R20"])
    img = (rng.random((1, 1, H, W)) * np.exp(1j * rng.standard_normal((1, 1, H, W)))).astype(np.complex64)
    y = np.repeat(kspace.sense_forward(img, maps, mask[None]), B, axis=1)
    x = (rng.standard_normal((B, 1, H, W)) + 1j * rng.standard_normal((B, 1, H, W))).astype(np.complex64)
    g = rng.standard_normal((2, B, 1, H, W)).astype(np.float32)
    n = rng.standard_normal((2, B, 1, H, W)).astype(np.float32)
    step, ns, alpha = np.float32(0.37), np.float32(np.sqrt(2 * 0.37)), 40.0
    coef = 0.05 * alpha / (4 * W)
    z = ((x.real + step * g[0] + n[0] * ns) + 1j * (x.imag + step * g[1] + n[1] * ns)).astype(np.complex64)
    want = kspace.l2_penalty_sense(z, y, alpha, 1.0, maps, mask[None])
    x_re, x_im = dev(x.real), dev(x.imag)
    work = ops.sense_workspace(B, maps.shape[0], H, W, "cuda")
    ops.ald_sense_step(x_re, x_im, dev(g[0]), dev(g[1]), dev(y), dev(maps.astype(np.float32)),
                       dev(mask.astype(np.uint8)), work, step=float(step), noise_scale=float(ns), coef=coef,
                       noise_re=dev(n[0]), noise_im=dev(n[1]))
    got = x_re.cpu().numpy() + 1j * x_im.cpu().numpy()
    assert np.abs(want - z).max() > 1e-3                         # the data term is visible in this test
    np.testing.assert_allclose(got, want, atol=5e-6)
    # device-resident schedule struct gives the same result
    sched = np.zeros(1, dtype=[("step", "f4"), ("ns", "f4"), ("coef", "f4"), ("sigma", "f4"), ("id", "i8"), ("seg", "f4"), ("rsv", "f4")])
    sched["step"], sched["ns"], sched["coef"] = step, ns, coef
    x_re2, x_im2 = dev(x.real), dev(x.imag)
    ops.ald_sense_step(x_re2, x_im2, dev(g[0]), dev(g[1]), dev(y), dev(maps.astype(np.float32)),
                       dev(mask.astype(np.uint8)), work, noise_re=dev(n[0]), noise_im=dev(n[1]),
                       dev_sched=dev(sched.view(np.uint8)))
    assert torch.equal(x_re2, x_re) and torch.equal(x_im2, x_im)


def test_langevin_and_philox(ops):
    rng = np.random.default_rng(7)
    x = rng.standard_normal((5, 1, 33, 31)).astype(np.float32)
    g = rng.standard_normal(x.shape).astype(np.float32)
    n = rng.standard_normal(x.shape).astype(np.float32)
    step, ns = np.float32(0.011), np.float32(np.sqrt(0.022))
    xd = dev(x)
    ops.langevin_step(xd, dev(g), float(step), float(ns), noise=dev(n))
    np.testing.assert_allclose(xd.cpu().numpy(), x + step * g + n * ns, atol=1e-6)
    # Philox: same numbers regardless of how samples are sharded, standard-normal moments
    full = ops.philox_normal((8, 128 * 128), "cuda", seed=11, sample_offset=0, step_id=5)
    part = ops.philox_normal((3, 128 * 128), "cuda", seed=11, sample_offset=4, step_id=5)
    assert torch.equal(full[4:7], part)
    assert not torch.equal(full[0], ops.philox_normal((1, 128 * 128), "cuda", seed=11, step_id=6)[0])
    # seeds are independent streams: (seed 1, step s) never replays (seed 0, step s ^ 1) (XOR-folded key)
    s0 = torch.stack([ops.philox_normal((2, 4096), "cuda", seed=0, step_id=s) for s in range(8)])
    s1 = torch.stack([ops.philox_normal((2, 4096), "cuda", seed=1, step_id=s) for s in range(8)])
    assert all(not torch.equal(s0[a], s1[b]) for a in range(8) for b in range(8))
    assert float((s0[0] * s1[1]).mean().abs()) < 0.05 and float((s0[1] * s1[0]).mean().abs()) < 0.05
    assert abs(float(full.mean())) < 0.01 and abs(float(full.std()) - 1) < 0.01
    assert abs(float((full ** 4).mean()) - 3.0) < 0.1
    xd2 = dev(x.reshape(5, -1))
    ops.langevin_step(xd2, dev(g.reshape(5, -1)), float(step), float(ns), seed=3, sample_offset=2, step_id=9)
    nz = ops.philox_normal((5, x[0].size), "cuda", seed=3, sample_offset=2, step_id=9)
    np.testing.assert_allclose(xd2.cpu().numpy(), (x + step * g).reshape(5, -1) + nz.cpu().numpy() * ns, atol=1e-6)


# ---- score-network glue -----------------------------------------------------------------------
def test_instnorm_plus(ops, golden):
    g = golden("g07_layers")
    p = state_dict_from_golden(g, "in")
    x = dev(g["in_x"])
    coef = ops.instnorm_plus_coef(x, p["alpha"].cuda(), p["gamma"].cuda(), p["beta"].cuda())
    y = ops.affine_act(x, coef, ops.ACT_NONE).cpu().numpy()
    np.testing.assert_allclose(y, g["in_y"], atol=3e-6)
    y = ops.affine_act(x, coef, ops.ACT_ELU).cpu().numpy()
    np.testing.assert_allclose(y, F.elu(torch.from_numpy(g["in_y"])).numpy(), atol=3e-6)


@pytest.mark.parametrize("shape", [(2, 128, 128, 128), (3, 512, 16, 16), (1, 7, 9, 5), (2, 256, 32, 32), (5, 3, 4, 4), (2, 6, 8, 20)])
def test_instnorm_plus_sizes(ops, shape):
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(shape, generator=gen) * 3 + 50.0            # large mean: exercises the (x - mu) form
    C = shape[1]
    p = {"alpha": 1 + 0.1 * torch.randn(C, generator=gen), "gamma": 1 + 0.1 * torch.randn(C, generator=gen),
         "beta": 0.1 * torch.randn(C, generator=gen)}
    want = scorenet.instance_norm_plus(x, p)
    exact = scorenet.instance_norm_plus(x.double(), {k: v.double() for k, v in p.items()})
    coef = ops.instnorm_plus_coef(x.cuda(), p["alpha"].cuda(), p["gamma"].cuda(), p["beta"].cuda())
    got = ops.affine_act(x.cuda(), coef, ops.ACT_NONE).cpu()
    # plane means of 50 +- 0.02 make the cross-channel term ill-conditioned in fp32: judge both the
    # kernel and the fp32 CPU oracle against float64, the kernel must be at least as accurate
    err_gpu = (got.double() - exact).abs().max()
    err_cpu = (want.double() - exact).abs().max()
    assert err_gpu < max(2e-5, 2 * float(err_cpu)), (float(err_gpu), float(err_cpu))


def test_elementwise(ops):
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(3, 5, 17, 13, generator=gen) * 2
    y = torch.randn(3, 5, 17, 13, generator=gen)
    for name, fn in [("elu", F.elu), ("relu", F.relu), ("lrelu", lambda t: F.leaky_relu(t, 0.2)),
                     ("swish", lambda t: t * torch.sigmoid(t))]:
        got = ops.act(x.cuda(), ops.ACT_CODES[name]).cpu()
        assert (got - fn(x)).abs().max() < 1e-6, name
    assert torch.equal(ops.add(x.cuda(), y.cuda()).cpu(), x + y)
    assert torch.equal(ops.scale_shift(x.cuda(), 2.0, -1.0).cpu(), 2 * x - 1.0)
    sig = torch.tensor(kspace.get_sigmas(348, 0.01, 2311))
    labels = torch.tensor([0, 2310, 77])
    got = ops.div_sigma(x.cuda(), sig.cuda(), labels.cuda()).cpu()
    assert torch.equal(got, x / sig[labels].view(-1, 1, 1, 1))


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (1, 4, 128, 128), (2, 2, 37, 70), (1, 1, 3, 2)])
def test_maxpool5(ops, shape):
    x = torch.randn(shape, generator=torch.Generator().manual_seed(10))
    assert torch.equal(ops.maxpool5(x.cuda()).cpu(), F.max_pool2d(x, 5, 1, 2))


def test_meanpool2(ops):
    x = torch.randn(2, 5, 12, 10, generator=torch.Generator().manual_seed(11))
    assert torch.equal(ops.meanpool2(x.cuda()).cpu(), scorenet.mean_pool2(x))


@pytest.mark.parametrize("ish,osh", [((6, 5), (12, 10)), ((16, 16), (32, 32)), ((64, 64), (128, 128)),
                                     ((16, 16), (16, 16)), ((5, 7), (11, 9)), ((1, 1), (4, 4)),
                                     ((32, 32), (64, 64)), ((20, 24), (52, 60)), ((8, 128), (24, 256)), ((12, 16), (12, 32))])
def test_bilinear(ops, ish, osh):
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(2, 3, *ish, generator=gen)
    acc = torch.randn(2, 3, *osh, generator=gen)
    want = F.interpolate(x, size=osh, mode="bilinear", align_corners=True)
    got = ops.bilinear(x.cuda(), osh).cpu()
    assert (got - want).abs().max() < 2e-6
    out = acc.clone().cuda()
    ops.bilinear(x.cuda(), osh, out=out, accumulate=True)
    assert (out.cpu() - (acc + want)).abs().max() < 2e-6
    if ish == osh:
        assert torch.equal(got, x)                               # identity resize is exact


@pytest.mark.parametrize("ish,osh", [((3, 4, 5), (6, 8, 10)), ((8, 8, 12), (8, 8, 24)), ((4, 4, 6), (4, 4, 6)),
                                     ((2, 5, 7), (5, 11, 9)), ((1, 1, 1), (2, 3, 4)), ((8, 8, 24), (8, 8, 12))])
def test_trilinear(ops, ish, osh):
    """F.interpolate(mode='trilinear', align_corners=True) of the 3-D MSF block (layers3d.py:185,214), plain and accumulated"""
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(2, 3, *ish, generator=gen)
    acc = torch.randn(2, 3, *osh, generator=gen)
    want = F.interpolate(x.double(), size=osh, mode="trilinear", align_corners=True)
    got = ops.trilinear(x.cuda(), osh).cpu()
    assert (got.double() - want).abs().max() < 2e-5                  # fp32 source coordinates, as in ATen's fp32 path
    assert (got - F.interpolate(x, size=osh, mode="trilinear", align_corners=True)).abs().max() < 2e-6
    out = acc.clone().cuda()
    ops.trilinear(x.cuda(), osh, out=out, accumulate=True, act=ops.ACT_ELU)
    assert (out.cpu().double() - F.elu(acc.double() + want)).abs().max() < 2e-5
    if ish == osh:
        assert torch.equal(got, x)                               # identity resize is exact


def test_msf_block_3d_unequal_volumes(ops):
    """3-D MSF block on volumes of different sizes: sum_i trilinear(conv3d_i(x_i)) against torch"""
    from inverseproblemwithdiffusionmodel_amd.ncsn.models import layers3d
    gen = torch.Generator().manual_seed(14)
    msf = layers3d.MSFBlock([16, 16], 16).cuda()
    for p_ in msf.parameters():
        p_.data = (0.1 * torch.randn(p_.shape, generator=gen)).cuda()
    xs = [torch.randn(2, 16, 4, 4, 12, generator=gen), torch.randn(2, 16, 4, 4, 6, generator=gen)]
    shape = (4, 4, 12)
    with torch.no_grad():
        got = msf([v.cuda() for v in xs], shape).cpu().double()
        want = 0
        for conv, v in zip(msf.convs, xs):
            h = F.conv3d(v.double(), conv.weight.detach().cpu().double(), conv.bias.detach().cpu().double(), padding=1)
            want = want + F.interpolate(h, size=shape, mode="trilinear", align_corners=True)
    assert (got - want).abs().max() <= 2e-5 * want.abs().max()


@pytest.mark.parametrize("B,Cin,Cout,H,W,pool,res", [(2, 128, 128, 64, 64, False, True), (3, 32, 64, 40, 36, False, False),
                                                     (2, 64, 128, 64, 32, True, True), (1, 32, 64, 34, 44, True, False),
                                                     (2, 32, 64, 32, 32, False, True)])
@pytest.mark.parametrize("fmt", ["bx3", "hx2"])
def test_wino_bx3_statistics_epilogue(ops, B, Cin, Cout, H, W, pool, res, fmt):
    """the Winograd kernel's statistics epilogue: the InstanceNorm++ coefficients from its partials equal those from a pass
    over the stored tensor (ragged tile blocks, pooled epilogue, residual); the result itself is bit-identical with and
    without the epilogue"""
    from inverseproblemwithdiffusionmodel_amd import _lib
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(B, Cin, H, W, generator=gen).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.1).cuda()
    b = torch.randn(Cout, generator=gen).cuda()
    oh, ow = (H // 2, W // 2) if pool else (H, W)
    r = (torch.randn(B, Cout, oh, ow, generator=gen) * 3 + 5).cuda() if res else None      # a mean far from zero
    U = ops.conv_wino_bx3_weight(w, fmt=fmt)
    P = int(_lib.lib.ipdm_conv2d_wino_bx3_stats_partials(Cin, Cout, H, W, 1, int(pool)))
    assert P == 2 * ((W + 31) // 32) * ((H + 7) // 8)
    y0 = ops.conv2d_wino_bx3(x, U, b, r, pool2=pool)
    y1, y1a = ops.conv2d_wino_bx3(x, U, b, r, pool2=pool, act_out=ops.ACT_ELU, want_stats=True)
    assert torch.equal(y0, y1) and hasattr(y1, "_ipdm_partials")
    part, tag = y1._ipdm_partials                                        # (partials, (version, data_ptr) of the tensor)
    assert tuple(part.shape) == (B, Cout, P, 3) and tag == (y1._version, y1.data_ptr())
    assert float(part[..., 0].sum(dim=2).min()) == oh * ow == float(part[..., 0].sum(dim=2).max())
    alpha, gamma, beta = (torch.randn(Cout, generator=gen).cuda() for _ in range(3))
    c_part = ops.instnorm_plus_coef(y1, alpha, gamma, beta)
    assert not hasattr(y1, "_ipdm_partials")                                # single use
    y2 = ops.conv2d_wino_bx3(x, U, b, r, pool2=pool, want_stats=True)
    y2.mul_(2.0)                                                            # written in place: the partials are stale
    c_stale = ops.instnorm_plus_coef(y2, alpha, gamma, beta)                # ... and must be ignored
    assert (c_stale[..., 0] - 2.0 * c_part[..., 0]).abs().max() <= 1e-4 * c_part[..., 0].abs().max() + 1e-5
    c_full = ops.instnorm_plus_coef(y0, alpha, gamma, beta)                 # no partials on y0: reads the tensor
    assert (c_part - c_full).abs().max() <= 2e-5 * c_full.abs().max()
    yd = y0.double()
    mean = yd.mean(dim=(2, 3))
    rstd = 1.0 / torch.sqrt(yd.var(dim=(2, 3), unbiased=False) + 1e-5)
    assert (c_part[..., 0].double() - mean).abs().max() <= 1e-6 * mean.abs().max()
    assert ((c_part[..., 1] / gamma).double() / rstd - 1).abs().max() <= 1e-5


@pytest.mark.parametrize("B,Cin,Cout,H,W,res,act", [(3, 256, 256, 16, 16, True, True), (1, 64, 64, 16, 16, False, False),
                                                     (2, 512, 256, 16, 16, True, False), (2, 128, 128, 8, 16, False, True),
                                                     (5, 256, 128, 16, 12, True, True)])
@pytest.mark.parametrize("fmt", ["bx3", "hx2", "hx2-co32"])
def test_wino_bx3_split_k(ops, B, Cin, Cout, H, W, res, act, fmt, monkeypatch):
    """16-pixel layers with fewer than 512 output channels run as two K halves of the persistent Winograd kernel plus the
    fixed-order reduction -- or (f16x2 family, the default since round 4) as ONE launch of 32-channel workgroups: against a float64
    convolution, and bit-identical whatever the batch around a sample"""
    monkeypatch.setattr(ops, "WBX3_CO32", fmt == "hx2-co32")
    fmt = fmt.split("-")[0]
    gen = torch.Generator().manual_seed(43)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05
    b = torch.randn(Cout, generator=gen)
    r = torch.randn(B, Cout, H, W, generator=gen) if res else None
    assert ops.wino_bx3_splitk(Cin, Cout, H, W) == 2 and ops.wino_bx3_pays(Cin, Cout, H, W)
    U = ops.conv_wino_bx3_weight(w.cuda(), fmt=fmt)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res:
        want = want + r.double()
    got = ops.conv2d_wino_bx3(x.cuda(), U, b.cuda(), None if r is None else r.cuda(),
                              act_out=ops.ACT_ELU if act else ops.ACT_NONE)
    if act:
        got, got_act = got
        assert (got_act.cpu().double() - F.elu(want)).abs().max() <= 4e-6 * want.abs().max()
    assert (got.cpu().double() - want).abs().max() <= 4e-6 * want.abs().max()
    one = ops.conv2d_wino_bx3(x[B - 1:].cuda(), U, b.cuda(), None if r is None else r[B - 1:].cuda())
    assert torch.equal(one.cpu(), got.cpu()[B - 1:])


@pytest.mark.parametrize("B,Cin,Cout,dil,res", [(3, 64, 64, 2, True), (2, 256, 512, 2, False), (1, 32, 128, 4, True),
                                                 (5, 128, 64, 4, False)])
@pytest.mark.parametrize("fmt", ["bx3", "hx2"])
def test_wino_bx3_dilated_16px_polyphase(ops, B, Cin, Cout, dil, res, fmt):
    """dilated 16 x 16 layers: the persistent kernel on the polyphase tile space (whole padded image in LDS) against a float64
    convolution, and independent of the batch around a sample"""
    gen = torch.Generator().manual_seed(47)
    x = torch.randn(B, Cin, 16, 16, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05
    b = torch.randn(Cout, generator=gen)
    r = torch.randn(B, Cout, 16, 16, generator=gen) if res else None
    U = ops.conv_wino_bx3_weight(w.cuda(), fmt=fmt)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=dil, dilation=dil)
    if res:
        want = want + r.double()
    got, got_act = ops.conv2d_wino_bx3(x.cuda(), U, b.cuda(), None if r is None else r.cuda(), act_out=ops.ACT_ELU,
                                       dilation=dil)
    assert (got.cpu().double() - want).abs().max() <= 4e-6 * want.abs().max()
    assert (got_act.cpu().double() - F.elu(want)).abs().max() <= 4e-6 * want.abs().max()
    one = ops.conv2d_wino_bx3(x[B - 1:].cuda(), U, b.cuda(), None if r is None else r[B - 1:].cuda(), dilation=dil)
    assert torch.equal(one.cpu(), got.cpu()[B - 1:])


def test_wino_bx3_split_k_rule_is_shape_only(ops):
    assert ops.wino_bx3_splitk(256, 512, 16, 16) == 1          # enough channel tiles: plain launch
    assert ops.wino_bx3_splitk(256, 256, 32, 32) == 1          # wide image
    assert ops.wino_bx3_splitk(256, 256, 16, 16, 2) == 1       # dilated
    assert ops.wino_bx3_splitk(32, 64, 16, 16) == 1            # a K half would be a single chunk
    assert ops.wino_bx3_splitk(256, 256, 16, 16) == 2


@pytest.mark.parametrize("fmt", ["bx3", "hx2"])
def test_wino_bx3_statistics_epilogue_large_mean(ops, fmt):
    """planes with |mean| >> spread (1000 +- 1): the epilogue's one-pass sums are taken about a shift from the data, so the
    variance keeps the precision the two-pass kernel has"""
    gen = torch.Generator().manual_seed(42)
    B, Cin, Cout, H, W = 2, 32, 64, 40, 36
    x = torch.randn(B, Cin, H, W, generator=gen).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05).cuda()
    r = (torch.randn(B, Cout, H, W, generator=gen) + 1000.0).cuda()
    U = ops.conv_wino_bx3_weight(w, fmt=fmt)
    y = ops.conv2d_wino_bx3(x, U, None, r, want_stats=True)
    assert hasattr(y, "_ipdm_partials")
    ones = torch.ones(Cout).cuda()
    c_part = ops.instnorm_plus_coef(y, ones, ones, None)
    yd = y.double()
    rstd = 1.0 / torch.sqrt(yd.var(dim=(2, 3), unbiased=False) + 1e-5)
    assert (c_part[..., 0].double() - yd.mean(dim=(2, 3))).abs().max() < 1e-3            # 1000 +- 6e-5 ulp data
    assert (c_part[..., 1].double() / rstd - 1).abs().max() < 2e-4
    c_full = ops.instnorm_plus_coef(y, ones, ones, None)                                   # partials consumed: full pass
    assert (c_full[..., 1].double() / rstd - 1).abs().max() < 2e-4


@pytest.mark.parametrize("fmt", ["bx3", "hx2"])
def test_wino_bx3_statistics_epilogue_unsupported(ops, fmt):
    from inverseproblemwithdiffusionmodel_amd import _lib
    lib = _lib.lib
    assert lib.ipdm_conv2d_wino_bx3_stats_partials(256, 256, 16, 16, 1, 0) == 0        # small-image kernels
    assert lib.ipdm_conv2d_wino_bx3_stats_partials(256, 256, 32, 32, 2, 0) == 0        # dilated
    assert lib.ipdm_conv2d_wino_bx3_stats_partials(16, 64, 64, 64, 1, 0) == 0          # one chunk: not the persistent kernel
    x = torch.randn(1, 256, 16, 16).cuda()
    U = ops.conv_wino_bx3_weight(torch.randn(512, 256, 3, 3).cuda(), fmt=fmt)
    y = ops.conv2d_wino_bx3(x, U, want_stats=True)                                     # silently without partials
    assert not hasattr(y, "_ipdm_partials")


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(3, 1, 128, 32, 32), (2, 2, 20, 12, 16), (2, 3, 128, 16, 32), (1, 1, 5, 1, 4),
                                            (3, 128, 1, 32, 32), (2, 24, 2, 12, 16), (2, 128, 3, 16, 32), (1, 7, 1, 1, 4),
                                            (2, 1, 1, 8, 8), (2, 3, 3, 9, 12), (2, 16, 1, 6, 512), (1, 16, 2, 5, 24)])
def test_conv3x3_thin(ops, B, Cin, Cout, H, W):
    """first / last layer streaming kernels (begin_conv / end_conv, ncsnv2.py:40,45) against a float64 convolution; the
    matrix-core path must agree to rounding"""
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.2
    b = torch.randn(Cout, generator=gen)
    assert ops.conv3x3_thin_ok(Cin, Cout, H, W)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    got = ops.conv3x3_thin(x.cuda(), w.cuda(), b.cuda()).cpu().double()
    assert (got - want).abs().max() <= 2e-6 * max(1.0, want.abs().max())
    got_nb = ops.conv3x3_thin(x.cuda(), w.cuda(), None).cpu().double()
    assert (got_nb - F.conv2d(x.double(), w.double(), None, padding=1)).abs().max() <= 2e-6 * max(1.0, want.abs().max())
    mfma = ops.conv2d(x.cuda(), ops.conv_weight(w.cuda()), b.cuda()).cpu().double()
    assert (mfma - got).abs().max() <= 4e-6 * max(1.0, want.abs().max())
    if Cin <= 3:                                              # fused input affine: padding stays zero
        coef = torch.randn(B, Cin, 3, generator=gen)
        xa = (x - coef[:, :, 0, None, None]) * coef[:, :, 1, None, None] + coef[:, :, 2, None, None]
        want_a = F.conv2d(xa.double(), w.double(), b.double(), padding=1)
        got_a = ops.conv3x3_thin(x.cuda(), w.cuda(), b.cuda(), coef.cuda()).cpu().double()
        assert (got_a - want_a).abs().max() <= 4e-6 * max(1.0, want_a.abs().max())
    # a sample's result does not depend on the batch around it
    if B > 1:
        one = ops.conv3x3_thin(x[1:2].cuda(), w.cuda(), b.cuda()).cpu()
        assert torch.equal(one, ops.conv3x3_thin(x.cuda(), w.cuda(), b.cuda()).cpu()[1:2])


def test_conv3x3_thin_unsupported(ops):
    from inverseproblemwithdiffusionmodel_amd import _lib
    assert not ops.conv3x3_thin_ok(16, 16, 32, 32) and not ops.conv3x3_thin_ok(1, 16, 32, 30)
    x = torch.randn(1, 16, 8, 8).cuda()
    w = torch.randn(16, 16, 3, 3).cuda()
    with pytest.raises(_lib.IpdmUnsupported):
        ops.conv3x3_thin(x, w)


# ---- MFMA convolution ---------------------------------------------------------------------------
CONV_CASES = [
    # B, Cin, Cout, H, W, k, dil, fused-norm, act, residual
    (2, 16, 32, 32, 32, 3, 1, False, "none", False),
    (2, 128, 128, 64, 64, 3, 1, True, "elu", True),          # big-tile config
    (1, 128, 256, 32, 32, 3, 1, False, "elu", False),
    (3, 256, 256, 16, 16, 3, 1, True, "elu", True),          # 16x16 stage
    (2, 256, 512, 16, 16, 3, 2, True, "elu", False),         # dilation 2
    (2, 512, 512, 16, 16, 3, 4, False, "elu", True),         # dilation 4
    (2, 1, 128, 32, 32, 3, 1, False, "none", False),         # begin_conv: Cin = 1
    (2, 128, 1, 32, 32, 3, 1, True, "elu", False),           # end_conv: Cout = 1
    (2, 128, 256, 32, 32, 1, 1, False, "none", False),       # 1x1 shortcut
    (1, 6, 5, 12, 10, 3, 1, True, "elu", True),              # ragged everything
    (1, 7, 9, 19, 45, 3, 1, False, "relu", False),
    (2, 24, 40, 8, 8, 3, 2, False, "none", False),
]


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,dil,norm,actname,res", CONV_CASES)
def test_conv2d(ops, B, Cin, Cout, H, W, k, dil, norm, actname, res):
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=gen)
    resid = torch.randn(B, Cout, H, W, generator=gen) if res else None
    p = {"alpha": 1 + 0.1 * torch.randn(Cin, generator=gen), "gamma": 1 + 0.1 * torch.randn(Cin, generator=gen),
         "beta": 0.1 * torch.randn(Cin, generator=gen)}
    fn = {"none": lambda t: t, "elu": F.elu, "relu": F.relu}[actname]
    h = scorenet.instance_norm_plus(x.double(), {a: b.double() for a, b in p.items()}) if norm else x.double()
    want = F.conv2d(fn(h), w.double(), bias.double(), padding=(k // 2) * dil, dilation=dil)
    if res:
        want = want + resid.double()
    xd = x.cuda()
    coef = ops.instnorm_plus_coef(xd, p["alpha"].cuda(), p["gamma"].cuda(), p["beta"].cuda()) if norm else None
    wt = ops.conv_pack_weight(w.cuda())
    assert torch.equal(wt.cpu(), w.permute(2, 3, 1, 0).reshape(k * k, Cin, Cout))
    got = ops.conv2d(xd, wt, bias.cuda(), coef, ops.ACT_CODES[actname], None if resid is None else resid.cuda(), dil)
    # fp32 fma chain over K = Cin*k*k products of O(1) terms normalised to unit variance
    err = (got.cpu().double() - want).abs().max()
    assert err < 2e-5 * max(1.0, float(want.abs().max())), float(err)


def test_conv2d_mfma_layout_identity(ops):
    """A = identity-like weights with an asymmetric input catches a transposed accumulator map."""
    Cin = Cout = 64
    x = torch.arange(2 * Cin * 8 * 32, dtype=torch.float32).reshape(2, Cin, 8, 32) % 251
    w = torch.zeros(Cout, Cin, 3, 3)
    for c in range(Cout):
        w[c, (c * 7 + 3) % Cin, 1, 1] = 1.0          # a permutation of channels, centre tap only
    got = ops.conv2d(x.cuda(), ops.conv_pack_weight(w.cuda())).cpu()
    want = x[:, [(c * 7 + 3) % Cin for c in range(Cout)]]
    assert torch.equal(got, want)


def test_conv2d_activated_second_output(ops):
    """epilogue emits act(result) next to (or instead of) the raw result"""
    gen = torch.Generator().manual_seed(14)
    x = torch.randn(2, 64, 16, 32, generator=gen)
    w = torch.randn(64, 64, 3, 3, generator=gen) / 24
    resid = torch.randn(2, 64, 16, 32, generator=gen)
    wt = ops.conv_pack_weight(w.cuda())
    want = F.conv2d(x.double(), w.double(), padding=1) + resid.double()
    raw, act = ops.conv2d(x.cuda(), wt, residual=resid.cuda(), act_out=ops.ACT_ELU)
    assert (raw.cpu().double() - want).abs().max() < 2e-5
    assert (act.cpu().double() - F.elu(want)).abs().max() < 2e-5
    none, act2 = ops.conv2d(x.cuda(), wt, residual=resid.cuda(), act_out=ops.ACT_ELU, raw=False)
    assert none is None and torch.equal(act2, act)
    # ELU accuracy of the branch-free form around zero (series) and for large negatives (exp)
    v = torch.tensor([-20.0, -3.0, -0.07, -0.0625, -0.06, -1e-3, -1e-6, 0.0, 1e-6, 2.0])
    xin = torch.zeros(1, 32, 1, 32)
    xin[0, 0, 0, :10] = v
    wid = torch.zeros(32, 32, 1, 1)
    wid[0, 0, 0, 0] = 1.0
    _, e = ops.conv2d(xin.cuda(), ops.conv_pack_weight(wid.cuda()), act_out=ops.ACT_ELU)
    got = e.cpu()[0, 0, 0, :10].double()
    ref = F.elu(v.double())
    assert ((got - ref).abs() <= 2e-7 + 2e-6 * ref.abs()).all()


@pytest.mark.parametrize("B,Cin,Cout,H,W,res,dil", [
    (2, 64, 64, 32, 32, True, 1), (1, 128, 128, 64, 64, False, 1), (3, 256, 256, 32, 32, True, 1),
    (2, 8, 64, 10, 36, False, 1), (1, 128, 256, 128, 128, False, 1),
    (3, 64, 64, 16, 16, True, 1), (2, 64, 128, 16, 16, True, 2), (2, 128, 64, 16, 16, False, 4),   # small / dilated
    (1, 16, 64, 24, 16, False, 2), (1, 16, 64, 32, 32, True, 2)])
def test_conv2d_winograd(ops, B, Cin, Cout, H, W, res, dil):
    """F(2x2,3x3) path vs a float64 direct convolution; also checks the weight transform U = G g G^T"""
    assert ops.conv_wino_supported(Cin, Cout, H, W, dil)
    assert not ops.conv_wino_supported(Cin + 1, Cout, H, W, dil) and not ops.conv_wino_supported(Cin, Cout, 18, 16, 4)
    gen = torch.Generator().manual_seed(15)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=gen)
    resid = torch.randn(B, Cout, H, W, generator=gen) if res else None
    want = F.conv2d(x.double(), w.double(), bias.double(), padding=dil, dilation=dil)
    if res:
        want = want + resid.double()
    U = ops.conv_wino_weight(w.cuda())
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    Uref = torch.einsum("ai,ocij,bj->aboc", G, w.double(), G).reshape(16, Cout, Cin).permute(0, 2, 1)
    assert (U.cpu().double() - Uref).abs().max() < 1e-6
    raw, act = ops.conv2d_wino(x.cuda(), U, bias.cuda(), None if resid is None else resid.cuda(), act_out=ops.ACT_ELU,
                               dilation=dil)
    err = (raw.cpu().double() - want).abs().max()
    assert err < 4e-5 * max(1.0, float(want.abs().max())), float(err)
    assert (act.cpu().double() - F.elu(want)).abs().max() < 4e-5 * max(1.0, float(want.abs().max()))


# ---- split-bf16 ("bf16x3") convolution: the same cases, held to a TIGHTER float64-referenced bound -------------
@pytest.mark.parametrize("B,Cin,Cout,H,W,k,dil,norm,actname,res", CONV_CASES + [
    (3, 32, 96, 40, 70, 3, 1, False, "none", True),          # several pixel tiles, ragged edges, 3 co-tiles of 32
    (2, 48, 64, 16, 16, 3, 4, False, "none", False),         # PW = 16 tiles with dilation 4
    (2, 20, 33, 9, 16, 1, 1, True, "elu", True),             # 1x1, ragged channels, fused norm + activation
])
@pytest.mark.parametrize("fmt", ["bx3", "hx2"])
def test_conv_bx3(ops, B, Cin, Cout, H, W, k, dil, norm, actname, res, fmt):
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, k, k, generator=gen) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=gen)
    resid = torch.randn(B, Cout, H, W, generator=gen) if res else None
    p = {"alpha": 1 + 0.1 * torch.randn(Cin, generator=gen), "gamma": 1 + 0.1 * torch.randn(Cin, generator=gen),
         "beta": 0.1 * torch.randn(Cin, generator=gen)}
    fn = {"none": lambda t: t, "elu": F.elu, "relu": F.relu}[actname]
    h = scorenet.instance_norm_plus(x.double(), {a: b.double() for a, b in p.items()}) if norm else x.double()
    want = F.conv2d(fn(h), w.double(), bias.double(), padding=(k // 2) * dil, dilation=dil)
    if res:
        want = want + resid.double()
    xd = x.cuda()
    coef = ops.instnorm_plus_coef(xd, p["alpha"].cuda(), p["gamma"].cuda(), p["beta"].cuda()) if norm else None
    wq = ops.conv_bx3_weight(w.cuda(), fmt=fmt)
    got = ops.conv_bx3(xd, wq, bias.cuda(), coef, ops.ACT_CODES[actname], None if resid is None else resid.cuda(), dil)
    err = (got.cpu().double() - want).abs().max()
    # six-term split: the dropped products are < 2^-26 of |w x|, i.e. below the rounding of the fp32 accumulation
    # itself; the fused norm/activation prologue (fp32) dominates when present
    tol = (2e-5 if norm or actname != "none" else 4e-6) * max(1.0, float(want.abs().max()))
    assert err < tol, float(err)
    # the modules' dispatching entry (ops.conv2d on a PackedBx3) is the same launch
    got2 = ops.conv2d(xd, wq, bias.cuda(), coef, ops.ACT_CODES[actname], None if resid is None else resid.cuda(), dil)
    assert torch.equal(got, got2)


def test_conv_bx3_exactness_and_layout(ops):
    """a channel permutation with full-significand inputs must come back BIT-exact: the three bf16 pieces carry all
    24 significand bits and the accumulator map is not transposed"""
    Cin = Cout = 64
    gen = torch.Generator().manual_seed(16)
    x = (torch.randn(2, Cin, 8, 32, generator=gen) * 1000.0)
    x[0, 0, 0, :4] = torch.tensor([1.0 + 2.0 ** -23, -(2.0 - 2.0 ** -22), 3.0e-30, 16777215.0])
    w = torch.zeros(Cout, Cin, 3, 3)
    for c in range(Cout):
        w[c, (c * 7 + 3) % Cin, 1, 1] = 1.0
    got = ops.conv_bx3(x.cuda(), ops.conv_bx3_weight(w.cuda())).cpu()
    assert torch.equal(got, x[:, [(c * 7 + 3) % Cin for c in range(Cout)]])
    # weights with full significands too: one tap, exact power-of-two input
    w2 = torch.zeros(32, 16, 1, 1)
    vals = torch.randn(32, generator=gen)
    w2[torch.arange(32), torch.arange(32) % 16, 0, 0] = vals
    x2 = torch.zeros(1, 16, 4, 32)
    x2[0, :, :, :] = 0.5
    got2 = ops.conv_bx3(x2.cuda(), ops.conv_bx3_weight(w2.cuda())).cpu()
    assert torch.equal(got2[0, :, 0, 0], vals * 0.5)


def test_conv_hx2_exactness_and_layout(ops):
    """the f16x2 twin of the test above: values with 22 significant bits inside fp16's exponent window are carried EXACTLY by
    the two fp16 pieces (11 + 11 bits), so a channel permutation comes back bit-exact; weights with full fp32 significands
    against power-of-two inputs lose at most the third piece (2^-24 relative: one fp32 rounding)"""
    Cin = Cout = 64
    gen = torch.Generator().manual_seed(16)
    x = torch.randint(-2 ** 21, 2 ** 21, (2, Cin, 8, 32), generator=gen).float() * 2.0 ** -12
    x[0, 0, 0, :4] = torch.tensor([1.0 + 2.0 ** -21, -(2.0 - 2.0 ** -20), 511.999755859375, -0.000244140625])
    w = torch.zeros(Cout, Cin, 3, 3)
    for c in range(Cout):
        w[c, (c * 7 + 3) % Cin, 1, 1] = 1.0
    got = ops.conv_bx3(x.cuda(), ops.conv_hx2_weight(w.cuda())).cpu()
    assert torch.equal(got, x[:, [(c * 7 + 3) % Cin for c in range(Cout)]])
    U = ops.conv_wino_hx2_weight(w.cuda())               # the Winograd form: its 4-term input sums exceed 22 bits -> 2^-22
    assert (ops.conv2d_wino_bx3(x.cuda(), U).cpu() - got).abs().max() <= 2.0 ** -21 * got.abs().max()
    w2 = torch.zeros(32, 16, 1, 1)
    vals = torch.randn(32, generator=gen)
    w2[torch.arange(32), torch.arange(32) % 16, 0, 0] = vals
    x2 = torch.full((1, 16, 4, 32), 0.5)
    got2 = ops.conv_bx3(x2.cuda(), ops.conv_hx2_weight(w2.cuda())).cpu()
    assert (got2[0, :, 0, 0] - vals * 0.5).abs().max() <= 2.0 ** -24 * (vals * 0.5).abs().max()
    # the range contract: an activation beyond fp16's range gives a non-finite result, never a wrong finite one
    x3 = x.clone()
    x3[1, 5, 2, 3] = 1.0e5
    got3 = ops.conv_bx3(x3.cuda(), ops.conv_hx2_weight(w.cuda())).cpu()
    assert not torch.isfinite(got3[1, [c for c in range(Cout) if (c * 7 + 3) % Cin == 5][0], 2, 3])
    assert torch.isfinite(got3[0]).all()


@pytest.mark.parametrize("scale", [1.0e5, 3.0e7, 1.0e-6, 1.0])
def test_conv_hx2_dynamic_range(ops, scale):
    """f16x2 with `in_amax`: every image is scaled into fp16's range by an exact power of two (ipdm_absmax_f32 ->
    hx_dynamic_scale) and unscaled in the epilogue, so inputs far beyond 65504 -- or far below fp16's normal range -- give the
    same float64-referenced accuracy as O(1) inputs; images of one batch may differ by many orders of magnitude.  Direct,
    Winograd (wide, small / dilated, split-K) and 3-D forms."""
    gen = torch.Generator().manual_seed(23)
    for (B, Cin, Cout, H, W, dil, wino) in [(3, 64, 64, 32, 32, 1, True), (2, 32, 64, 16, 16, 2, True), (3, 256, 256, 16, 16, 1, True),
                                            (2, 32, 48, 20, 24, 1, False), (2, 128, 64, 40, 36, 1, True)]:
        x = torch.randn(B, Cin, H, W, generator=gen)
        x[0] *= scale
        if B > 2:
            x[2] *= 1.0e-3                                   # a third image many orders below the first
        w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (9 * Cin) ** 0.5
        b = torch.randn(Cout, generator=gen)
        want = F.conv2d(x.double(), w.double(), None, padding=dil, dilation=dil)
        if wino:
            assert ops.conv_wino_bx3_supported(Cin, Cout, H, W, dil)
            got = ops.conv2d_wino_bx3(x.cuda(), ops.conv_wino_hx2_weight(w.cuda()), None, dilation=dil, in_amax=True)
        else:
            got = ops.conv_bx3(x.cuda(), ops.conv_hx2_weight(w.cuda()), None, dilation=dil, in_amax=True)
        got = got.cpu().double()
        assert torch.isfinite(got).all()
        for i in range(B):                                   # per image: error relative to THAT image's output range
            assert (got[i] - want[i]).abs().max() <= 1e-5 * want[i].abs().max(), (scale, i, Cin, H)
    am = ops.amax_value(ops.absmax_per_image(x.cuda())).cpu()
    assert torch.equal(am, x.abs().amax(dim=(1, 2, 3)))
    x3 = torch.randn(2, 16, 6, 8, 24, generator=gen) * scale
    w3 = torch.randn(32, 16, 3, 3, 3, generator=gen) / (27 * 16) ** 0.5
    want3 = F.conv3d(x3.double(), w3.double(), padding=1)
    got3 = ops.conv_bx3(x3.cuda(), ops.conv_hx2_weight(w3.cuda()), in_amax=True).cpu().double()
    assert (got3 - want3).abs().max() <= 1e-5 * want3.abs().max()
    if scale > 65504:                                        # ... and without in_amax the static contract answers with inf / NaN
        bad = ops.conv_bx3(x3.cuda(), ops.conv_hx2_weight(w3.cuda())).cpu()
        assert not torch.isfinite(bad).all()


@pytest.mark.parametrize("fmt", ["bx3", "hx2"])
def test_conv_bx3_activated_second_output(ops, fmt):
    gen = torch.Generator().manual_seed(14)
    x = torch.randn(2, 64, 16, 32, generator=gen)
    w = torch.randn(64, 64, 3, 3, generator=gen) / 24
    resid = torch.randn(2, 64, 16, 32, generator=gen)
    wq = ops.conv_bx3_weight(w.cuda(), fmt=fmt)
    want = F.conv2d(x.double(), w.double(), padding=1) + resid.double()
    raw, act = ops.conv_bx3(x.cuda(), wq, residual=resid.cuda(), act_out=ops.ACT_ELU)
    assert (raw.cpu().double() - want).abs().max() < 4e-6
    assert (act.cpu().double() - F.elu(want)).abs().max() < 4e-6
    none, act2 = ops.conv_bx3(x.cuda(), wq, residual=resid.cuda(), act_out=ops.ACT_ELU, raw=False)
    assert none is None and torch.equal(act2, act)


@pytest.mark.parametrize("dil,Cin,Cout,D,Hh,Ww", [(1, 8, 16, 8, 8, 24), (2, 16, 32, 8, 8, 12), (4, 32, 32, 8, 8, 24),
                                                  (1, 1, 8, 5, 6, 7), (2, 8, 1, 4, 8, 24), (1, 128, 128, 8, 8, 24)])
@pytest.mark.parametrize("fmt", ["bx3", "hx2"])
def test_conv_bx3_3d(ops, dil, Cin, Cout, D, Hh, Ww, fmt):
    gen = torch.Generator().manual_seed(20)
    x = torch.randn(3, Cin, D, Hh, Ww, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (Cin * 27) ** 0.5
    b = torch.randn(Cout, generator=gen)
    r = torch.randn(3, Cout, D, Hh, Ww, generator=gen)
    want = F.conv3d(x.double(), w.double(), b.double(), padding=dil, dilation=dil) + r.double()
    got = ops.conv3d(x.cuda(), ops.conv_bx3_weight(w.cuda(), fmt=fmt), b.cuda(), residual=r.cuda(), dilation=dil)
    assert (got.cpu().double() - want).abs().max() < 4e-6 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("B,Cin,Cout,H,W,res,dil", [
    (2, 64, 64, 32, 32, True, 1), (1, 128, 128, 64, 64, False, 1), (3, 256, 256, 32, 32, True, 1),
    (2, 16, 64, 10, 36, False, 1), (1, 128, 256, 128, 128, False, 1), (2, 32, 64, 40, 70, True, 1),   # ragged edges
    (2, 32, 64, 40, 36, True, 1), (1, 48, 128, 72, 96, True, 1),                                  # ... on the 16-byte DMA form
    (3, 64, 64, 16, 16, True, 1), (2, 64, 128, 16, 16, True, 2), (2, 128, 64, 16, 16, False, 4),      # small / dilated
    (1, 16, 64, 24, 16, False, 2), (1, 16, 64, 32, 32, True, 2),
    (40, 32, 256, 16, 16, True, 1), (80, 32, 128, 12, 16, False, 1)])        # 16-pixel images, persistent DMA path
@pytest.mark.parametrize("fmt", ["bx3", "hx2"])
def test_conv2d_winograd_bx3(ops, B, Cin, Cout, H, W, res, dil, fmt):
    """split-bf16 F(2x2,3x3) path (LDS-DMA raw stage on wide images, register path on small / dilated ones) vs a
    float64 direct convolution"""
    assert ops.conv_wino_bx3_supported(Cin, Cout, H, W, dil)
    assert not ops.conv_wino_bx3_supported(Cin + 8, Cout, H, W, dil) and not ops.conv_wino_bx3_supported(Cin, Cout + 32, H, W, dil)
    gen = torch.Generator().manual_seed(15)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=gen)
    resid = torch.randn(B, Cout, H, W, generator=gen) if res else None
    want = F.conv2d(x.double(), w.double(), bias.double(), padding=dil, dilation=dil)
    if res:
        want = want + resid.double()
    U = ops.conv_wino_bx3_weight(w.cuda(), fmt=fmt)
    raw, act = ops.conv2d_wino_bx3(x.cuda(), U, bias.cuda(), None if resid is None else resid.cuda(), act_out=ops.ACT_ELU,
                                   dilation=dil)
    scale = max(1.0, float(want.abs().max()))
    err = (raw.cpu().double() - want).abs().max()
    assert err < 1e-5 * scale, float(err)          # Winograd transforms in fp32: a few ulp above the direct kernels
    assert (act.cpu().double() - F.elu(want)).abs().max() < 1e-5 * scale
    # repeated launches are bit-identical (no race between the DMA refill, the transform and the MFMA reads)
    raw2 = ops.conv2d_wino_bx3(x.cuda(), U, bias.cuda(), None if resid is None else resid.cuda(), dilation=dil)
    assert torch.equal(raw, raw2)


# ---- TV baseline -----------------------------------------------------------------------------------------------------
def test_tv_value_and_gradient_vs_autograd(ops):
    """ipdm_tv_c64 / ipdm_tv_grad_c64 against the published definition (oracle/tv.py) and torch autograd of it on the CPU
    (complex parameter: the gradient torch.optim.Adam would see), including zero differences (flat patches -> 0)"""
    from oracle import tv as otv
    gen = torch.Generator().manual_seed(5)
    x = torch.complex(torch.randn(3, 1, 20, 28, generator=gen), torch.randn(3, 1, 20, 28, generator=gen))
    x[0, 0, 4:9, 5:11] = 0.7 - 0.2j                                   # flat patch: zero differences inside
    xp = x.clone().requires_grad_(True)
    val = otv.total_variation(xp)
    val.sum().backward()
    got_v = ops.tv_value(x.cuda()).cpu().reshape(3, 1)
    np.testing.assert_allclose(got_v.numpy(), val.detach().double().numpy(), rtol=1e-6)
    got_g = ops.tv_grad(x.cuda()).cpu()
    np.testing.assert_allclose(torch.view_as_real(got_g).numpy(), torch.view_as_real(xp.grad).numpy(), atol=2e-6)


def test_tv_map_model_vs_autograd_adam(ops):
    """MAPModel.fit (HIP: closed-form data gradient + TV gradient kernel + Adam kernel) against the same 25 epochs of
    torch autograd + torch.optim.Adam on the CPU oracle operators"""
    from oracle import tv as otv
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import SENSE
    from inverseproblemwithdiffusionmodel_amd.ncsn.models.MAP_optimizers import MAPModel, TotalVariation
    from inverseproblemwithdiffusionmodel_amd.synthetic import phantom_image
    H = W = 32
    op = SENSE("exp", 4, 8, 0.05, (1, H, W), seed=0)
    # texture everywhere: in FLAT regions the TV gradient is the unit vector of a rounding-sized difference -- an arbitrary
    # direction that Adam turns into a full step, so two fp32 evaluations legitimately part ways there (seen on the bare
    # phantom: the zero-filled start agrees to 6e-8 and one epoch later 2 lr apart in the background)
    gen = torch.Generator().manual_seed(11)
    img = phantom_image(H, W, seed=1) + 0.2 * torch.complex(torch.randn(1, 1, H, W, generator=gen), torch.randn(1, 1, H, W, generator=gen))
    meas = op(img.cuda())
    model = MAPModel(meas, op, TotalVariation(), 0.01)
    x = model.fit(25, 1e-2).numpy()
    maps = torch.from_numpy(op.sens_maps.numpy()).to(torch.complex64)          # (4, H, W)
    mask = op.random_under_fourier.mask.reshape(-1, 1, 1, W)[0].to(torch.float32)

    def fft2c(t):
        return torch.fft.fftshift(torch.fft.fft2(torch.fft.ifftshift(t, dim=(-1, -2)), norm="ortho"), dim=(-1, -2))

    def ifft2c(t):
        return torch.fft.fftshift(torch.fft.ifft2(torch.fft.ifftshift(t, dim=(-1, -2)), norm="ortho"), dim=(-1, -2))

    fwd = lambda X: mask * fft2c(maps[:, None, None] * X[None])
    adj = lambda S: (torch.conj(maps[:, None, None]) * ifft2c(S)).sum(0)
    m_cpu = meas.cpu()
    assert torch.allclose(fwd(img), m_cpu, atol=1e-5)                          # the oracle operators are the op's
    want = otv.tv_map(m_cpu, fwd, adj, 0.01, 1e-2, 25).numpy()
    # Adam turns a gradient of ANY size into a step of ~lr, so a component whose gradient is at rounding level may step the
    # other way in fp32 (the same caveat as the MAP golden test): bound the outliers by a couple of steps and require the
    # bulk to agree to rounding
    diff = np.abs(x - want)
    stats = (float(diff.max()), float(np.sqrt((diff ** 2).mean())), float((diff > 1e-4 * np.abs(want).max()).mean()))
    assert diff.max() <= 2.5 * 1e-2 and np.sqrt((diff ** 2).mean()) < 2e-3 * np.abs(want).max(), stats
    assert (diff > 1e-4 * np.abs(want).max()).mean() < 0.05, stats
    assert np.abs(want - adj(m_cpu).numpy()).max() > 0.05 * np.abs(want).max()   # 25 epochs moved the image


# ---- on-device reporting ---------------------------------------------------------------------------------------------
def test_device_metrics_match_host_definitions(ops):
    """NRMSE / SSIM / posterior mean-std on the GPU against the numpy definitions of helpers/metrics.py (the reference's
    argument order: the reconstruction normalises NRMSE) and against a direct per-window float64 evaluation of SSIM"""
    from inverseproblemwithdiffusionmodel_amd.helpers import metrics as hm
    from oracle import metrics as om
    rng = np.random.default_rng(31)
    B, H, W = 5, 48, 40
    ref = (rng.random((1, 1, H, W)) * np.exp(1j * rng.standard_normal((1, 1, H, W)))).astype(np.complex64)
    rec = (ref + 0.05 * (rng.standard_normal((B, 1, H, W)) + 1j * rng.standard_normal((B, 1, H, W)))).astype(np.complex64)
    got = hm.compute_metrics_device(["NRMSE", "SSIM", "MAE"], dev(rec), dev(ref))
    want = hm.compute_metrics(["NRMSE", "SSIM", "MAE"], np.abs(rec), np.abs(ref))
    for k in ("NRMSE", "SSIM", "MAE"):
        np.testing.assert_allclose(got[k], want[k], rtol=1e-6, atol=1e-9, err_msg=k)
    assert abs(float(got["NRMSE"][0]) - om.nrmse(np.abs(rec[0]), np.abs(ref[0]))) < 1e-9
    # direct window evaluation of one SSIM map entry and of the mean (independent of scipy's uniform_filter)
    a, b = np.abs(rec[1, 0]).astype(np.float64), np.abs(ref[0, 0]).astype(np.float64)
    tot = 0.0
    for oy in range(H - 6):
        for ox in range(W - 6):
            wa, wb = a[oy:oy + 7, ox:ox + 7], b[oy:oy + 7, ox:ox + 7]
            ux, uy = wa.mean(), wb.mean()
            vx, vy = wa.var(ddof=1), wb.var(ddof=1)
            vxy = ((wa - ux) * (wb - uy)).sum() / 48.0
            tot += ((2 * ux * uy + 0.02 ** 2) * (2 * vxy + 0.06 ** 2)) / ((ux ** 2 + uy ** 2 + 0.02 ** 2) * (vx + vy + 0.06 ** 2))
    assert abs(float(got["SSIM"][1]) - tot / ((H - 6) * (W - 6))) < 1e-8     # one-pass float64 window moments
    # per-image references and the reduce= path
    refs = np.repeat(ref, B, axis=0) * (1 + 0.1 * np.arange(B))[:, None, None, None].astype(np.float32)
    got2 = hm.compute_metrics_device(["NRMSE", "SSIM"], dev(rec), dev(refs.astype(np.complex64)), reduce="mean")
    want2 = hm.compute_metrics(["NRMSE", "SSIM"], np.abs(rec), np.abs(refs), reduce="mean")
    for k in ("NRMSE", "SSIM"):
        assert abs(got2[k] - want2[k]) < 1e-6
    # posterior panels
    mm, pm, ms, ps = (t.cpu().numpy() for t in hm.compute_mean_and_std_device(dev(rec)))
    wm, wp, ws_, wps = hm.compute_mean_and_std(rec)
    np.testing.assert_allclose(mm, wm, atol=1e-6)
    np.testing.assert_allclose(pm, wp, atol=2e-6)
    np.testing.assert_allclose(ms, ws_, atol=2e-5)
    np.testing.assert_allclose(ps, wps, atol=2e-5)      # float64 moment planes
    np.testing.assert_allclose(ops.magnitude(dev(rec)).cpu().numpy(), np.abs(rec), rtol=1e-6)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(3, 128, 256, 64, 64), (2, 32, 64, 32, 96), (2, 64, 128, 40, 36)])
@pytest.mark.parametrize("fmt", ["bx3", "hx2"])
def test_conv_wino_bx3_pooled_epilogue(ops, B, Cin, Cout, H, W, fmt):
    """ConvMeanPool in one launch (the Winograd output tile is the 2x2 pooling window): conv + bias -> 2x2 mean -> + pooled
    residual -> (ELU copy), against float64 torch, same 4e-6 bound as the unpooled kernel"""
    gen = torch.Generator().manual_seed(41)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=gen)
    r = torch.randn(B, Cout, H // 2, W // 2, generator=gen)
    want = torch.nn.functional.avg_pool2d(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1), 2) + r.double()
    U = ops.conv_wino_bx3_weight(w.cuda(), fmt=fmt)
    out, out_act = ops.conv2d_wino_bx3(x.cuda(), U, b.cuda(), r.cuda(), act_out=ops.ACT_ELU, pool2=True)
    assert out.shape == (B, Cout, H // 2, W // 2)
    scale = float(want.abs().max())
    assert float((out.cpu().double() - want).abs().max()) < 4e-6 * scale
    assert float((out_act.cpu().double() - torch.nn.functional.elu(want)).abs().max()) < 4e-6 * scale
    # and exactly the unfused chain's value up to the summation order of the four window elements
    full = ops.conv2d_wino_bx3(x.cuda(), U, b.cuda())
    chain = ops.add(ops.meanpool2(full), r.cuda())
    assert float((chain - out).abs().max()) < 2e-6 * scale
    only_act = ops.conv2d_wino_bx3(x.cuda(), U, b.cuda(), r.cuda(), act_out=ops.ACT_ELU, raw=False, pool2=True)
    assert only_act[0] is None and torch.equal(only_act[1], out_act)
    with pytest.raises(Exception):
        ops.conv2d_wino_bx3(x.cuda()[:, :, :16, :16].contiguous(), U, b.cuda(), pool2=True)     # small-image kernels: unsupported


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["hx2", "bx3"])
@pytest.mark.parametrize("dil", [1, 2])
def test_conv_bx3_channel_tile_choice_keeps_the_bits(ops, fmt, dil):
    """16-pixel layers: the direct split-operand kernel picks 128 output channels per workgroup when the launch is large
    enough and 64 otherwise (conv_bx3_dispatch) -- a sample's result must not depend on how many samples share the launch
    (sharding relies on it): 512 images at once (128-channel tiles) == the same images 8 at a time (64-channel tiles),
    2-D and 3-D (depth folded into K), bit for bit"""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(512, 32, 8, 12, generator=g).cuda()
    w = (torch.randn(160, 32, 3, 3, generator=g) * 0.1).cuda()          # 160: a ragged second 128-channel tile
    bias = torch.randn(160, generator=g).cuda()
    pk = ops.conv_bx3_weight(w, fmt=fmt)
    big = ops.conv_bx3(x, pk, bias, dilation=dil)
    small = torch.cat([ops.conv_bx3(x[i:i + 8].contiguous(), pk, bias, dilation=dil) for i in range(0, 512, 8)])
    assert torch.equal(big, small)
    ref = torch.nn.functional.conv2d(x[:4].double(), w.double(), bias.double(), padding=dil, dilation=dil)
    assert (big[:4].double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    if dil == 1:
        xv = torch.randn(64, 16, 8, 8, 12, generator=g).cuda()          # 64 volumes x 8 slices: 128-channel tiles
        wv = (torch.randn(128, 16, 3, 3, 3, generator=g) * 0.1).cuda()
        pv = ops.conv_bx3_weight(wv, fmt=fmt)
        bigv = ops.conv3d(xv, pv)
        smallv = torch.cat([ops.conv3d(xv[i:i + 2].contiguous(), pv) for i in range(0, 64, 2)])
        assert torch.equal(bigv, smallv)


def test_undersampling_fourier_and_uniform_mask_operators(ops, golden):
    """UndersamplingFourier (reference undersampling_fourier.py:10-36) against the reference class's outputs, its adjointness, and
    RandomUndersamplingFourier / SENSE built on the legacy uniform mask against the numpy oracle"""
    from inverseproblemwithdiffusionmodel_amd.ncsn.linear_transforms.undersampling_fourier import (
        UndersamplingFourier, RandomUndersamplingFourier, SENSE)
    from oracle import kspace
    g = golden("g30_legacy_operators")
    x = torch.from_numpy(g["x"]).cuda()
    for n in (2, 3):
        op = UndersamplingFourier(n, (1, 12, 10))
        y = op(x)
        np.testing.assert_allclose(y.cpu().numpy(), g[f"uf{n}_y"], atol=2e-6)
        np.testing.assert_allclose(op.conj_op(y).cpu().numpy(), g[f"uf{n}_adj"], atol=2e-6)
        gen = torch.Generator().manual_seed(n)
        s = torch.complex(torch.randn(y.shape, generator=gen), torch.randn(y.shape, generator=gen)).cuda()
        lhs, rhs = torch.vdot(op(x).flatten(), s.flatten()), torch.vdot(x.flatten(), op.conj_op(s).flatten())
        assert abs(lhs - rhs) <= 1e-5 * abs(lhs)
    gen = torch.Generator().manual_seed(9)
    img = torch.complex(torch.randn(3, 1, 64, 64, generator=gen), torch.randn(3, 1, 64, 64, generator=gen))
    op = RandomUndersamplingFourier(6, 0.08, (1, 64, 64), seed=4, mask_mode="uniform")
    m = op.mask.numpy().astype(bool)
    want = kspace.fft2c(img.numpy()) * m
    np.testing.assert_allclose(op(img.cuda()).cpu().numpy(), want, atol=3e-5)
    sense = SENSE("exp", 4, 6, 0.08, (1, 64, 64), seed=4, mask_mode="uniform")
    assert torch.equal(sense.random_under_fourier.mask, op.mask)
    y = sense(img.cuda())
    want = kspace.sense_forward(img.numpy(), sense.sens_maps.numpy(), m)
    np.testing.assert_allclose(y.cpu().numpy(), want, atol=3e-5)


@pytest.mark.parametrize("fmt", ["hx2", "bx3"])
def test_residual_into_the_second_output_only(ops, fmt):
    """res_second: out = conv + bias, out_act = act_out(conv + bias + residual) from ONE launch -- CRPBlock's
    `path = conv(pool(path)); x = path + x` (reference ncsn/models/layers.py:76-83) without the separate add; every epilogue
    that a CRP convolution can meet (wide / small / split-K Winograd, direct 2-D and 3-D), bits equal to the two-launch form"""
    gen = torch.Generator().manual_seed(61)
    for (B, Cin, Cout, H, W, kind) in [(2, 64, 64, 32, 64, "wino"), (3, 64, 64, 16, 16, "wino"), (3, 256, 256, 16, 16, "wino"),
                                       (2, 32, 48, 20, 24, "direct"), (1, 256, 64, 8, 16, "direct")]:
        x = torch.randn(B, Cin, H, W, generator=gen).cuda()
        w = (torch.randn(Cout, Cin, 3, 3, generator=gen) / (9 * Cin) ** 0.5).cuda()
        res = torch.randn(B, Cout, H, W, generator=gen).cuda()
        if kind == "wino":
            U = ops.conv_wino_bx3_weight(w, fmt=fmt)
            path, xs = ops.conv2d_wino_bx3(x, U, None, res, act_out=ops.ACT_COPY, res_second=True, want_amax=True)
            path2 = ops.conv2d_wino_bx3(x, U)
            both = ops.conv2d_wino_bx3(x, U, None, res)
            _, act = ops.conv2d_wino_bx3(x, U, None, res, act_out=ops.ACT_ELU, res_second=True)
        else:
            wq = ops.conv_bx3_weight(w, fmt=fmt)
            path, xs = ops.conv_bx3(x, wq, residual=res, act_out=ops.ACT_COPY, res_second=True, want_amax=True)
            path2 = ops.conv_bx3(x, wq)
            both = ops.conv_bx3(x, wq, residual=res)
            _, act = ops.conv_bx3(x, wq, residual=res, act_out=ops.ACT_ELU, res_second=True)
        assert torch.equal(path, path2) and torch.equal(xs, both), (kind, Cin, H)
        assert torch.equal(xs, ops.add(path, res))                      # == the separate add kernel
        assert (act - F.elu(both)).abs().max() < 4e-6
        assert torch.equal(ops.amax_value(ops.amax_of(path)), path.abs().amax(dim=(1, 2, 3)))
        assert torch.equal(ops.amax_value(ops.amax_of(xs)), xs.abs().amax(dim=(1, 2, 3)))
    x3 = torch.randn(2, 16, 4, 8, 12, generator=gen).cuda()
    w3 = (torch.randn(32, 16, 3, 3, 3, generator=gen) / (27 * 16) ** 0.5).cuda()
    r3 = torch.randn(2, 32, 4, 8, 12, generator=gen).cuda()
    wq3 = ops.conv_bx3_weight(w3, fmt=fmt)
    p3, s3 = ops.conv3d(x3, wq3, residual=r3, act_out=ops.ACT_COPY, res_second=True)
    assert torch.equal(p3, ops.conv3d(x3, wq3)) and torch.equal(s3, ops.conv3d(x3, wq3, residual=r3))
    with pytest.raises(ValueError):
        ops.conv_bx3(x3, wq3, act_out=ops.ACT_COPY, res_second=True)      # no residual


@pytest.mark.parametrize("fmt", ["hx2", "bx3"])
@pytest.mark.parametrize("B,Cin,Cout,D,Hh,Ww", [(2, 128, 128, 8, 8, 12), (1, 64, 128, 5, 8, 12), (1, 128, 256, 3, 6, 16),
                                                  (1, 256, 256, 2, 8, 12), (2, 32, 128, 7, 11, 9)])
def test_conv3d_two_slices_per_workgroup(ops, fmt, B, Cin, Cout, D, Hh, Ww):
    """undilated 3x3x3 layers on slices <= 16 pixels wide with >= 128 output channels run the two-slice form of the direct kernel
    (conv_bx3.hip, ZT = 2: a workgroup owns the same pixel tile of two consecutive depth slices -- half the weight traffic per
    output).  Even and odd depths, ragged planes, bias / residual / activated copy / maxima / residual-into-second-output, against
    a float64 convolution; IPDM_BX3_ZT2=0 (the one-slice form) must agree to rounding."""
    gen = torch.Generator().manual_seed(20 + D)
    x = torch.randn(B, Cin, D, Hh, Ww, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (Cin * 27) ** 0.5
    b = torch.randn(Cout, generator=gen)
    r = torch.randn(B, Cout, D, Hh, Ww, generator=gen)
    want = F.conv3d(x.double(), w.double(), b.double(), padding=1) + r.double()
    wq = ops.conv_bx3_weight(w.cuda(), fmt=fmt)
    xg, rg = x.cuda(), r.cuda()
    am = ops.absmax_per_image(xg) if fmt == "hx2" else None
    out, act = ops.conv3d(xg, wq, b.cuda(), residual=rg, act_out=ops.ACT_ELU, in_amax=am, want_amax=True)
    tol = 4e-6 * max(1.0, float(want.abs().max()))
    assert (out.cpu().double() - want).abs().max() < tol
    assert (act.cpu().double() - F.elu(want)).abs().max() < tol
    assert torch.equal(ops.amax_value(ops.amax_of(out)), out.abs().amax(dim=(1, 2, 3, 4)))
    assert torch.equal(ops.amax_value(ops.amax_of(act)), act.abs().amax(dim=(1, 2, 3, 4)))
    path, summed = ops.conv3d(xg, wq, b.cuda(), residual=rg, act_out=ops.ACT_COPY, in_amax=am, res_second=True)
    assert torch.equal(summed, out) and (path.cpu().double() - (want - r.double())).abs().max() < tol
    plain = ops.conv3d(xg, wq, in_amax=am)
    assert (plain.cpu().double() - F.conv3d(x.double(), w.double(), padding=1)).abs().max() < tol
    # a sample's bits do not depend on the batch it is computed in (the form is a rule of the layer shape)
    one = ops.conv3d(xg[:1].contiguous(), wq, b.cuda(), residual=rg[:1].contiguous(),
                     in_amax=None if am is None else am[:1].contiguous())
    assert torch.equal(one[0], out[0])
