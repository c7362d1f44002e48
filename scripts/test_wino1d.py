#!/usr/bin/env python3
"""1-D Winograd kernel (conv_wino1d.hip) against float64 and against the 2-D f16x2 kernel: error and time per launch.
   python scripts/test_wino1d.py [--bench]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from inverseproblemwithdiffusionmodel_amd import _lib, ops  # noqa: E402

L = ctypes.CDLL(_lib.LIB_PATH)
L.ipdm_conv_wino1d_weight_bytes.restype = ctypes.c_int64
L.ipdm_conv_wino1d_weight_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
L.ipdm_conv_wino1d_pack_weight.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L.ipdm_conv2d_wino1d_f32.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p]


def pack(w):
    Cout, Cin = w.shape[:2]
    n = L.ipdm_conv_wino1d_weight_bytes(Cout, Cin)
    blob = torch.empty(n, dtype=torch.uint8, device=w.device)
    rc = L.ipdm_conv_wino1d_pack_weight(w.data_ptr(), blob.data_ptr(), Cout, Cin, None)
    assert rc == 0, rc
    return blob


def p(t):
    return None if t is None else t.data_ptr()


def conv1d(x, blob, Cout, bias=None, residual=None, act_out=0, raw=True):
    B, Cin, H, W = x.shape
    out = torch.empty((B, Cout, H, W), device=x.device) if raw else None
    oa = torch.empty((B, Cout, H, W), device=x.device) if act_out else None
    rc = L.ipdm_conv2d_wino1d_f32(p(x), p(blob), p(bias), p(residual), p(out), p(oa), act_out, B, Cin, Cout, H, W, None,
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return out, oa


def check(B, Cin, Cout, H, W, res=True, act=True):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, Cin, H, W, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (3 * Cin ** 0.5)
    bias = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(B, Cout, H, W, device="cuda", generator=g) if res else None
    blob = pack(w)
    out, oa = conv1d(x, blob, Cout, bias, r, ops.ACT_ELU if act else 0)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1)
    if res:
        ref = ref + r.double()
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    msg = f"B={B} {Cin}->{Cout} @{H}x{W}: max err / range = {err:.2e}"
    if act:
        ea = (oa.double() - torch.nn.functional.elu(ref)).abs().max().item() / ref.abs().max().item()
        msg += f", act {ea:.2e}"
    # the production 2-D kernel on the same input
    U = ops.conv_wino_hx2_weight(w)
    if U is not None:
        o2 = ops.conv2d_wino_bx3(x, U, bias, r)
        e2 = (o2.double() - ref).abs().max().item() / ref.abs().max().item()
        msg += f"   (2-D kernel {e2:.2e})"
    print(msg, flush=True)
    return err


def bench(B, Cin, Cout, H, W, n=20):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, Cin, H, W, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (3 * Cin ** 0.5)
    bias = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(B, Cout, H, W, device="cuda", generator=g)
    blob = pack(w)
    U = ops.conv_wino_hx2_weight(w)

    def t(f):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t1 = t(lambda: conv1d(x, blob, Cout, bias, r, ops.ACT_ELU))
    t2 = t(lambda: ops.conv2d_wino_bx3(x, U, bias, r, act_out=ops.ACT_ELU))
    fl = 2 * B * Cout * Cin * 9 * H * W
    print(f"B={B} {Cin}->{Cout} @{H}x{W}: 1-D {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF)   2-D {t2:7.1f} us ({fl / t2 / 1e6:6.1f} TF)   ratio {t1 / t2:.3f}",
          flush=True)


if __name__ == "__main__":
    check(1, 32, 128, 8, 32)
    check(2, 64, 128, 16, 64)
    check(3, 128, 256, 40, 36)
    check(2, 128, 128, 128, 128)
    if "--bench1" in sys.argv:
        bench(28, 128, 128, 128, 128)
        bench(28, 256, 256, 64, 64)
    if "--bench" in sys.argv:
        bench(28, 128, 128, 128, 128)
        bench(28, 256, 256, 64, 64)
        bench(28, 256, 128, 128, 128)
        bench(28, 512, 256, 64, 64)
        bench(28, 256, 256, 32, 32)
