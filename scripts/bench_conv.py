#!/usr/bin/env python3
"""Micro-benchmark: the hand-written fp32 MFMA convolution vs MIOpen (torch.nn.functional.conv2d) on
the layer shapes of NCSNv2Deepest at 128x128 (SURVEY.md 3.5 conv census).  GPU only."""
import sys
import os
import time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops

B = int(os.environ.get("BENCH_B", 28))
ONLY = os.environ.get("BENCH_ONLY")          # e.g. "0,1,2": indices into SHAPES
NO_MIOPEN = os.environ.get("BENCH_NO_MIOPEN") == "1"
WINO = os.environ.get("BENCH_WINO") == "1"     # Winograd F(2x2,3x3) path where eligible
BX3 = os.environ.get("BENCH_BX3") == "1"       # split-operand kernels (fp32-faithful on the 16-bit matrix cores)
FMT = os.environ.get("BENCH_FMT", "hx2")       # ... which family: hx2 (two fp16 pieces) or bx3 (three bf16 pieces)
FUSED = os.environ.get("BENCH_FUSED") == "1"   # ELU + InstanceNorm++ coefficients on the input, residual on the output
SHAPES = [  # (count per forward, Cin, Cout, H, dil)
    (18, 128, 128, 128, 1), (9, 128, 128, 64, 1), (8, 256, 256, 64, 1), (17, 256, 256, 32, 1),
    (25, 256, 256, 16, 1), (16, 512, 512, 16, 2), (5, 512, 512, 16, 4), (1, 128, 256, 128, 1),
    (1, 256, 512, 16, 2), (1, 1, 128, 128, 1), (1, 128, 1, 128, 1), (16, 512, 512, 16, 1),
]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot_m = tot_o = 0.0
print(f"B={B}  (fp32 MFMA peak 157.3 TFLOP/s)")
for idx, (cnt, ci, co, hw, dil) in enumerate(SHAPES):
    if ONLY and str(idx) not in ONLY.split(","):
        continue
    x = torch.randn(B, ci, hw, hw, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    bias = torch.randn(co, device="cuda")
    wt = ops.conv_pack_weight(w)
    out = torch.empty(B, co, hw, hw, device="cuda")
    flop = 2.0 * B * hw * hw * ci * co * 9
    t_m = 1e9 if NO_MIOPEN else timeit(lambda: F.conv2d(x, w, bias, padding=dil, dilation=dil))
    if BX3 and WINO and ops.conv_wino_bx3_supported(ci, co, hw, hw, dil):
        one_d = FMT == "hx2" and os.environ.get("BENCH_W1D", "1") == "1" and dil == 1 and ops.wino1d_pays(ci, co, hw, hw)
        Uq = ops.conv_wino1d_weight(w) if one_d else ops.conv_wino_bx3_weight(w, fmt=FMT)    # the layer dispatch's choice
        t_o = timeit(lambda: ops.conv2d_wino_bx3(x, Uq, bias, dilation=dil))
        out = ops.conv2d_wino_bx3(x, Uq, bias, dilation=dil)
        ref64 = F.conv2d(x[:2].double(), w.double(), bias.double(), padding=dil, dilation=dil)
        e_w = ((ops.conv2d_wino_bx3(x[:2].contiguous(), Uq, bias, dilation=dil).double() - ref64).abs().max() / ref64.abs().max()).item()
        e_f = ((ops.conv2d_wino(x[:2].contiguous(), ops.conv_wino_weight(w), bias, dilation=dil).double() - ref64).abs().max() / ref64.abs().max()).item() if ops.conv_wino_supported(ci, co, hw, hw, dil) else float("nan")
        print(f"      rel-to-max error vs float64: wino-bx3 {e_w:.2e}   wino-fp32 {e_f:.2e}")
    elif BX3:
        wq = ops.conv_bx3_weight(w, fmt=FMT)
        t_o = timeit(lambda: ops.conv_bx3(x, wq, bias, dilation=dil, out=out))
        x64, w64 = x[:2].double(), w.double()
        ref64 = F.conv2d(x64, w64, bias.double(), padding=dil, dilation=dil)
        e_bx3 = ((ops.conv_bx3(x[:2].contiguous(), wq, bias, dilation=dil).double() - ref64).abs().max() / ref64.abs().max()).item()
        e_f32 = ((ops.conv2d(x[:2].contiguous(), wt, bias, dilation=dil).double() - ref64).abs().max() / ref64.abs().max()).item()
        print(f"      rel-to-max error vs float64: bx3 {e_bx3:.2e}   fp32 MFMA {e_f32:.2e}")
    elif WINO and ops.conv_wino_supported(ci, co, hw, hw, dil):
        U = ops.conv_wino_weight(w)
        t_o = timeit(lambda: ops.conv2d_wino(x, U, bias, dilation=dil))
        out = ops.conv2d_wino(x, U, bias, dilation=dil)
    elif FUSED and ci > 1:
        coef = torch.randn(B, ci, 3, device="cuda")
        res = torch.randn(B, co, hw, hw, device="cuda")
        t_o = timeit(lambda: ops.conv2d(x, wt, bias, coef, ops.ACT_ELU, res, dilation=dil, out=out))
        ops.conv2d(x, wt, bias, dilation=dil, out=out)
    else:
        t_o = timeit(lambda: ops.conv2d(x, wt, bias, dilation=dil, out=out))
    err = (out - F.conv2d(x, w, bias, padding=dil, dilation=dil)).abs().max().item()
    tot_m += cnt * t_m; tot_o += cnt * t_o
    print(f"{cnt:3d}x {ci:4d}->{co:4d} @{hw:3d}^2 d{dil}: miopen {t_m:8.3f} ms {flop / t_m / 1e9:7.1f} TF | "
          f"ipdm {t_o:8.3f} ms {flop / t_o / 1e9:7.1f} TF | maxdiff {err:.2e}", flush=True)
print(f"weighted total per forward: miopen {tot_m:.1f} ms, ipdm {tot_o:.1f} ms")
