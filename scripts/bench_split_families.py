#!/usr/bin/env python3
"""Per-layer A/B of the two split-operand convolution families on the NCSNv2Deepest conv census (B = 28): time (10 calls
per hipGraph replay, HIP events) and max error against a float64 convolution, relative to the output range, for
  hx2  two fp16 pieces, three fp16 MFMAs per product        bx3  three bf16 pieces, six bf16 MFMAs per product
  f32  the exact-fp32 MFMA kernels (error column only)
Inputs: x ~ ELU(N(0,1)) scaled by BENCH_XSCALE (default 1), w ~ N(0, 1/(9 Cin)).  GPU only."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops

B = int(os.environ.get("BENCH_B", 28))
XS = float(os.environ.get("BENCH_XSCALE", 1.0))
SHAPES = [  # (count per forward, Cin, Cout, H, dil)
    (18, 128, 128, 128, 1), (9, 128, 128, 64, 1), (8, 256, 256, 64, 1), (17, 256, 256, 32, 1), (25, 256, 256, 16, 1),
    (16, 512, 512, 16, 2), (5, 512, 512, 16, 4), (1, 128, 256, 128, 1), (1, 256, 512, 16, 2), (16, 512, 512, 16, 1),
]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


tot = {"hx2": 0.0, "bx3": 0.0}
print(f"B={B} xscale={XS}")
for cnt, ci, co, hw, dil in SHAPES:
    gen = torch.Generator(device="cuda").manual_seed(ci + hw)
    x = F.elu(torch.randn(B, ci, hw, hw, device="cuda", generator=gen)) * XS
    w = torch.randn(co, ci, 3, 3, device="cuda", generator=gen) / (9 * ci) ** 0.5
    bias = torch.randn(co, device="cuda", generator=gen)
    ref = F.conv2d(x[:2].double(), w.double(), bias.double(), padding=dil, dilation=dil)
    rng = ref.abs().max()
    row = f"{cnt:3d}x {ci:4d}->{co:4d} @{hw:3d}^2 d{dil}:"
    wino = ops.wino_bx3_pays(ci, co, hw, hw, dil)
    for fmt in ("hx2", "bx3"):
        if wino:
            U = ops.conv_wino_bx3_weight(w, fmt=fmt)
            fn = lambda: ops.conv2d_wino_bx3(x, U, bias, dilation=dil)
            y = ops.conv2d_wino_bx3(x[:2].contiguous(), U, bias, dilation=dil)
        else:
            wq = ops.conv_bx3_weight(w, fmt=fmt)
            fn = lambda: ops.conv_bx3(x, wq, bias, dilation=dil)
            y = ops.conv_bx3(x[:2].contiguous(), wq, bias, dilation=dil)
        t = timeit(fn)
        tot[fmt] += cnt * t
        err = ((y.double() - ref).abs().max() / rng).item()
        row += f"  {fmt} {'wino' if wino else 'direct'} {t * 1e3:7.1f} us err {err:.2e}"
    wq = ops.conv_bx3_weight(w, fmt="hx2")
    yd = ops.conv_bx3(x[:2].contiguous(), wq, bias, dilation=dil)
    row += f"  | hx2 direct err {((yd.double() - ref).abs().max() / rng).item():.2e}"
    y32 = ops.conv2d(x[:2].contiguous(), ops.conv_pack_weight(w), bias, dilation=dil)
    row += f"  f32-mfma err {((y32.double() - ref).abs().max() / rng).item():.2e}"
    if ops.conv_wino_supported(ci, co, hw, hw, dil):
        yw = ops.conv2d_wino(x[:2].contiguous(), ops.conv_wino_weight(w), bias, dilation=dil)
        row += f"  f32-wino err {((yw.double() - ref).abs().max() / rng).item():.2e}"
    print(row, flush=True)
print(f"weighted conv total per forward: hx2 {tot['hx2']:.2f} ms, bx3 {tot['bx3']:.2f} ms")
