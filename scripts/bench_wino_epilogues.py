#!/usr/bin/env python3
"""what each fused epilogue of the wide split-operand Winograd kernel costs against the separate pass it replaces:
plain / + residual / + statistics partials / + 2x2 mean (ConvMeanPool), hipGraph-timed on the headline's layer shapes"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops

B = int(os.environ.get("BENCH_B", 28))
FMT = os.environ.get("BENCH_FMT", ops.CONV_IMPL)


def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for ci, co, hw in ((128, 128, 128), (128, 256, 128), (256, 256, 64), (256, 256, 32)):
    x = torch.randn(B, ci, hw, hw, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.03
    bias = torch.randn(co, device="cuda")
    U = ops.conv_wino_split_weight(w, FMT)
    res = torch.randn(B, co, hw, hw, device="cuda")
    resp = torch.randn(B, co, hw // 2, hw // 2, device="cuda")
    y = ops.conv2d_wino_bx3(x, U, bias)
    rows = [("plain", lambda: ops.conv2d_wino_bx3(x, U, bias)),
            ("+ residual", lambda: ops.conv2d_wino_bx3(x, U, bias, residual=res)),
            ("+ stats", lambda: ops.conv2d_wino_bx3(x, U, bias, want_stats=True)),
            ("+ residual + stats", lambda: ops.conv2d_wino_bx3(x, U, bias, residual=res, want_stats=True)),
            ("+ act copy", lambda: ops.conv2d_wino_bx3(x, U, bias, act_out=ops.ACT_ELU)),
            ("separate: plane stats of the result", lambda: ops.plane_stats(y)),
            ("separate: add", lambda: ops.add(y, res))]
    if hw >= 64:
        rows += [("pool2 + residual + stats", lambda: ops.conv2d_wino_bx3(x, U, bias, residual=resp, pool2=True, want_stats=True)),
                 ("pool2 + residual", lambda: ops.conv2d_wino_bx3(x, U, bias, residual=resp, pool2=True)),
                 ("separate: meanpool2", lambda: ops.meanpool2(y))]
    print(f"{ci} -> {co} @ {hw}^2, B = {B}, {FMT}")
    for name, fn in rows:
        try:
            print(f"    {name:38s} {graph_time(fn):8.1f} us", flush=True)
        except Exception as e:                      # an epilogue that is not built for this shape
            print(f"    {name:38s} unsupported ({type(e).__name__})", flush=True)
