#!/usr/bin/env python3
"""f16x2 family, per layer shape of the headline census (B = 28): the Winograd kernel against the direct kernel (and its tile
configurations, IPDM_BX3_CFG) -- 10 calls per hipGraph replay, HIP events.  With (raw + activated) outputs and a residual, as
the RefineNet layers run."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops
from bench_split_families import timeit, SHAPES, B

print(f"B={B}")
for cnt, ci, co, hw, dil in SHAPES:
    gen = torch.Generator(device="cuda").manual_seed(ci + hw)
    x = F.elu(torch.randn(B, ci, hw, hw, device="cuda", generator=gen))
    w = torch.randn(co, ci, 3, 3, device="cuda", generator=gen) / (9 * ci) ** 0.5
    res = torch.randn(B, co, hw, hw, device="cuda", generator=gen)
    U = ops.conv_wino_bx3_weight(w, fmt="hx2")
    wq = ops.conv_bx3_weight(w, fmt="hx2")
    am = ops.absmax_per_image(x)
    row = f"{cnt:3d}x {ci:4d}->{co:4d} @{hw:3d}^2 d{dil}:"
    for name, fn in (("wino", lambda: ops.conv2d_wino_bx3(x, U, None, dilation=dil, in_amax=am)),
                     ("direct", lambda: ops.conv_bx3(x, wq, None, dilation=dil, in_amax=am)),
                     ("wino+res+act", lambda: ops.conv2d_wino_bx3(x, U, None, res, act_out=ops.ACT_ELU, dilation=dil, in_amax=am)),
                     ("direct+res+act", lambda: ops.conv_bx3(x, wq, None, residual=res, act_out=ops.ACT_ELU, dilation=dil, in_amax=am))):
        row += f"  {name} {timeit(fn) * 1e3:7.1f} us"
    print(row, flush=True)
