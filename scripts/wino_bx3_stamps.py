#!/usr/bin/env python3
"""per-workgroup phase times of the split-bf16 Winograd kernel from in-kernel s_memtime stamps (tuning aid)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops, _lib
B = 28
for ci, co, hw, dil in [(128, 128, 128, 1), (256, 256, 64, 1), (256, 256, 16, 1), (512, 512, 16, 2)]:
    x = torch.randn(B, ci, hw, hw, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    U = ops.conv_wino_bx3_weight(w)
    buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
    for _ in range(3): ops.conv2d_wino_bx3(x, U, dilation=dil)
    torch.cuda.synchronize()
    _lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv2d_wino_bx3(x, U, dilation=dil); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    _lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(0))
    t = buf.cpu().view(-1, 4)
    t = t[t[:, 0] != 0].double()
    nblk = t.shape[0]
    pro, loop, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
    nch = ci // 16
    tot = (t[:, 3] - t[:, 0]).median()
    print(f"{ci}->{co}@{hw} d{dil}: {ms * 1e3:.0f} us, blocks {nblk} ({nblk / 256:.1f}/CU); cycles: prologue {pro.median():.0f} "
          f"loop {loop.median():.0f} ({loop.median() / nch:.0f}/chunk; MFMA-bound 3072) epilogue {epi.median():.0f} total {tot:.0f} "
          f"-> implied clock {tot * nblk / 256 / ms / 1e6:.2f} GHz")
