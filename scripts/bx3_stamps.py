#!/usr/bin/env python3
"""per-workgroup phase times of the bf16x3 convolution from in-kernel s_memtime stamps (tuning aid)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops, _lib
B = 28
for ci, co, hw, dil in [(128, 128, 128, 1), (256, 256, 16, 1), (512, 512, 16, 1)]:
    x = torch.randn(B, ci, hw, hw, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wq = ops.conv_bx3_weight(w)
    buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
    for _ in range(3): ops.conv_bx3(x, wq, dilation=dil)
    torch.cuda.synchronize()
    _lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv_bx3(x, wq, dilation=dil); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    _lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(0))
    t = buf.cpu().view(-1, 4)
    t = t[t[:, 0] != 0].double()
    nblk = t.shape[0]
    pro, loop, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
    nch = ci // 16
    span = (t[:, 3].max() - t[:, 0].min()).item()
    print(f"{ci}->{co}@{hw}: blocks {nblk}; shader cycles: prologue {pro.median():.0f} loop {loop.median():.0f} "
          f"({loop.median() / nch:.1f}/chunk) epilogue {epi.median():.0f} total {(t[:,3]-t[:,0]).median():.0f}; "
          f"kernel span {span:.0f} cycles in {ms * 1e3:.0f} us -> {span / ms / 1e6:.2f} GHz; blocks/CU {nblk / 256:.1f}; "
          f"MFMA-bound loop = {nch * 9 * (4 if hw > 16 else 2) * 6 * 32} cycles per wave")
