#!/usr/bin/env python3
"""per-workgroup phase times of the Winograd kernel from in-kernel s_memtime stamps (tuning aid)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops, _lib
B, ci, co, hw = 28, 128, 128, 128
x = torch.randn(B, ci, hw, hw, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
U = ops.conv_wino_weight(w)
nblk = B * (hw // 8) * (hw // 32) * (co // 64)
buf = torch.zeros(nblk * 4, dtype=torch.int64, device="cuda")
for _ in range(3): ops.conv2d_wino(x, U)
torch.cuda.synchronize()
_lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(buf.data_ptr()))
ops.conv2d_wino(x, U); torch.cuda.synchronize()
_lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(0))
t = buf.cpu().view(nblk, 4).double()
pro, loop, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
print(f"blocks {nblk}; s_memtime ticks (shader clocks): prologue {pro.median():.0f}  loop {loop.median():.0f} "
      f"({loop.median() / 16:.0f} per chunk; MFMA-bound would be 4096)  epilogue {epi.median():.0f}  total {(t[:,3]-t[:,0]).median():.0f}")
print("kernel span:", (t[:, 3].max() - t[:, 0].min()).item(), " blocks per CU ~", nblk / 256)
