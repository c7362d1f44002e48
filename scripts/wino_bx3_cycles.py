#!/usr/bin/env python3
"""whole-workgroup cycle counts (s_memtime) of the persistent split-bf16 Winograd kernel: cycles per (tile, chunk) and the
implied shader clock.  Used with diagnostic builds (scripts/build_variant.sh) to price pieces of the kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops, _lib
B = 28
for ci, co, hw in [(128, 128, 128), (256, 256, 64)]:
    x = torch.randn(B, ci, hw, hw, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    U = ops.conv_wino_bx3_weight(w, fmt=os.environ.get('FMT', 'hx2'))
    buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
    for _ in range(3): ops.conv2d_wino_bx3(x, U)
    torch.cuda.synchronize()
    _lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv2d_wino_bx3(x, U); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    _lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(0))
    t = buf.cpu()[:256 * 4].view(-1, 4)            # persistent kernel: one workgroup per CU, 4 stamps each
    t = t[t[:, 0] != 0].double()
    tot = (t[:, 3] - t[:, 0])
    px_per_tile = 256
    tiles = B * hw * hw / px_per_tile * (co // 64)
    per_wg = tiles / t.shape[0]
    nch = ci // 16
    mfma_bound = (48 if os.environ.get('FMT', 'hx2') == 'hx2' else 96) * 32
    print(f"    last tile's epilogue (t3 - t2): median {(t[:, 3] - t[:, 2]).median():.0f} cycles; prologue (t1 - t0): {(t[:, 1] - t[:, 0]).median():.0f}")
    print(f"{ci}->{co}@{hw}: {ms * 1e3:.0f} us, {t.shape[0]} workgroups x {per_wg:.1f} tiles; cycles/WG median {tot.median():.0f} max {tot.max():.0f}; "
          f"per (tile, chunk) {tot.median() / per_wg / nch:.0f} (MFMA-bound {mfma_bound}); implied clock {tot.max() / ms / 1e6:.2f} GHz")
