#!/usr/bin/env python3
"""step-level timeline of one chunk (chunk 2 of a workgroup's second tile) of the persistent split-bf16 Winograd kernel;
needs a diagnostic build: scripts/build_variant.sh trace conv_wino_bx3.hip -DIPDM_WBX3_TRACE, run with IPDM_LIB=_variants/libipdm_trace.so"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops, _lib
B, ci, co, hw = 28, 128, 128, 128
x = torch.randn(B, ci, hw, hw, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
U = ops.conv_wino_bx3_weight(w, fmt=os.environ.get('FMT', 'hx2'))
nblk = 256                                       # persistent kernel: one workgroup per CU
buf = torch.zeros(nblk * 4 + nblk * 64 + nblk * 8 * 18 + nblk * 8 * 6 + nblk * 8 * 18, dtype=torch.int64, device="cuda")
for _ in range(3): ops.conv2d_wino_bx3(x, U)
torch.cuda.synchronize()
_lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(buf.data_ptr()))
ops.conv2d_wino_bx3(x, U); torch.cuda.synchronize()
_lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(0))
t = buf.cpu()[nblk * 4:nblk * 68].view(nblk, 8, 8).double()
te = buf.cpu()[nblk * 68:nblk * (68 + 144)].view(nblk, 8, 18).double()
tf = buf.cpu()[nblk * (68 + 144):nblk * (68 + 144 + 48)].view(nblk, 8, 6).double()
tq = buf.cpu()[nblk * (68 + 144 + 48):].view(nblk, 8, 18).double()
# stamps (conv_wino_bx3.hip, IPDM_TR): 0 chunk start, 5 patches in registers (after the fragment / DMA wait and the LDS reads),
# 1..4 after MFMA steps 0..3, 7 after the chunk's barrier; 6 (DMA wait) exists only in the shared-raw-stage forms
wave_private = bool((t[:, :, 6] == 0).all())
seq = [0, 5, 1, 2, 3, 4, 7] if wave_private else [0, 1, 5, 2, 3, 4, 6, 7]
names = (["wait+patch reads", "step0", "step1", "step2", "step3", "end barrier"] if wave_private else
         ["st0(+B0 split)", "mid barrier", "step1 (dma)", "step2", "step3", "dma wait", "end barrier"])
d = torch.stack([t[:, :, seq[i + 1]] - t[:, :, seq[i]] for i in range(len(seq) - 1)], dim=-1)   # [blk, wave, segments]
med = d.median(dim=0).values
print("median cycles per segment, per wave (rows = wave 0..7; waves w and w+4 share a SIMD):")
print("   " + "  ".join(f"{n:>16s}" for n in names))
for wv in range(8):
    print(f"w{wv} " + "  ".join(f"{med[wv, i]:16.0f}" for i in range(len(names))), f"  total {med[wv].sum():.0f}")
for blk in (0, 100):
    base = t[blk, :, 0].min()
    print(f"workgroup {blk}: stamps relative to the first wave's chunk start ({', '.join(['start'] + names)})")
    for wv in range(8):
        print(f"w{wv} " + " ".join(f"{t[blk, wv, k] - base:7.0f}" for k in seq))

# epilogue of the second tile: stamp 0 start, then per round (exchange stores issued, barrier, transform + global stores issued, barrier)
if bool((te[:, :, 0] != 0).all()):
    de = te[:, :, 1:] - te[:, :, :-1]
    mede = de.median(dim=0).values
    print("epilogue, median cycles per segment and wave; rounds 0..3 x (exchange stores, barrier, reads+transform+stores, barrier):")
    for wv in range(8):
        print(f"w{wv} " + " | ".join(" ".join(f"{mede[wv, 4 * r + k]:6.0f}" for k in range(4)) for r in range(4)), f"  total {(te[:, wv, 16] - te[:, wv, 0]).median():.0f}")
if bool((tf[:, :, 0] != 0).all()):
    print("round 1's transform phase: phase start -> channel 0: reads + first sums, second sums + stores issued; channel 1: the same")
    df = (tf[:, :, 1:5] - tf[:, :, 0:4]).median(dim=0).values
    for wv in range(8):
        print(f"w{wv} " + " ".join(f"{df[wv, k]:6.0f}" for k in range(4)))
if bool((tq[:, :, 0] != 0).all()):
    nch = ci // 16
    w0 = tq[:, 0, :]
    seg = [(w0[:, k + 1] - w0[:, k]).median().item() for k in range(nch - 1)] + [(w0[:, 16] - w0[:, nch - 1]).median().item()]
    print("second tile, wave 0: cycles of chunk 0 .. %d:" % (nch - 1), " ".join(f"{v:.0f}" for v in seg),
          f"| epilogue {(w0[:, 17] - w0[:, 16]).median().item():.0f} | whole tile {(w0[:, 17] - w0[:, 0]).median().item():.0f}")
    print(f"first tile's epilogue end -> second tile's first chunk start: {(w0[:, 0] - te[:, 0, 16]).median().item():.0f} cycles (te of tile 1 is taken on tile index 1: see source)")
