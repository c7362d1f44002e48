#!/usr/bin/env python3
"""in-kernel cycle split of the 1-D Winograd kernel (s_memtime stamps: total, prologue, chunk loops, epilogues per workgroup)"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import _lib
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location("tw", os.path.join(os.path.dirname(__file__), "test_wino1d.py"))
tw = importlib.util.module_from_spec(spec); spec.loader.exec_module(tw)
B = 28
VAR = [("res+act", True, 1), ("act", False, 1), ("plain", False, 0), ("res", True, 0)]
for ci, co, hw, (vn, use_r, act) in [(128, 128, 128, v) for v in VAR] + [(256, 256, 64, v) for v in VAR]:
    x = torch.randn(B, ci, hw, hw, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    bias = torch.randn(co, device="cuda"); r = torch.randn(B, co, hw, hw, device="cuda") if use_r else None
    blob = tw.pack(w)
    buf = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    for _ in range(3): tw.conv1d(x, blob, co, bias, r, act)
    torch.cuda.synchronize()
    _lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); tw.conv1d(x, blob, co, bias, r, act); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    _lib.lib.ipdm_debug_set_stamp_buffer(_lib.P(0))
    t = buf.cpu()[:256 * 4].view(-1, 4).double()
    t = t[t[:, 0] != 0]
    te = buf.cpu()[256 * 4:256 * 8].view(-1, 4).double()
    tiles = B * (hw // 8) * (hw // 32) * (co // 128) / t.shape[0]
    nch = ci // 16
    print(f"{ci}->{co}@{hw} {vn:8s}: {ms * 1e3:.0f} us; {t.shape[0]} WGs x {tiles:.2f} passes; cycles total median {t[:, 0].median():.0f} max {t[:, 0].max():.0f} "
          f"(clock {t[:, 0].max() / ms / 1e6:.2f} GHz); prologue {t[:, 1].median():.0f}; per pass: loop {t[:, 2].median() / tiles:.0f} "
          f"= {t[:, 2].median() / tiles / nch:.0f} per chunk (MFMA floor 4608), epilogue {t[:, 3].median() / tiles:.0f}", flush=True)
    if te.abs().sum() > 0:
        m = te.median(0).values / tiles / 8
        print(f"      per round (wave 0): exchange stores {m[0]:.0f}, barrier {m[1]:.0f}, transform + stores {m[2]:.0f}, barrier {m[3]:.0f}", flush=True)
