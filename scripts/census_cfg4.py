#!/usr/bin/env python3
"""per-shape convolution census of ONE temporal score evaluation of config 4 (NCSN3DShallow on the 8 x 8 x T patches of a
128 x 128 x 24 series: 256 patches x 2 planes), HIP events per launch"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops
from inverseproblemwithdiffusionmodel_amd.helpers.load_model import reload_model
dev = torch.device("cuda")
net = reload_model("Diffusion3D", "CINE127", device=dev)
B = int(os.environ.get("BENCH_B", 512))
x = torch.randn(B, 64, 24, device=dev)
lab = torch.zeros(B, dtype=torch.long, device=dev)
with torch.no_grad():
    net(x, lab)
    reps = []
    for _ in range(3):
        ops.CONV_TRACE = []
        net(x, lab)
        torch.cuda.synchronize()
        reps.append(ops.CONV_TRACE)
        ops.CONV_TRACE = None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); net(x, lab); e1.record(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for rep in reps:
    for r in rep:
        ms = r["e0"].elapsed_time(r["e1"])
        key = (r["Cin"], r["Cout"], r["B"], r["H"], r["W"], r["k"], r["dil"], r.get("taps3d"), "wino" if r.get("wino") else "direct")
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += ms
tot = sum(a[1] for a in agg.values()) / len(reps)
print(f"forward {e0.elapsed_time(e1):.1f} ms; conv launches total {tot:.1f} ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{str(k):60s} n {a[0] // len(reps):3d}  {a[1] / a[0]:8.3f} ms each  {a[1] / len(reps):8.2f} ms  {a[1] / len(reps) / tot:6.1%}")
