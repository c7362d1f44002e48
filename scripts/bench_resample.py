#!/usr/bin/env python3
"""HBM bandwidth of the memory-bound kernels against the MI355X roof (8 TB/s spec, ~6.3 TB/s achievable):
upfirdn2d on the 36 call shapes of one NCSN++ forward (batch B), fused bias-act, the
InstanceNorm++ passes and the pooling / resize kernels.  Algorithmic bytes = 4*(elements read + written)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops
from inverseproblemwithdiffusionmodel_amd.models import up_or_down_sampling as uds

B = int(os.environ.get("BENCH_B", 8))


def timeit(fn, iters=20):
    """device time per call: `iters` calls captured into one hipGraph and replayed (what a graph-replayed forward pays;
    BENCH_EAGER=1 times eager launches instead, host launch overhead included)"""
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if os.environ.get("BENCH_EAGER") == "1":
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
    return best


def report(name, nbytes, t):
    print(f"{name:58s} {nbytes / 1e6:9.1f} MB {t * 1e6:9.1f} us {nbytes / t / 1e9:8.1f} GB/s", flush=True)
    return nbytes, t


print(f"B={B}")
tot_b = tot_t = 0.0
# (count, C, H) down then up, as in the celebahq_256 NCSN++ (two calls per resolution: h and x branch)
for cnt, C, H in [(2, 128, 256), (2, 128, 128), (2, 256, 64), (2, 256, 32), (2, 256, 16), (2, 256, 8), (1, 3, 256),
                  (1, 3, 128), (1, 3, 64), (1, 3, 32), (1, 3, 16), (1, 3, 8)]:
    x = torch.randn(B, C, H, H, device="cuda")
    t = timeit(lambda: uds.downsample_2d(x, (1, 3, 3, 1), factor=2))
    nb, _ = report(f"upfirdn2d down2 k4  ({B},{C},{H},{H})", 4 * x.numel() * 1.25, t)
    tot_b += cnt * nb; tot_t += cnt * t
for cnt, C, H in [(2, 256, 4), (2, 256, 8), (2, 256, 16), (2, 256, 32), (2, 256, 64), (2, 128, 128), (1, 3, 4), (1, 3, 8),
                  (1, 3, 16), (1, 3, 32), (1, 3, 64), (1, 3, 128)]:
    x = torch.randn(B, C, H, H, device="cuda")
    t = timeit(lambda: uds.upsample_2d(x, (1, 3, 3, 1), factor=2))
    nb, _ = report(f"upfirdn2d up2 k4    ({B},{C},{H},{H})", 4 * x.numel() * 5, t)
    tot_b += cnt * nb; tot_t += cnt * t
print(f"--> all 36 upfirdn2d calls of one NCSN++ forward: {tot_b / 1e6:.1f} MB in {tot_t * 1e3:.3f} ms = "
      f"{tot_b / tot_t / 1e9:.1f} GB/s (algorithmic)")

x = torch.randn(B, 128, 256, 256, device="cuda")
b = torch.randn(128, device="cuda")
report("fused_bias_act lrelu (B,128,256,256)", 8 * x.numel(), timeit(lambda: ops.fused_bias_act_raw(x, b, None, 3, 0, 0.2, 1.41)))
x = torch.randn(28, 128, 128, 128, device="cuda")
al = torch.ones(128, device="cuda")
coef = ops.instnorm_plus_coef(x, al, al, al)
y = torch.empty_like(x)
report("instnorm++ statistics (28,128,128,128)   [read x1, L2-hot x2]", 4 * x.numel(), timeit(lambda: ops.instnorm_plus_coef(x, al, al, al)))
report("affine+ELU (28,128,128,128)", 8 * x.numel(), timeit(lambda: ops.affine_act(x, coef, ops.ACT_ELU, out=y)))
report("maxpool5 (28,128,128,128)", 8 * x.numel(), timeit(lambda: ops.maxpool5(x)))
report("meanpool2 (28,128,128,128)", 5 * x.numel(), timeit(lambda: ops.meanpool2(x)))
report("add (28,128,128,128)", 12 * x.numel(), timeit(lambda: ops.add(x, y, out=y)))
xs = torch.randn(28, 128, 64, 64, device="cuda")
report("bilinear 64->128 accumulate (28,128,.,.)", 4 * (xs.numel() + 2 * x.numel()), timeit(lambda: ops.bilinear(xs, (128, 128), out=y, accumulate=True)))
g = torch.randn_like(x)
report("langevin (Philox) (28,128*128*128)", 12 * x.numel(), timeit(lambda: ops.langevin_step(x.view(28, -1), g.view(28, -1), 1e-3, 1e-2, seed=1)))
