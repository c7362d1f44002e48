import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, ci, co, hw, pool) in [(28, 128, 128, 128, False), (28, 256, 256, 64, False), (28, 128, 256, 128, True), (28, 256, 256, 32, False)]:
    x = torch.randn(B, ci, hw, hw, device="cuda"); w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    U = ops.conv_wino_bx3_weight(w)
    oh = hw // 2 if pool else hw
    r = torch.randn(B, co, oh, oh, device="cuda")
    alpha = torch.randn(co, device="cuda"); 
    t0 = timeit(lambda: ops.conv2d_wino_bx3(x, U, None, r, pool2=pool))
    t1 = timeit(lambda: ops.conv2d_wino_bx3(x, U, None, r, pool2=pool, want_stats=True))
    y = ops.conv2d_wino_bx3(x, U, None, r, pool2=pool)
    ys = ops.conv2d_wino_bx3(x, U, None, r, pool2=pool, want_stats=True)
    t2 = timeit(lambda: ops.instnorm_plus_coef(y, alpha, alpha, alpha))
    t3 = timeit(lambda: ops.instnorm_plus_coef(ys, alpha, alpha, alpha))
    print(f"{ci}->{co} @{hw} pool={pool}: conv {t0*1e3:.1f} us, conv+stats {t1*1e3:.1f} us (+{(t1-t0)*1e3:.1f}); coef from tensor {t2*1e3:.1f} us, from partials {t3*1e3:.1f} us (-{(t2-t3)*1e3:.1f})")
