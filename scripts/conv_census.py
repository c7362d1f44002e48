#!/usr/bin/env python3
"""per-shape convolution census of one score evaluation of the headline workload (HIP events per launch)"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import engine
dev = torch.device("cuda")
prob = engine.build_problem(dev, int(os.environ.get("BENCH_B", 28)) // 2)
run = engine.IterationRunner(prob, use_graph=False)
run.run(0)
x, labels = run.x.clone(), run.st["labels"].clone()
engine.conv_census(prob.scorenet, x, labels)
reps = [engine.conv_census(prob.scorenet, x, labels) for _ in range(5)]
agg = collections.OrderedDict()
for rep in reps:
    for r in rep:
        kind = "w1d" if r.get("wino1d") else "wino" if r.get("wino") else "direct"
        kind += ("+pool" if r.get("pool2") else "") + ("+res" if r.get("res") else "") + f"+{r.get('n_out', 1)}out" + (f"+ks{r['ksplit']}" if r.get("ksplit", 1) > 1 else "") + ("+stats" if r.get("stats") else "")
        key = (r["Cin"], r["Cout"], r["H"], r["W"], r["k"], r["dil"], kind)
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
tot = sum(a[1] for a in agg.values()) / len(reps)
print(f"{'shape':58s} {'n':>3s} {'ms each':>8s} {'ms total':>9s} {'TF/s':>7s} {'share':>6s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    n = a[0] // len(reps)
    print(f"{str(k):58s} {n:3d} {a[1] / a[0]:8.3f} {a[1] / len(reps):9.2f} {a[2] / a[1] / 1e9:7.1f} {a[1] / len(reps) / tot:6.1%}")
print(f"total {tot:.2f} ms")
