#!/usr/bin/env python3
"""Range probe for the f16x2 kernels' contract (|activation| < 65504): max |input| of every
split-operand convolution launch in (a) the NCSNv2Deepest headline forward at a head-of-schedule state (|x| ~ 1e3) and a
tail state, (b) the full-size NCSN++ 256 forward of tests/test_score_sde_gpu.py.  Prints the largest inputs and the first
launch whose output is not finite.  GPU only."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inverseproblemwithdiffusionmodel_amd import ops, engine

LOG = []


def wrap(name):
    fn = getattr(ops, name)

    def w(x, wq, *a, **k):
        y = fn(x, wq, *a, **k)
        outs = [t for t in (y if isinstance(y, tuple) else (y,)) if t is not None]
        LOG.append(dict(op=name, fmt=wq.fmt, shape=tuple(x.shape), cout=wq.Cout, xmax=float(x.abs().max()),
                        xrms=float(x.pow(2).mean().sqrt()), finite_in=bool(torch.isfinite(x).all()),
                        finite=all(bool(torch.isfinite(t).all()) for t in outs), ymax=float(outs[0].abs().max())))
        return y
    setattr(ops, name, w)


wrap("conv_bx3")
wrap("conv2d_wino_bx3")


def report(tag):
    bad = next((r for r in LOG if not r["finite"]), None)
    top = sorted(LOG, key=lambda r: -r["xmax"])[:6]
    small = sorted(LOG, key=lambda r: r["xrms"])[:4]
    print(f"== {tag}: {len(LOG)} launches; first non-finite output: {bad}")
    for r in top:
        print("   max", r)
    for r in small:
        print("   min-rms", {k: r[k] for k in ('op', 'shape', 'xmax', 'xrms')})
    LOG.clear()


dev = torch.device("cuda")
which = sys.argv[1:] or ["v2", "pp"]
if "v2" in which:
    prob = engine.build_problem(dev, 2)
    net = prob.scorenet
    for tag, scale, lab in [("NCSNv2Deepest head (|x|~1.3e3, level 11)", 450.0, 11), ("NCSNv2Deepest tail (level 2310)", 0.3, 2310)]:
        x = torch.randn(4, 1, 128, 128, device=dev) * scale
        with torch.no_grad():
            net(x, torch.full((4,), lab, dtype=torch.long, device=dev))
        report(tag)
if "pp" in which:
    from inverseproblemwithdiffusionmodel_amd.configs import ve_ncsnpp
    from inverseproblemwithdiffusionmodel_amd.models import ncsnpp
    from inverseproblemwithdiffusionmodel_amd.synthetic import synth_state_dict
    cfg = ve_ncsnpp.get_config()
    cfg.device = dev
    net = ncsnpp.NCSNpp(cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=0)
    net.load_state_dict(sd, strict=False)
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(220)
    x = torch.rand(1, 3, 256, 256, generator=gen) + 2.0 * torch.randn(1, 3, 256, 256, generator=gen)
    with torch.no_grad():
        net(x.cuda(), torch.tensor([10.0], device=dev))
    report("NCSN++ 256 (synthetic weights)")
