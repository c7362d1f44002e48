#!/usr/bin/env python3
"""Benchmark of the ALD reconstruction hot path (BASELINE.json metric: ALD reconstructions/sec, 128x128 complex,
R=40, 4 coils) on N MI355X of one node.

    python bench.py                       # N=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (SURVEY.md 8d, config 3 of BASELINE.json): 105 posterior samples block-partitioned over 8 GPUs
= 14 on rank 0 and 13 on the others; for N < 8 every rank keeps that per-GPU share (rank 0: 14, others: 13),
so the critical path per GPU is the same at every N ("weak" scaling) and N = 8 is exactly the 105-sample run.
Score net NCSNv2Deepest (ngf 128, 94.1 M parameters, seeded random init), sigma 348 -> 0.01 geometric in
2311 levels x 3 Langevin steps, step_lr 9e-7, L2Penalty proximal, final denoise; synthetic phantom k-space.

One "step" = one Langevin iteration of a rank's whole sample batch: the score network on the (2B, 1, 128, 128)
[real | imaginary] batch + the fused Langevin/SENSE-proximal kernel, replayed as one hipGraph.  A full
reconstruction is 2311*3 = 6933 such iterations plus one denoising score evaluation (counted as a 6934th
iteration); the iteration cost does not depend on the noise level (same launches, same shapes), so
    value = total samples / (ms_per_step * 6934)
with the timed steps strided evenly over the 6933-iteration schedule.  `--full` runs the entire schedule instead
and reports the directly measured rate (takes ~6934/steps times longer).

The JSON line also carries
  roofline      fp32 MFMA roofline of the dominant kernel family (conv_mfma_kernel<...>, the dense 3x3/1x1
                convolutions): algorithmic FLOPs of the 113 conv launches of one score evaluation divided by
                their summed durations, measured with HIP events around every conv launch on the launch stream.
  cpu_baseline  the CPU oracle (oracle/, a torch-CPU restatement pinned to the reference) timed on this host's
                cores on a bounded slice of the same workload (1 sample, a few iterations), rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md, dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
L_LEVELS, N_EACH = 2311, 3
ITER_PER_RECON = L_LEVELS * N_EACH + 1  # + the denoising score evaluation


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--full", action="store_true", help="run the whole 6933-iteration schedule + denoise")
    ap.add_argument("--samples-rank0", type=int, default=14)
    ap.add_argument("--samples-other", type=int, default=13)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the fp32-MFMA comparison measurement")
    ap.add_argument("--cpu-steps-b1", type=int, default=60, help="CPU baseline: timed iterations at B=1 (after 3 warm-ups)")
    ap.add_argument("--cpu-steps-b8", type=int, default=8, help="CPU baseline: timed iterations at B=8 (0 = skip)")
    ap.add_argument("--no-cpu-mnist", action="store_true", help="skip the end-to-end MNIST-shaped CPU anchor")
    ap.add_argument("--no-full-batch", action="store_true", help="N=1: skip the extra 105-sample measurement")
    ap.add_argument("--no-graph", action="store_true")
    return ap.parse_args()


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU share of this process: affinity mask capped by the cgroup quota (a GPU box gives 16 per GPU)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, q // int(g.read())))
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get("IPDM_CPU_THREADS", 32))))


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(prob, steps_b1, steps_b8, mnist):
    """SURVEY.md 8(d) / BASELINE.md 3: the CPU restatement of the reference sampler (oracle/, pinned to the reference at
    this very size by tests/golden/g20_fullsize_ald.npz) timed on this host's cores: `steps` consecutive Langevin + proximal
    iterations after 3 warm-up iterations at B = 1 and at B = 8 (n_steps_each = 1, so a level is one iteration; the cost
    per iteration does not depend on the level), extrapolated linearly to the 6934 iterations of a reconstruction, plus
    the un-extrapolated anchor: the whole MNIST-shaped unconditional schedule (config 1: 32x32, 232 levels x 3 steps +
    denoise = 697 score evaluations) end to end."""
    from oracle import scorenet as oracle_net, ald as oracle_ald, kspace as oracle_kspace
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads ({cpu_model()}) ...")
    sd = {k: v.detach().cpu() for k, v in prob.scorenet.state_dict().items()}
    maps, mask = prob.op.sens_maps.numpy(), prob.op.random_under_fourier.mask.numpy()
    score = lambda x, lab: oracle_net.ncsnv2_deepest(x, lab, sd)
    out = {}

    def timed_run(B, steps):
        meas = prob.measurement[:, :1].cpu().numpy().repeat(B, axis=1)
        gen = torch.Generator().manual_seed(0)
        noise = lambda like: torch.randn(like.shape, generator=gen)
        run = lambda levels: oracle_ald.ald_sense_real_imag(score, prob.sigmas.cpu().numpy(), meas, maps, mask,
                                                            prob.params["step_lr"], 1, 1.0, False, noise, n_levels=levels)
        with torch.no_grad():
            run(3)                                           # 3 warm-up iterations
            t0 = time.perf_counter()
            run(steps)
            dt = (time.perf_counter() - t0) / steps
        return dict(batch=B, steps=steps, warmup=3, ms_per_iteration=dt * 1e3,
                    reconstructions_per_s=B / (dt * ITER_PER_RECON))
    out["b1"] = timed_run(1, steps_b1)
    log(f"cpu baseline B=1: {out['b1']['ms_per_iteration']:.0f} ms per iteration")
    if steps_b8 > 0:
        out["b8"] = timed_run(8, steps_b8)
        log(f"cpu baseline B=8: {out['b8']['ms_per_iteration']:.0f} ms per iteration")
    if mnist:
        sig = oracle_kspace.get_sigmas(50.0, 0.01, 232)
        sd32 = dict(sd, sigmas=torch.from_numpy(sig))
        gen = torch.Generator().manual_seed(1)
        x0 = torch.rand(1, 1, 32, 32, generator=gen)
        with torch.no_grad():
            t0 = time.perf_counter()
            x = oracle_ald.ald_unconditional(lambda x_, lab: oracle_net.ncsnv2_deepest(x_, lab, sd32), sig, x0, 6.2e-6, 3,
                                             True, lambda like: torch.randn(like.shape, generator=gen))
            dt = time.perf_counter() - t0
        out["mnist_full"] = dict(seconds=dt, score_evaluations=232 * 3 + 1, finite=bool(torch.isfinite(x).all()),
                                 note="config 1 shape: 1 sample, 32x32, NCSNv2Deepest ngf 128, sigma 50 -> 0.01 x 232 levels "
                                      "x 3 steps + denoise, run END TO END (no extrapolation)")
        log(f"cpu baseline MNIST-shaped full schedule: {dt:.1f} s")
    best = max((v for k, v in out.items() if k in ("b1", "b8")), key=lambda v: v["reconstructions_per_s"])
    return dict(value=best["reconstructions_per_s"], unit="reconstructions/s", cores=cores, cpu_model=cpu_model(),
                kind="port",
                sample=f"B={best['batch']}: {best['steps']} Langevin+proximal iterations after 3 warm-up iterations "
                       f"({best['ms_per_iteration']:.0f} ms per iteration, 2 score evaluations per sample each) of the same "
                       f"128x128 R=40 4-coil workload, extrapolated to {ITER_PER_RECON} iterations; torch-CPU fp32 oracle "
                       "(oracle/, pinned to the reference at this size by tests/golden/g20), all host cores",
                **out)


def upfirdn2d_hbm(dev):
    """north_star asks for upfirdn2d's HBM GB/s in the report: the largest call of an NCSN++ celebahq_256 forward at B = 8
    (FIR down-sampling of (8,128,256,256): 268.4 MB read + 67.1 MB written = 335.5 MB algorithmic, and its up-sampling twin
    (8,128,128,128) -> 256^2: 67.1 MB read + 268.4 MB written), 20 calls per hipGraph replay, HIP events on the launch
    stream; `traffic_ratio` is the PMC measurement of profiles/r02_upfirdn2d_pmc.csv (FETCH x2 + WRITE over algorithmic)."""
    from inverseproblemwithdiffusionmodel_amd.models import up_or_down_sampling as uds
    res = {"peak": 8000.0, "unit": "GB/s", "traffic_ratio": 1.02, "traffic_source": "profiles/r02_upfirdn2d_pmc.csv"}

    def timed(fn, iters=20):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
        return best
    try:
        x = torch.randn(8, 128, 256, 256, device=dev)
        t = timed(lambda: uds.downsample_2d(x, (1, 3, 3, 1), factor=2))
        nb = 4 * x.numel() * 1.25
        res["down2_8x128x256x256"] = {"algorithmic_MB": nb / 1e6, "us": t * 1e6, "achieved": nb / t / 1e9, "frac": nb / t / 1e9 / 8000.0}
        x2 = torch.randn(8, 128, 128, 128, device=dev)
        t2 = timed(lambda: uds.upsample_2d(x2, (1, 3, 3, 1), factor=2))
        nb2 = 4 * x2.numel() * 5
        res["up2_8x128x128x128"] = {"algorithmic_MB": nb2 / 1e6, "us": t2 * 1e6, "achieved": nb2 / t2 / 1e9, "frac": nb2 / t2 / 1e9 / 8000.0}
        res["achieved"], res["frac"] = res["down2_8x128x256x256"]["achieved"], res["down2_8x128x256x256"]["frac"]
        del x, x2
        torch.cuda.empty_cache()
    except Exception as e:                                    # never lose the headline line to the extra measurement
        res["error"] = repr(e)[:300]
    return res


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal knobs (1-GPU box): IPDM_BENCH_DEVICE pins every rank to one card, IPDM_DIST_BACKEND=gloo avoids RCCL's
    # one-rank-per-GPU rule; the driver's multi-GPU runs use neither (rank r -> cuda:r, RCCL)
    dev_index = int(os.environ.get("IPDM_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("IPDM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from inverseproblemwithdiffusionmodel_amd import engine, ops, sharding

    n_local = args.samples_rank0 if rank == 0 else args.samples_other
    total = args.samples_rank0 + args.samples_other * (world - 1)
    offset = 0 if rank == 0 else args.samples_rank0 + args.samples_other * (rank - 1)
    prob = engine.build_problem(dev, n_local, R=40, H=128, W=128, num_sens=4, seed=0)
    runner = engine.IterationRunner(prob, seed=0, sample_offset=offset, use_graph=not args.no_graph)
    n_sched = runner.n_iterations                                   # 6933
    log(f"rank {rank}: setup done, {n_local} local samples")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.full:
        steps, warm = n_sched, 0
        order = np.arange(n_sched)
    else:
        steps, warm = args.steps, args.warmup
        order = (np.arange(warm + steps) * (n_sched // max(warm + steps, 1))) % n_sched   # strided over the schedule
    for k in order[:warm]:
        runner.run(int(k))
    if warm == 0:
        runner.run(0)                                                # capture outside the timed region
    barrier()
    t0 = time.perf_counter()
    for k in order[warm:warm + steps]:
        runner.run(int(k))
    if args.full:                                                    # denoise: one more score evaluation
        with torch.no_grad():
            runner.st["labels"].fill_(L_LEVELS - 1)
            prob.scorenet(runner.x, runner.st["labels"])
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    log(f"rank {rank}: {steps} steps in {elapsed:.3f} s")
    finite = bool(torch.isfinite(runner.x).all())

    # the one collective of the path: posterior moments of the current samples
    post = sharding.all_reduce_posterior(runner.current(), total)
    ms_per_step = elapsed * 1e3 / steps
    if args.full:
        value = total / elapsed
    else:
        value = total / (ms_per_step * 1e-3 * ITER_PER_RECON)

    out = {
        "metric": "ALD reconstructions/sec (128x128 complex, R=40, 4-coil)",
        "value": value, "unit": "reconstructions/s", "n_gpus": world, "steps": steps, "warmup": warm,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32",
                  "bx3": "f32 (convolutions: exact bf16x3 split on the bf16 MFMA, fp32 accumulate)",
                  "hx2": ("f32 (convolutions: operands as two fp16 pieces = 22 significand bits + 2 signs, three fp16 MFMAs per "
                          "product, fp32 accumulate; every image scaled into fp16's range by an exact power of two from its "
                          "producer's per-image maxima: any fp32 input, DESIGN.md 4.2)" if ops.dynamic_range() else
                          "f32 (convolutions: two fp16 pieces per operand, three fp16 MFMAs per product, fp32 accumulate; STATIC range "
                          "contract |x| < 65504, fp32-faithful for activations of O(1) and above only: IPDM_HX2_DYNAMIC=0)")}[ops.CONV_IMPL],
        "data": "synthetic",
        "config": {
            "workload": "ACDC-style 128x128 complex SENSE R=40 4-coil ALD reconstruction, NCSNv2Deepest ngf=128 "
                        "(94.1M params, seeded random init), sigma 348->0.01 x 2311 levels x 3 steps, L2Penalty, denoise; "
                        f"{total} posterior samples = {args.samples_rank0} on rank 0 + {args.samples_other} per further "
                        "rank (the 105-sample / 8-GPU partition of BASELINE config 3)",
            "step": "one Langevin iteration of a rank's sample batch: score net on (2B,1,128,128) + fused "
                    "Langevin/SENSE-proximal kernel, one hipGraph replay",
            "iterations_per_reconstruction": ITER_PER_RECON,
            "value_formula": "total_samples / elapsed" if args.full else
                             "total_samples / (ms_per_step * iterations_per_reconstruction)",
            "samples_total": total, "samples_this_rank": n_local, "parallelism": f"sample-sharded x{world}",
            "hip_graph": not args.no_graph, "state_finite": finite,
            "posterior_mag_mean": float(post["mag_mean"].mean()),
        },
    }

    if rank == 0:
        # ---- roofline of the dominant kernel family: per-launch HIP events around every conv of one forward
        x = runner.x.clone()
        labels = runner.st["labels"].clone()
        engine.conv_census(prob.scorenet, x, labels)                 # warm
        reps = [engine.conv_census(prob.scorenet, x, labels) for _ in range(3)]
        flops = sum(r["flops"] for r in reps[0])
        # Winograd F(2x2,3x3) launches execute 16 instead of 36 multiply-adds per 2x2 output tile
        executed = sum(r["flops"] / (1.5 if r.get("wino1d") else 2.25 if r.get("wino") else 1.0) for r in reps[0])
        n_wino = sum(1 for r in reps[0] if r.get("wino"))
        # algorithmic HBM bytes of a launch: input + (residual) read once, each requested output written once,
        # weights read once
        bytes_alg = sum(4.0 * r["B"] * r["H"] * r["W"] * (r["Cin"] + r["Cout"] * (int(r.get("res", False)) + r.get("n_out", 1))
                                                          * (0.25 if r.get("pool2") else 1.0))
                        + 4.0 * r["Cin"] * r["Cout"] * r["k"] ** 2 for r in reps[0]) / len(reps[0])
        conv_ms = float(np.median([sum(r["ms"] for r in rep) for rep in reps]))
        log(f"conv census: {len(reps[0])} launches, {conv_ms:.2f} ms, {flops / 1e12:.3f} TFLOP per step")
        achieved = flops / (conv_ms * 1e-3) / 1e12
        # HBM traffic per conv launch comes from PMC counters, which only rocprofv3 can collect (scripts/profile_pmc.sh:
        # separate FETCH_SIZE / WRITE_SIZE passes over THIS command); the number is therefore a recorded measurement and is
        # stamped with the kernel family and source revision it was taken on -- it is dropped when the family differs
        traffic, traffic_source = None, None
        pmc = os.path.join(REPO, "profiles", "conv_hbm_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                rec = json.load(f)
            if rec.get("conv_impl", "bx3") == ops.CONV_IMPL:
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_source = {k: rec.get(k) for k in ("conv_impl", "measured_at_commit", "measured_on", "launches_profiled")}
                traffic_source["file"] = "profiles/conv_hbm_traffic.json"
        n_bx3 = sum(1 for r in reps[0] if r.get("bx3"))
        mfma_per_product = 3.0 if ops.CONV_IMPL == "hx2" else 6.0
        # 16-bit MFMA FLOPs actually executed: three (fp16 pair) or six (bf16 triple) per fp32 multiply-add, 2.25x fewer on
        # the Winograd launches
        flops_bx3 = sum(r["flops"] / (1.5 if r.get("wino1d") else 2.25 if r.get("wino") else 1.0) for r in reps[0] if r.get("bx3"))
        common = {
            "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": bytes_alg, "launches_per_step": len(reps[0]),
            "avg_launch_ms": conv_ms / len(reps[0]), "algorithmic_flops_per_step": flops, "conv_ms_per_step": conv_ms,
            "conv_share_of_step": conv_ms / ms_per_step,
        }
        if n_bx3 * 2 > len(reps[0]):
            # split-bf16 kernels: every fp32 multiply-add is six bf16 MFMA multiply-adds, so the fp32-equivalent roof
            # is the dense bf16 MFMA peak / 6
            # headline fraction = MATRIX-PIPE utilisation: bf16 MFMA FLOPs actually executed (six per fp32 multiply-add,
            # 2.25x fewer on the Winograd launches) over the dense bf16 peak.  Winograd's saving is NOT counted as
            # utilisation; the algorithmic view (direct-convolution FLOPs over the fp32-equivalent roof
            # 2500 / 6) is carried beside it.
            executed_tflops = mfma_per_product * flops_bx3 / (conv_ms * 1e-3) / 1e12
            out["roofline"] = dict({
                "bound": "mfma", "achieved": executed_tflops, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": executed_tflops / PEAK_BF16_MFMA_TFLOPS,
                "kernel": "conv_wino1d_kernel (1-D Winograd F(2,3) along x, filter rows as K) + conv_wino_bx3_wide_kernel (Winograd "
                          "F(2x2,3x3), 16-pixel stages) + conv_bx3_kernel<...> (direct): fp32 convolution as "
                          + ("3 x v_mfma_f32_32x32x16_f16 on two-piece fp16 operand splits" if ops.CONV_IMPL == "hx2" else
                             "6 x v_mfma_f32_32x32x16_bf16 on exact bf16x3 operand splits") + ", fp32 accumulate",
                "mfma_per_fp32_product": mfma_per_product,
                "achieved_note": "executed 16-bit MFMA FLOPs (mfma_per_fp32_product per fp32 multiply-add of the split; 1-D Winograd "
                                 "launches at 24/36, 2-D ones at 16/36 of the direct multiply-adds) / measured conv time, against the dense "
                                 "bf16 / fp16 MFMA peak (same rate)",
                "algorithmic": {"achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS / mfma_per_product,
                                "frac": achieved / (PEAK_BF16_MFMA_TFLOPS / mfma_per_product), "unit": "TFLOP/s",
                                "frac_of_round2_roof": achieved / (PEAK_BF16_MFMA_TFLOPS / 6.0),
                                "note": "direct-convolution fp32 FLOPs (2*MACs, SURVEY.md 8d: 209.59 GFLOP per image) / measured "
                                        "conv time over the fp32-equivalent roof = 16-bit MFMA peak / MFMAs per product; "
                                        "frac_of_round2_roof prices the same rate against round 2's roof (peak / 6)"},
                "bx3_launches": n_bx3, "winograd_launches": n_wino,
                "winograd_1d_launches": sum(1 for r in reps[0] if r.get("wino1d")),
                "fp32_mfma_peak_tflops": PEAK_FP32_MFMA_TFLOPS,
            }, **common)
        else:
            out["roofline"] = dict({
                "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                "kernel": "conv_mfma_kernel<...> + conv_wino_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit-GEMM / Winograd conv)",
                "achieved_note": "algorithmic FLOPs (2*MACs of the direct convolution) / measured conv time; the Winograd "
                                 "launches execute 2.25x fewer MFMA FLOPs, see executed_*",
                "winograd_launches": n_wino, "executed_mfma_tflops": executed / (conv_ms * 1e-3) / 1e12,
                "executed_mfma_frac": executed / (conv_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            }, **common)
        if world == 1 and not args.full and not args.no_alt and ops.split_impl():
            # the same iteration through the other kernel families, reported beside the headline so that the split result can
            # be judged against them: the exact-fp32-MFMA kernels (conv_mfma_kernel + fp32 Winograd) and the other split
            headline_impl = ops.CONV_IMPL
            notes = {"f32": ("alt_fp32_mfma", "IPDM_CONV_IMPL=f32: v_mfma_f32_32x32x2_f32 direct + fp32 Winograd kernels"),
                     "bx3": ("alt_bf16x3", "IPDM_CONV_IMPL=bx3: three bf16 pieces, six bf16 MFMAs per product (round 2's default)"),
                     "hx2": ("alt_f16x2", "IPDM_CONV_IMPL=hx2: two fp16 pieces, three fp16 MFMAs per product")}
            for impl in ("f32", "bx3", "hx2"):
                if impl == headline_impl:
                    continue
                try:
                    ops.CONV_IMPL = impl
                    alt = engine.IterationRunner(prob, seed=0, sample_offset=offset, use_graph=not args.no_graph)
                    for k in order[:3]:
                        alt.run(int(k))
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    n_alt = min(steps, 20)
                    for k in order[warm:warm + n_alt]:
                        alt.run(int(k))
                    torch.cuda.synchronize()
                    alt_ms = (time.perf_counter() - t1) * 1e3 / n_alt
                    out[notes[impl][0]] = {
                        "ms_per_step": alt_ms, "value": total / (alt_ms * 1e-3 * ITER_PER_RECON), "steps": n_alt,
                        "note": notes[impl][1] + ", same graph structure; not the headline",
                    }
                    del alt
                finally:
                    ops.CONV_IMPL = headline_impl
        if world == 1 and not args.full and not args.no_alt:
            out["upfirdn2d"] = upfirdn2d_hbm(dev)
        if world == 1 and not args.full and not args.no_full_batch and total != 105:
            # the WHOLE of BASELINE config 3 (105 posterior samples) on this one GPU: with the driver's N = 2, 4, 8 runs this
            # is the strong-scaling reference point of the 1 -> 8 curve (the headline `value` keeps the per-GPU share).
            # Same kernels, same dispatch (the rule is batch-independent), one hipGraph of the (210, 1, 128, 128) iteration.
            try:
                del runner
                torch.cuda.empty_cache()
                prob105 = engine.build_problem(dev, 105, R=40, H=128, W=128, num_sens=4, seed=0, scorenet=prob.scorenet,
                                               cfg=prob.cfg)
                r105 = engine.IterationRunner(prob105, seed=0, sample_offset=0, use_graph=not args.no_graph)
                n105 = 8
                o105 = (np.arange(2 + n105) * (n_sched // (2 + n105))) % n_sched
                for k in o105[:2]:
                    r105.run(int(k))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for k in o105[2:]:
                    r105.run(int(k))
                torch.cuda.synchronize()
                ms105 = (time.perf_counter() - t1) * 1e3 / n105
                out["config3_on_one_gpu"] = {
                    "samples": 105, "ms_per_step": ms105, "steps": n105, "value": 105 / (ms105 * 1e-3 * ITER_PER_RECON),
                    "unit": "reconstructions/s", "state_finite": bool(torch.isfinite(r105.x).all()),
                    "note": "all 105 samples of config 3 as ONE batch on one MI355X (score net on (210,1,128,128)); "
                            "value_formula as the headline",
                }
                del r105, prob105
                torch.cuda.empty_cache()
            except Exception as e:                              # never lose the headline line to the extra measurement
                out["config3_on_one_gpu"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, args.cpu_steps_b1, args.cpu_steps_b8, not args.no_cpu_mnist)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
