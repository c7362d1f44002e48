"""Posterior-sample sharding over the GPUs of one node (new functionality: the reference is
single-process, SURVEY.md 2 'Parallelism').  Samples are independent given (measurement, mask, coils,
weights), so the data path has NO collective: global sample ids are block-partitioned over ranks, each
sample's Langevin noise is keyed by its global id (results do not depend on the number of GPUs), and
the only exchange is one all-reduce(SUM) of seven moment planes at the end (RCCL over xGMI on GPUs, gloo in
the CPU tests)."""
import fcntl
import os

import torch
import torch.distributed as dist

# ---- ranks that SHARE one card (rehearsing N > 1 on a one-GPU box; never the production layout) -------------------------
# Measured on MI355X (scripts/shared_card_probe.hip, profiles/r03_shared_card_probe.txt): while the direct 3x3 split-operand
# convolution kernels of one HSA queue are resident, packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32, which hipcc
# emits freely) of waves of ANOTHER queue on the same compute units return wrong values in their last 16 lanes -- 2 streams of
# one process reproduce it in under a second, and so does a 30-line synthetic kernel (ds_read_b128 -> chained f16 MFMAs) in place of
# the convolution; scalar fp32 and v_pk_fma_f32 victims are never hit.  One process per GPU with one stream (the product layout: kernels of a stream never overlap) is unaffected.
# IPDM_DEVICE_TURNS=1 makes ranks that share a card take turns: a rank computes only while it holds an exclusive file lock
# and passes it on around every collective, so the rehearsal is bit-reproducible (tests/test_scripts_gpu.py).
_TURN = {"fd": None, "held": False}


def _take_turn():
    if _TURN["fd"] is not None and not _TURN["held"]:
        fcntl.flock(_TURN["fd"], fcntl.LOCK_EX)
        _TURN["held"] = True


def _pass_turn():
    if _TURN["fd"] is not None and _TURN["held"]:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        fcntl.flock(_TURN["fd"], fcntl.LOCK_UN)
        _TURN["held"] = False


def barrier(last=False):
    """last: the rank does no more GPU work afterwards (does not queue for another turn)"""
    _pass_turn()
    dist.barrier()
    if not last:
        _take_turn()


def all_reduce(t, op=None):
    """dist.all_reduce; ranks taking turns on a shared card pass the turn on while they wait in the collective"""
    _pass_turn()
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op is None else op)
    _take_turn()



def init_distributed():
    """-> (world, rank, device) from the launcher's environment (torchrun: WORLD_SIZE / RANK / LOCAL_RANK / MASTER_*), one
    process per GPU, backend "nccl" (= RCCL over xGMI).  Rehearsal knobs for a one-GPU box: IPDM_DIST_BACKEND=gloo (RCCL
    allows one rank per card) and IPDM_BENCH_DEVICE=<index> (every rank on that card).  World 1: no process group."""
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    dev_index = int(os.environ.get("IPDM_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("IPDM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    turns = os.environ.get("IPDM_DEVICE_TURNS", "0") == "1"
    if world > 1:
        _refuse_shared_cards(world, rank, dev_index, turns)
    if world > 1 and turns:
        start_turns()
    return world, rank, device


def _refuse_shared_cards(world, rank, dev_index, turns):
    """two ranks on one card WITHOUT the turn-taking give silently wrong bits (module header): raise on every rank instead.
    (host name, device index) of all ranks are exchanged once through the process group just created."""
    import socket
    mine = (socket.gethostname(), int(dev_index))
    seats = [None] * world
    dist.all_gather_object(seats, mine)
    shared = sorted({s for s in seats if seats.count(s) > 1})
    if shared and not turns:
        raise RuntimeError(
            f"ipdm sharding: ranks {[r for r, s in enumerate(seats) if s in shared]} resolve to the same card(s) {shared} "
            "(LOCAL_RANK collision or IPDM_BENCH_DEVICE). Two processes computing on one MI355X at once corrupt each other's "
            "packed-fp32 results (profiles/r03_shared_card_probe.txt). Give every rank its own GPU, or -- to rehearse on one card "
            "-- set IPDM_DEVICE_TURNS=1 so that the ranks take turns.")


def start_turns():
    """ranks of this job take turns from here on (the lock file is keyed by the rendezvous port); idempotent"""
    if _TURN["fd"] is None:
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ipdm_device_turn_{os.environ.get('MASTER_PORT', '0')}.lock")
        _TURN["fd"] = os.open(path, os.O_CREAT | os.O_RDWR, 0o600)
    _take_turn()


def gather_samples(local, total, world, rank):
    """every rank's block of samples -> the (total, ...) tensor in global sample order on every rank, bit for bit.
    RCCL: ONE all_gather_into_tensor of equal blocks (every rank pads its block to the largest shard: 105 -> 14 rows) -- each
    byte crosses a link once; config 4's (32, 24, 1, 128, 128) complex64 = 100 MB moves as 100 MB, not as an 8x larger
    all-reduce.  gloo (the CPU / one-card rehearsals; no all_gather_into_tensor for every build): one all-reduce(SUM) of a
    zero-filled buffer in which a rank fills only its own rows (adding zeros is exact, so the bits survive any reduction order).
    IPDM_GATHER=allgather|allreduce overrides the choice."""
    if world == 1:
        return local
    sizes = shard_sizes(total, world)
    lo, hi = shard_range(total, world, rank)
    real = torch.view_as_real(local) if local.is_complex() else local
    mode = os.environ.get("IPDM_GATHER") or ("allgather" if dist.get_backend() == "nccl" else "allreduce")
    if mode == "allgather":
        blk = max(sizes)
        mine = torch.zeros((blk,) + tuple(real.shape[1:]), dtype=real.dtype, device=real.device)
        mine[: hi - lo] = real[: hi - lo]
        gathered = torch.empty((world * blk,) + tuple(real.shape[1:]), dtype=real.dtype, device=real.device)
        _pass_turn()
        dist.all_gather_into_tensor(gathered, mine)
        _take_turn()
        buf = torch.cat([gathered[r * blk: r * blk + n] for r, n in enumerate(sizes)], dim=0)
    else:
        buf = torch.zeros((total,) + tuple(real.shape[1:]), dtype=real.dtype, device=real.device)
        buf[lo:hi] = real[: hi - lo]
        all_reduce(buf)
    return torch.view_as_complex(buf) if local.is_complex() else buf


def shard_sizes(total, world_size):
    """block partition, remainder to the lowest ranks: 105 over 8 -> [14, 13, 13, 13, 13, 13, 13, 13]"""
    base, rem = divmod(total, world_size)
    return [base + (1 if r < rem else 0) for r in range(world_size)]


def shard_range(total, world_size, rank):
    sizes = shard_sizes(total, world_size)
    start = sum(sizes[:rank])
    return start, start + sizes[rank]


def moment_planes(samples):
    """samples (n, 1, H, W) complex -> (7, 1, H, W) float64 partial sums:
    sum|x|, sum|x|^2, sum angle, sum angle^2, sum Re, sum Im, sum|angle|  (helpers/metrics.py:77-92 semantics: the
    reference's phase std is np.std(np.abs(angle)) -- its real-valued branch takes |.| before the std).
    GPU tensors: one kernel (ipdm_posterior_moments_c64); CPU tensors (the gloo tests): the same sums in torch."""
    if samples.shape[0] == 0:                                  # a rank without samples contributes zeros
        return torch.zeros((7,) + tuple(samples.shape[1:]), dtype=torch.float64, device=samples.device)
    if samples.is_cuda:
        from . import ops
        return ops.posterior_moment_planes(samples.to(torch.complex64).contiguous())
    # host tensors only reach this line from the CPU (gloo) tests of the collective; the product path is the kernel above
    mag, ph = samples.abs().float().double(), samples.angle().float().double()
    return torch.stack([mag.sum(0), (mag * mag).sum(0), ph.sum(0), (ph * ph).sum(0),
                        samples.real.double().sum(0), samples.imag.double().sum(0), ph.abs().sum(0)])


def posterior_from_moments(m, n):
    """-> dict(mag_mean, phase_mean, mag_std, phase_std, mean) float32 / complex64 (population std, as np.std)"""
    mag_mean, ph_mean = m[0] / n, m[2] / n
    return dict(mag_mean=mag_mean.float(), phase_mean=ph_mean.float(),
                mag_std=(m[1] / n - mag_mean ** 2).clamp_min(0).sqrt().float(),
                phase_std=(m[3] / n - (m[6] / n) ** 2).clamp_min(0).sqrt().float(),
                mean=torch.complex((m[4] / n).float(), (m[5] / n).float()))


def all_reduce_posterior(local_samples, total):
    """one all-reduce of 7 float64 planes (896 KiB at 128x128); works without an initialised process group (N=1)."""
    m = moment_planes(local_samples)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        all_reduce(m)
    return posterior_from_moments(m, total)
