"""Posterior-sample sharding over the GPUs of one node (new functionality: the reference is
single-process, SURVEY.md 2 'Parallelism').  Samples are independent given (measurement, mask, coils,
weights), so the data path has NO collective: global sample ids are block-partitioned over ranks, each
sample's Langevin noise is keyed by its global id (results do not depend on the number of GPUs), and
the only exchange is one all-reduce(SUM) of seven moment planes at the end (RCCL over xGMI on GPUs, gloo in
the CPU tests)."""
import torch
import torch.distributed as dist


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
    if samples.is_cuda:
        from . import ops
        return ops.posterior_moment_planes(samples.to(torch.complex64).contiguous())
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
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
    return posterior_from_moments(m, total)
