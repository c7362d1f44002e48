"""RCCL on the GPU box: the one collective of the path (all-reduce of the seven float64 moment planes, sharding.py) through
the "nccl" (= RCCL) backend.  A one-GPU box can only form a world of one rank, so this covers what such a box can: backend
initialisation the way bench.py / the scripts do it (device_id, 127.0.0.1 rendezvous), the collective on device memory with
the production dtype and shape, the barrier and the teardown -- in a child process, so that the process group does not
outlive the test.  The N > 1 arithmetic is covered by the gloo tests (tests/test_distributed_cpu.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["IPDM_REPO"])
from inverseproblemwithdiffusionmodel_amd import sharding
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator().manual_seed(3)
x = torch.complex(torch.randn(5, 1, 128, 128, generator=g), torch.randn(5, 1, 128, 128, generator=g)).to(dev)
m = sharding.moment_planes(x)
assert m.dtype == torch.float64 and m.is_cuda and m.shape[0] == 7
before = m.clone()
dist.all_reduce(m, op=dist.ReduceOp.SUM)          # world of one: RCCL still runs the collective
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(m, before)
post = sharding.posterior_from_moments(m, 5)
ref = sharding.all_reduce_posterior(x, 5)
for k in post:
    assert torch.allclose(post[k], ref[k])
dist.destroy_process_group()
print("rccl ok", dist.is_nccl_available())
"""


def test_rccl_all_reduce_of_moment_planes():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IPDM_REPO=os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "rccl ok True" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
