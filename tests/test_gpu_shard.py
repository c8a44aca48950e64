"""The result gather of bench.py on DEVICE tensors through RCCL (torch.distributed backend "nccl"): a one-GPU box can only form a process group of one rank, but
that is enough to execute the very calls the N-GPU run makes — init_process_group(nccl, device_id), the tensor staged to the device, dist.gather(async_op=True),
the wait and the copy back — and to compare what arrives with what was sent.  (The N > 1 logic — block partition, padding, static counts — runs on CPU ranks over
gloo in tests/test_shard.py and tests/test_bench_shard.py.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from mulls_amd import shard
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
rng = np.random.default_rng(3)
dev = torch.device("cuda", 0)
for n in (1, 128, 4096):
    table = rng.normal(size=(n, shard.RECORD))
    pending = None
    for step in range(3):  # posted while the "next step" would run, completed afterwards: bench.py's pattern
        h = shard.gather_post(table + step, device=dev, counts=[n], force_collective=True)
        if pending is not None:
            got = shard.gather_wait(pending[0])
            assert got.shape == (n, shard.RECORD) and np.array_equal(got, pending[1])
        pending = (h, table + step)
    assert np.array_equal(shard.gather_wait(pending[0]), pending[1])
    got = shard.gather_wait(shard.gather_post(table, device=dev, force_collective=True))  # without static counts: the size exchange runs too
    assert np.array_equal(got, table)
dist.barrier()
dist.destroy_process_group()
print("rccl gather ok")
""" % ROOT


def test_result_gather_runs_through_rccl_on_device_tensors():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29621", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "rccl gather ok" in p.stdout, p.stderr[-3000:]
