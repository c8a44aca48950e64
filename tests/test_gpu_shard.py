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


# ---------------------------------------------------------------------------------------------------------------------------------------------
# The C ABI's own forms of "independent pairs over several contexts" (round 5: mulls_icp_batch_sharded, mulls_pipe) — what a C++ caller of
# libmulls_hip.so gets without a second process.  One GPU on the box: the contexts share device 0, driven from separate host threads at once.
def _rows(res, n):
    import numpy as np

    return [(res[i].code, res[i].iters, tuple(res[i].ncorr), tuple(res[i].nsrc0), np.array(res[i].T[:]).tobytes(), np.array(res[i].info[:]).tobytes(),
             np.float32(res[i].sigma).tobytes(), np.float32(res[i].confidence).tobytes()) for i in range(n)]


def _mixed_pairs(pairs_small, reps):
    import numpy as np
    from conftest import planes_scene, transformed_copy
    from mulls_amd import abi, synth

    rng = np.random.default_rng(41)
    tgt = planes_scene(rng)
    far = abi.PairData(tgt, transformed_copy(tgt, synth.se3(200.0, 0, 0)))  # -2: too few correspondences
    empty = abi.PairData(tgt, [None] * 6)
    return ([p for p, _ in pairs_small] + [far, empty]) * reps


def test_c_abi_sharding_over_contexts_driven_by_threads(pairs_small):
    """mulls_icp_batch_sharded: the pair list block-partitioned over 1, 2, 3 and more contexts than pairs — every context on device 0, every shard driven by
    its own host thread at the same time — returns, pair by pair, the bits of the serial mulls_icp_batch on one context (healthy, failing and empty pairs)."""
    from mulls_amd import abi, lib, shard

    plist = _mixed_pairs(pairs_small, 5)  # 25 pairs
    P = abi.kitti_params(dis_thre_unit=2.4)
    ctxs = [lib.Context(0) for _ in range(4)]
    want = _rows(ctxs[0].icp_batch(plist, P), len(plist))
    assert {r[0] for r in want} >= {1, -2}
    for n_ctx in (1, 2, 3, 4):
        for _ in range(2):  # (the contexts' cached batches are reused from the second call on)
            got = lib.icp_batch_sharded(ctxs[:n_ctx], plist, P)
            assert _rows(got, len(plist)) == want, n_ctx
    few = plist[:3]
    assert _rows(lib.icp_batch_sharded(ctxs, few, P), 3) == want[:3]  # more contexts than pairs: the empty shards idle
    # the partition is the one of mulls_amd/shard.py (what bench.py --total-pairs gives each rank)
    bounds = [shard.block_partition(len(plist), 3, r) for r in range(3)]
    assert bounds[0][0] == 0 and bounds[-1][1] == len(plist) and all(bounds[r][1] == bounds[r + 1][0] for r in range(2))
    # a context named twice is refused (one host thread at a time per context), nothing runs
    with pytest.raises(lib.MullsError):
        lib.icp_batch_sharded([ctxs[0], ctxs[0]], plist, P)
    for c in ctxs:
        c.close()


def test_pipelined_calls_from_host_buffers(pairs_small):
    """mulls_pipe: calls begun back to back on alternating contexts (the second one's staging runs under the first one's kernels), tickets ended out of
    order, a third call that has to wait for its lane, different pair lists and parameters per call — each call's results are the serial call's bits."""
    from mulls_amd import abi, lib

    lists = [_mixed_pairs(pairs_small, 4), _mixed_pairs(pairs_small, 2)[::-1], [p for p, _ in pairs_small] * 6, _mixed_pairs(pairs_small, 1)]
    params = [abi.kitti_params(dis_thre_unit=2.4), abi.default_params(), abi.kitti_params(dis_thre_unit=2.4, max_iter_num=3), abi.default_params(used_feature_type="111111", faithful=0)]
    ctx = lib.Context(0)
    want = [_rows(ctx.icp_batch(pl, P), len(pl)) for pl, P in zip(lists, params)]
    ctx.close()
    for depth in (2, 3):
        pipe = lib.Pipe(0, depth)
        pipe.set_option(abi.OPT_STAGGER, 4096)  # (an option goes to every lane)
        for _ in range(2):
            tickets = [pipe.begin(pl, P) for pl, P in zip(lists, params)]  # four calls on `depth` lanes: the later ones wait for their lane
            assert tickets == sorted(tickets) and len(set(tickets)) == 4
            order = [3, 2] if depth == 2 else [3, 1, 2]  # a lane's earlier ticket can no longer be waited for once the lane got its next call
            for k in order:
                res = pipe.end(tickets[k])
                assert _rows(res, len(lists[k])) == want[k], (depth, k)
        # tickets in pairs, the way a caller with a stream of requests uses it: begin(k + 1), end(k)
        pending = None
        for rep in range(6):
            k = rep % len(lists)
            t = pipe.begin(lists[k], params[k])
            if pending is not None:
                assert _rows(pipe.end(pending[0]), len(lists[pending[1]])) == want[pending[1]]
            pending = (t, k)
        assert _rows(pipe.end(pending[0]), len(lists[pending[1]])) == want[pending[1]]
        pipe.close()
