"""Multi-process (gloo, world_size 2, CPU) test of the pair sharding + result gather used by bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_partition_covers_everything():
    from mulls_amd import shard

    for n in (0, 1, 7, 8, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [shard.block_partition(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from mulls_amd import abi, shard, synth
    from oracle import pyoracle  # the CPU test has no GPU: the oracle stands in for the per-rank compute

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.block_partition(n_total, world, rank)
    P = abi.kitti_params(dis_thre_unit=2.4, max_iter_num=3)
    src = {abi.GROUND: 200, abi.PILLAR: 100, abi.FACADE: 250}
    tgt = {abi.GROUND: 600, abi.PILLAR: 300, abi.FACADE: 700}
    res = abi.make_result_array(max(hi - lo, 1))
    for k, p in enumerate(range(lo, hi)):
        pair, _ = synth.make_pair(500 + p, n_beams=12, n_az=300, src_counts=src, tgt_counts=tgt, vertex_count=0)
        r = pyoracle.icp(pair, P)[0]
        for f in ("code", "iters", "sigma", "confidence"):
            setattr(res[k], f, getattr(r, f))
        res[k].T[:] = r.T[:]
        res[k].info[:] = r.info[:]
    table = shard.pack_results(res, hi - lo)
    full = shard.gather_results(table)
    if rank == 0:
        q.put(full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    n_total = 5  # uneven split: 2 + 3
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert full.shape == (n_total, 56)

    sys.path.insert(0, ROOT)
    from mulls_amd import abi, synth
    from oracle import pyoracle

    P = abi.kitti_params(dis_thre_unit=2.4, max_iter_num=3)
    src = {abi.GROUND: 200, abi.PILLAR: 100, abi.FACADE: 250}
    tgt = {abi.GROUND: 600, abi.PILLAR: 300, abi.FACADE: 700}
    for p in range(n_total):
        pair, _ = synth.make_pair(500 + p, n_beams=12, n_az=300, src_counts=src, tgt_counts=tgt, vertex_count=0)
        r = pyoracle.icp(pair, P)[0]
        assert np.array_equal(full[p, :16], np.array(r.T[:]))
        assert full[p, 52] == r.code and full[p, 53] == r.iters


def test_pack_results_matches_field_access():
    """The vectorised packing reads the same bytes as per-record ctypes access."""
    from mulls_amd import abi, shard

    rng = np.random.default_rng(5)
    n = 7
    res = abi.make_result_array(n)
    for i in range(n):
        res[i].code, res[i].iters = int(rng.integers(-3, 2)), int(rng.integers(0, 21))
        res[i].sigma, res[i].confidence = float(rng.random()), float(rng.random())
        for k in range(16):
            res[i].T[k] = rng.normal()
        for k in range(36):
            res[i].info[k] = rng.normal()
    tab = shard.pack_results(res, n)
    for i in range(n):
        assert list(tab[i, :16]) == list(res[i].T[:]) and list(tab[i, 16:52]) == list(res[i].info[:])
        assert tab[i, 52] == res[i].code and tab[i, 53] == res[i].iters
        assert tab[i, 54] == res[i].sigma and tab[i, 55] == res[i].confidence


def test_native_pack_equals_the_numpy_statement():
    """mulls_pack_results (the library's host helper behind shard.pack_results) writes the very table the column slices of the raw records give."""
    import os

    import pytest
    from mulls_amd import abi, lib as mlib, shard

    if not os.path.exists(mlib.LIB_PATH):
        pytest.skip("libmulls_hip.so is not built")
    rng = np.random.default_rng(9)
    for n in (1, 7, 300):
        res = abi.make_result_array(n)
        for i in range(n):
            for k in range(16):
                res[i].T[k] = rng.normal()
            for k in range(36):
                res[i].info[k] = rng.normal()
            res[i].code, res[i].iters = int(rng.integers(-3, 2)), int(rng.integers(0, 41))
            res[i].sigma, res[i].confidence = float(rng.normal()), float(rng.uniform())
        assert np.array_equal(shard.pack_results(res, n, native=True), shard.pack_results(res, n, native=False))
