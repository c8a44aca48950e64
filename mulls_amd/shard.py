"""Sharding of independent scan pairs over the GPUs of one node (one process per GPU, torch.distributed).

Scan pairs are independent: the data path has no collective.  Every rank registers its own block of
pairs; the only exchange is one gather of the small per-pair result records onto rank 0 at the end of a step
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import numpy as np

RECORD = 16 + 36 + 4  # T (16) + information matrix (36) + code, iters, sigma, confidence


def block_partition(n_items, world_size, rank):
    """Static block partition: pair p -> rank floor(p * world_size / n_items) (contiguous, sizes differ by at most 1)."""
    lo = (n_items * rank) // world_size
    hi = (n_items * (rank + 1)) // world_size
    return lo, hi


def pack_results(results, n, native=True):
    """ctypes Result array -> (n, RECORD) float64 table: by the library's host helper mulls_pack_results where libmulls_hip.so is built (one pass over the records:
    0.05 ms for 4096 of them), else — and with native=False, the statement the helper is tested against — by column slices of the raw struct bytes (0.5 ms)."""
    from . import abi

    if native and n:
        try:
            from . import lib as _lib

            out = np.empty((n, RECORD), np.float64)
            _lib.load().mulls_pack_results(results, n, out.ctypes.data)
            return out
        except Exception:
            pass  # (the library is not built on this box: the numpy statement below)

    rec = abi.C.sizeof(abi.Result)
    raw = np.frombuffer(results, dtype=np.uint8, count=n * rec).reshape(n, rec)

    def col(field, dtype, count):
        off = getattr(abi.Result, field).offset
        return np.ascontiguousarray(raw[:, off:off + count * np.dtype(dtype).itemsize]).view(dtype).reshape(n, count)

    out = np.empty((n, RECORD), np.float64)
    out[:, :16] = col("T", np.float64, 16)
    out[:, 16:52] = col("info", np.float64, 36)
    out[:, 52] = col("code", np.int32, 1)[:, 0]
    out[:, 53] = col("iters", np.int32, 1)[:, 0]
    out[:, 54] = col("sigma", np.float32, 1)[:, 0]
    out[:, 55] = col("confidence", np.float32, 1)[:, 0]
    return out


def gather_results(table, device=None, dst=0, counts=None):
    """Gather every rank's (n_r, RECORD) table on rank `dst`.  Returns the concatenated table there, None elsewhere.
    Without an initialised process group (single-GPU run) the table is returned unchanged.
    counts: the ranks' row counts when the caller knows them (block_partition makes them static): no size exchange, no host sync before the gather."""
    h = gather_post(table, device, dst, counts)
    return gather_wait(h)


def gather_post(table, device=None, dst=0, counts=None, force_collective=False):
    """Post the gather of one step's result table without waiting for it (the next step's kernels run meanwhile); gather_wait() completes it.
    force_collective: issue the collective even in a process group of one rank (tests: the RCCL call path on a single-GPU box)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return ("local", table)
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.from_numpy(np.ascontiguousarray(table))
    if device is not None:
        t = t.to(device, non_blocking=True)
    if counts is None:
        cl = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
        dist.all_gather(cl, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device))
        counts = [int(c.item()) for c in cl]
    pad = max(max(counts), 1)
    buf = torch.zeros((pad, RECORD), dtype=torch.float64, device=t.device)
    buf[: t.shape[0]] = t
    parts = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    work = dist.gather(buf, parts, dst=dst, async_op=True)
    return ("dist", work, parts, list(counts), buf)


def gather_wait(handle):
    if handle is None:
        return None
    if handle[0] == "local":
        return handle[1]
    _, work, parts, counts, _buf = handle
    work.wait()
    if parts is None:
        return None
    return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)])
