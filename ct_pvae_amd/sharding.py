"""Batch sharding of the projector across the GPUs of one node.

The path shards over objects with no data-path collective: rank r projects objects [lo, hi) of the batch; angles,
tables and plans are replicated.  The only collectives are (1) the max-over-ranks of a timing scalar in bench.py and
(2) the single flat-bucket sum of the VAE gradients in the trainer (one all-reduce per step, ~3 MB: latency-bound, so
one bucket).  Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
import os

import torch
import torch.distributed as dist

__all__ = ["shard_range", "env_world", "init_from_env", "max_over_ranks", "allreduce_flat_", "gather_object_counts"]


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi): the first n_items % world ranks take one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(int(n_items), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_world():
    """(world, rank, local_rank) from the torch.distributed.run environment (1, 0, 0 when launched plainly)."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_from_env(backend=None):
    """One process per GPU: bind the device and join the default group.  Returns (world, rank, device index).

    CTPVAE_REHEARSE_ONE_GPU=1 rehearses a multi-rank launch on a box with ONE GPU: every rank binds device 0 and the
    group runs over gloo (RCCL needs one device per rank).  Same code path otherwise; used by the tests only."""
    world, rank, local = env_world()
    use_cuda = torch.cuda.is_available()
    rehearse = os.environ.get("CTPVAE_REHEARSE_ONE_GPU", "") == "1"
    if rehearse:
        local, backend = 0, "gloo"
    if use_cuda:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if use_cuda else "gloo")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return world, rank, local


def _scratch_device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def max_over_ranks(value):
    """Max of a python float over all ranks (identity for a single process)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_scratch_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_object_counts(n_local):
    """Sum over ranks of the number of objects each processed."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return int(n_local)
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=_scratch_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def allreduce_flat_(tensors, average=True):
    """Sum (or mean) a list of same-dtype tensors across ranks IN PLACE with ONE all-reduce of one flat bucket."""
    if not tensors or not dist.is_initialized() or dist.get_world_size() == 1:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return tensors
