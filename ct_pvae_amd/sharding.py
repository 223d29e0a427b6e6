"""Batch sharding of the projector across the GPUs of one node.

The path shards over objects with no data-path collective: rank r projects objects [lo, hi) of the batch; angles,
tables and plans are replicated.  The only collectives are (1) the max-over-ranks of a timing scalar in bench.py and
(2) the single flat-bucket sum of the VAE gradients in the trainer (one all-reduce per step, ~3 MB: latency-bound, so
one bucket).  Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
import os

import torch
import torch.distributed as dist

__all__ = ["shard_range", "env_world", "init_from_env", "max_over_ranks", "allreduce_flat_", "gather_object_counts",
           "FlatGradBucket"]


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


class FlatGradBucket:
    """The gradients of a parameter list in ONE resident flat buffer (round 4): `fill(grads)` gathers them with one multi-tensor
    copy, `allreduce_()` is one collective on that memory -- no torch.cat of ~40 tensors, no per-tensor copy back (what
    allreduce_flat_ does every step: ~80 small launches around one 2.8 MB all-reduce) --, `views` are the per-parameter 1-D
    slices (the NaN filter and the per-tensor clip run on them) and `attach(params)` points every p.grad at its slice, so the
    optimiser reads the reduced, clipped gradients where they already are."""

    def __init__(self, like):
        like = list(like)
        if not like:
            raise ValueError("FlatGradBucket needs at least one tensor")
        self.shapes = [tuple(t.shape) for t in like]
        self.numels = [t.numel() for t in like]
        self.flat = torch.zeros(sum(self.numels), dtype=like[0].dtype, device=like[0].device)
        self.views = list(self.flat.split(self.numels))
        self.shaped = [v.view(sh) for v, sh in zip(self.views, self.shapes)]

    def matches(self, tensors):
        return len(tensors) == len(self.shapes) and all(tuple(t.shape) == sh for t, sh in zip(tensors, self.shapes)) and \
            tensors[0].device == self.flat.device and tensors[0].dtype == self.flat.dtype

    def fill(self, grads):
        torch._foreach_copy_(self.shaped, list(grads))
        return self

    def allreduce_(self, average=True):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if average:
                self.flat /= dist.get_world_size()
        return self

    def attach(self, params):
        for p, v in zip(params, self.shaped):
            p.grad = v
        return self
