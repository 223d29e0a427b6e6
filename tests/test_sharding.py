"""Multi-process (world_size 2, gloo, CPU) tests of the batch-sharding helpers the N>1 bench and the trainer use.
The projector itself has no collective; here the oracle stands in for the per-rank compute (tests may use it)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ct_pvae_amd import sharding


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 5, 50, 400, 401):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_range(400, 3, 8) == (150, 200)     # config 4: 50 objects per GPU
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    w, r, _ = sharding.init_from_env(backend="gloo")
    assert (w, r) == (world, rank)
    # (1) timing aggregation: max over ranks; object count: sum over ranks
    assert sharding.max_over_ranks(1.0 + rank) == float(world)
    lo, hi = sharding.shard_range(7, rank, world)
    assert sharding.gather_object_counts(hi - lo) == 7
    # (2) one flat-bucket gradient all-reduce == sum / mean of the per-rank gradients
    rng = np.random.default_rng(0)
    full = [rng.standard_normal(s).astype(np.float32) for s in ((3, 4), (5,), (2, 2, 2))]
    mine = [torch.from_numpy(f * (rank + 1)) for f in full]
    sharding.allreduce_flat_(mine, average=False)
    scale = sum(range(1, world + 1))
    for got, f in zip(mine, full):
        np.testing.assert_allclose(got.numpy(), f * scale, rtol=1e-6)
    mine = [torch.from_numpy(f * (rank + 1)) for f in full]
    sharding.allreduce_flat_(mine, average=True)
    np.testing.assert_allclose(mine[0].numpy(), full[0] * scale / world, rtol=1e-6)
    # (2b) round 4: the resident flat bucket gives the bits of the cat / copy-back path, step after step, and leaves the reduced
    # gradients where the optimiser reads them (p.grad = a slice of the bucket)
    params = [torch.nn.Parameter(torch.zeros(f.shape)) for f in full]
    bucket = None
    for step in range(3):
        grads = [torch.from_numpy(f * (rank + 1 + step)) for f in full]
        ref = [g.clone() for g in grads]
        sharding.allreduce_flat_(ref, average=False)
        if bucket is None:
            bucket = sharding.FlatGradBucket(grads)
        assert bucket.matches(grads)
        bucket.fill(grads).allreduce_(average=False).attach(params)
        for p_, r_ in zip(params, ref):
            assert torch.equal(p_.grad, r_) and p_.grad.data_ptr() >= bucket.flat.data_ptr()
    assert not bucket.matches(grads[:2])
    # (3) sharded projection == unsharded projection, slice for slice (no collective on the data path)
    from oracle import radon_oracle as orc
    imgs = np.random.default_rng(1).random((5, 12, 12), dtype=np.float32)
    theta = np.linspace(0, np.pi, 6, endpoint=False)
    geom = orc.Geometry(12, 12, True)
    T = orc.rotate_transforms(theta, geom.PH, geom.PW)
    lo, hi = sharding.shard_range(5, rank, world)
    part = orc.rotate_fwd(imgs[lo:hi], geom, T, 0)
    np.save(os.path.join(tmp, f"part{rank}.npy"), part)
    dist.barrier()
    if rank == 0:
        whole = orc.rotate_fwd(imgs, geom, T, 0)
        parts = np.concatenate([np.load(os.path.join(tmp, f"part{r}.npy")) for r in range(world)])
        np.testing.assert_array_equal(parts, whole)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
