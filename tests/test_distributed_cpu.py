"""CPU, world size 2, gloo: the bucketed gradient reducer of the data-parallel path
(multimodalanalytical_amd/trainer.py) sums the flat gradient buffer exactly, in back-to-front
buckets, whatever the order/granularity of the `ready()` marks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, marks, n, bucket):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodalanalytical_amd.trainer import BucketedReducer
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    red = BucketedReducer(flat, bucket_elems=bucket)
    for step in range(2):                       # two optimiser steps reuse the reducer
        if step:
            flat.copy_(mine)
        red.reset()
        for lo in marks:
            red.ready(lo)
        red.finish()
        expect = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
        assert torch.allclose(flat, expect, atol=1e-6), (rank, step, float((flat - expect).abs().max()))
        spans = sorted(red.launched)
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))      # a partition of the buffer
        assert all(hi - lo == bucket for lo, hi in spans[1:])           # whole buckets from the back
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_reducer_world2_gloo():
    n, bucket = 10_000, 1024
    marks = [9000, 8999, 5000, 4990, 100]
    mp.spawn(_worker, args=(2, _free_port(), marks, n, bucket), nprocs=2, join=True)


def test_bucketed_reducer_single_flush():
    mp.spawn(_worker, args=(2, _free_port(), [], 3000, 1 << 20), nprocs=2, join=True)


def _sync_worker(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodalanalytical_amd.trainer import sync_mean
    got = sync_mean(torch.tensor(float(rank + 1)))          # Lightning sync_dist=True: mean over ranks
    assert abs(float(got) - sum(range(1, world + 1)) / world) < 1e-6
    got = sync_mean(2.0 * (rank + 1))                       # plain floats are accepted too
    assert abs(float(got) - 2.0 * sum(range(1, world + 1)) / world) < 1e-6
    dist.barrier()
    dist.destroy_process_group()


def test_sync_dist_scalar_mean_world2_gloo():
    mp.spawn(_sync_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_mixture_stream_is_rank_sharded():
    """the iterable mixture dataset needs rank-strided sharding (the reference has none): ranks draw the same
    index stream and keep disjoint, equally long row sets whose union is the (evenly divisible part of the) round."""
    import numpy as np
    from multimodalanalytical_amd.preprocess import mix_indices, shard_rows
    cfg = {"n_compounds": 2, "parallel_samples": 64, "train_max_n_samples": 256}
    for world in (2, 4, 8):
        rounds = list(mix_indices(50, cfg, "train", seed=3247))
        assert len(rounds) >= 2
        for ri in rounds:
            parts = [shard_rows(ri, r, world) for r in range(world)]
            assert len({len(p) for p in parts}) == 1
            allrows = np.concatenate(parts)
            assert len(allrows) == (len(ri) // world) * world
            assert len(np.unique(allrows, axis=0)) == len(allrows)          # disjoint
            assert set(map(tuple, allrows)) <= set(map(tuple, ri))
    assert shard_rows(rounds[0], 0, 1) is rounds[0]


def _flag_worker(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodalanalytical_amd.trainer import sync_flag
    # only rank 0 knows the decision (it ranks the checkpoints): every rank must leave the epoch loop in the same epoch
    stale, left_at = 0, None
    for epoch in range(6):
        if rank == 0:
            stale = stale + 1 if epoch >= 2 else 0
        if sync_flag(rank == 0 and stale >= 2):
            left_at = epoch
            break
        dist.barrier()          # stands for the next epoch's collectives: a rank that stayed behind would hang here
    assert left_at == 3, (rank, left_at)
    dist.barrier()
    dist.destroy_process_group()


def test_early_stopping_flag_reaches_every_rank_world2_gloo():
    mp.spawn(_flag_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_bucket_size_follows_the_gradient_buffer():
    from multimodalanalytical_amd.trainer import bucket_elems_for
    assert bucket_elems_for(47_000_000) * 8 >= 47_000_000 > bucket_elems_for(47_000_000) * 7      # c2: about eight buckets
    assert bucket_elems_for(100) == 1 << 20                                                        # tiny models: one bucket
