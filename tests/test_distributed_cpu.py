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
