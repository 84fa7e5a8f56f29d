"""World-size-2 rehearsal (gloo, CPU) of the multi-GPU path of bench.py: contiguous column shards,
one gather of the [cols, 12] flux blocks to rank 0, max-over-ranks timing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from grtcode_amd import multi


def test_shard_is_a_partition():
    for ncol, world in [(1800, 8), (100, 8), (7, 2), (3, 8), (16, 1)]:
        blocks = [multi.shard(ncol, r, world) for r in range(world)]
        cols = [c for first, n in blocks for c in range(first, first + n)]
        assert cols == list(range(ncol))
        assert max(n for _, n in blocks) == -(-ncol // world)
    assert multi.shard(1800, 3, 8) == (675, 225)        # 1 800 columns -> 225 per GPU


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cols_per_rank, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = multi.shard(world * cols_per_rank, rank, world)
    # stand-in for the device-resident [cols, 12] block a rank's pipeline produces
    local = torch.tensor([[1000.0 * c + k for k in range(12)] for c in range(first, first + count)], dtype=torch.float64)
    got = multi.gather_fluxes(local, rank, world)
    slowest = multi.max_over_ranks(0.5 + rank, torch.device("cpu"))
    dist.barrier()
    if rank == 0:
        np.save(result_path, np.concatenate([t.numpy() for t in got]))
        assert slowest == 0.5 + (world - 1)
    else:
        assert got is None
    dist.destroy_process_group()


def test_two_rank_gather_orders_columns(tmp_path):
    world, cols = 2, 5
    path = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), cols, path), nprocs=world, join=True)
    got = np.load(path)
    want = np.array([[1000.0 * c + k for k in range(12)] for c in range(world * cols)])
    assert np.array_equal(got, want)
