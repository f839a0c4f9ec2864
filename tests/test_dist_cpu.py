"""The N > 1 path on CPU: two processes over gloo exercise the sharding, per-trial seeding and the rank-ordered
batch-statistics reduction that bench.py uses over RCCL on the GPU node."""
import os
import socket

import numpy as np
import pytest

from nuslam_hip import dist as nd


def test_shard_partitions_every_filter_once():
    for total in (1, 7, 1024, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = nd.shard(total, world, r)
                seen += list(range(s, s + c))
            assert seen == list(range(total))
    assert nd.shard(1024, 8, 3) == (384, 128)


def test_replica_seed_is_world_size_independent():
    for world in (1, 2, 4, 8):
        seeds = []
        for r in range(world):
            s, c = nd.shard(16, world, r)
            seeds += [nd.replica_seed(12345, s + i) for i in range(c)]
        assert seeds == [12345 + i for i in range(16)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_filters, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = nd.shard(total_filters, world, rank)
    L = 9
    # stand-in for Batch.stats(): per-filter "states" derived from the GLOBAL filter index only
    rows = np.stack([np.random.default_rng(nd.replica_seed(7, start + i)).normal(size=L) for i in range(count)])
    local = np.concatenate([rows.sum(0), (rows ** 2).sum(0), [rows[:, 0].sum()], [count]])
    total, per_rank = nd.reduce_stats(local)
    tmax = nd.max_over_ranks(0.5 + rank)
    q.put((rank, total, per_rank, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduction_matches_single_process():
    import torch.multiprocessing as mp
    world, total_filters, L = 2, 10, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_filters, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda x: x[0])
    # every rank holds the same, rank-ordered total
    assert np.array_equal(res[0][1], res[1][1])
    assert res[0][3] == res[1][3] == 1.5
    # and it equals the sum formed from the per-rank rows in rank order (what a 1-process run would gather)
    per_rank = res[0][2]
    assert np.array_equal(per_rank[0] + per_rank[1], res[0][1])
    assert per_rank[:, -1].tolist() == [5.0, 5.0] and res[0][1][-1] == total_filters
    # the statistics themselves match a single-process evaluation of all ten trials to rounding
    rows = np.stack([np.random.default_rng(nd.replica_seed(7, i)).normal(size=L) for i in range(total_filters)])
    ref = np.concatenate([rows.sum(0), (rows ** 2).sum(0), [rows[:, 0].sum()], [total_filters]])
    assert np.allclose(res[0][1], ref, rtol=1e-13, atol=1e-13)
