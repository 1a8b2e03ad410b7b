"""CPU, world_size 2 over gloo: the multi-GPU path's only distributed pieces -- the sample partition
and the (sum units, max time) reduction -- behave; every sample is genotyped exactly once."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tredparse_amd import shard


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 8, 1000, 8001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                a, b = shard.shard_range(n, r, world)
                assert 0 <= a <= b <= n
                seen += list(range(a, b))
                for s in range(a, b):
                    assert shard.sample_owner(s, n, world) == r
            assert seen == list(range(n))
            sizes = [shard.shard_range(n, r, world) for r in range(world)]
            assert max(b - a for a, b in sizes) - min(b - a for a, b in sizes) <= 1


def _worker(rank, world, port, n_samples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard.shard_range(n_samples, rank, world)
    # each rank "genotypes" its own samples x 30 loci; rank 1 is slower
    units_local = (b - a) * 30
    elapsed_local = 0.5 + 0.25 * rank
    dist.barrier()
    units, elapsed = shard.aggregate(units_local, elapsed_local, dist)
    owned = torch.zeros(n_samples, dtype=torch.int32)
    owned[a:b] = 1
    dist.all_reduce(owned)
    q.put((rank, units, elapsed, owned.numpy().tolist()))
    dist.destroy_process_group()


def test_two_rank_aggregate_over_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_samples = 11
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_samples, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, units, elapsed, owned in res:
        assert units == n_samples * 30           # every unit counted once across ranks
        assert elapsed == pytest.approx(0.75)    # max over ranks
        assert owned == [1] * n_samples          # every sample owned by exactly one rank


def test_balanced_owners_spread_unequal_samples():
    """--gpus N assigns samples by cost (BAM size): longest first, each to the least loaded rank; every sample has one
    owner, equal costs degrade to round-robin, and one huge sample does not drag a block of others with it."""
    assert shard.balanced_owners([5, 5, 5, 5, 5, 5], 3) == [0, 1, 2, 0, 1, 2]
    owners = shard.balanced_owners([100, 1, 1, 1, 1, 1, 1, 1], 2)
    assert owners[0] == 0 and owners[1:] == [1] * 7                   # the block partition would give rank 0 100 + 3
    rng = np.random.default_rng(5)
    costs = rng.integers(1, 1000, 200).tolist()
    owners = shard.balanced_owners(costs, 8)
    load = [sum(c for c, o in zip(costs, owners) if o == r) for r in range(8)]
    assert sorted(set(owners)) == list(range(8)) and max(load) - min(load) <= max(costs)
    assert max(load) <= 1.02 * sum(costs) / 8 + 1
    assert shard.balanced_owners([], 4) == [] and shard.balanced_owners([0, 0, 0], 2) == [0, 1, 0]
