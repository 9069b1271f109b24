"""CPU tier: the N>1 bookkeeping path with world_size 2 over gloo."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hades252_amd import sharding  # noqa: E402


def test_shard_ranges_partition():
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 1 << 20, (1 << 30) + 5):
            spans = [sharding.shard_range(r, world, n) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.weak_shard(3, 1 << 26) == (3 << 26, 4 << 26)
    with pytest.raises(ValueError):
        sharding.shard_range(2, 2, 10)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert sharding.env_world() == (rank, rank, world)
    sharding.init_process_group("gloo")
    sharding.barrier()
    mx = sharding.reduce_max(1.0 + rank)
    total = sharding.reduce_sum_int(10 + rank)
    assert sharding.reduce_min_int(1 if rank == 0 else 0) == 0 and sharding.reduce_min_int(1) == 1
    assert sharding.gather_floats(0.5 + rank) == [0.5, 1.5]
    # shard digests add up to the whole-range digest (how bench.py combines them)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    words = (np.arange(4000, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0xABCDEF)
    b, e = sharding.shard_range(rank, world, words.size // 20)
    mine = oracle_lib.digest_ref(words[20 * b:20 * e], 20 * b)
    comb = sharding.combine_digests(mine)
    q.put((rank, mx, total, comb, oracle_lib.digest_ref(words, 0)))
    dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mx, total, comb, whole in res:
        assert mx == 2.0
        assert total == 21
        assert comb == whole
