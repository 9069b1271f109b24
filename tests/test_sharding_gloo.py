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


# ---------------------------------------------------------------------------------------------------------------
# The Merkle path's ONE exchange step: every rank builds whole sub-trees, an all_gather moves the
# 32-byte sub-roots, every rank hashes the top levels.  hades252_amd/merkle.py end to end over gloo, world sizes 2 and 4,
# with the per-device tree (`strategy.merkle4_root`, a HIP call) replaced by the ORACLE's tree -- tests may use the
# oracle; what is under test is the split, the gather order and the top levels.
# ---------------------------------------------------------------------------------------------------------------
def _oracle_merkle4_root(leaves_t, tag_mont, out_idx=1, scratch=None):
    import numpy as np
    import oracle_lib
    orc = oracle_lib.load()
    flat = leaves_t.contiguous().view(-1).numpy().view(np.uint64)
    root = orc.merkle4_root(flat, tag_mont, out_idx)
    return torch.from_numpy(np.ascontiguousarray(root).view(np.int64).copy())


def _merkle_worker(rank, world, port, n_leaves, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_lib
    import hades_spec as S
    from hades252_amd import merkle, strategy
    strategy.merkle4_root = _oracle_merkle4_root          # the module attribute merkle.py calls (H.merkle4_root)
    sharding.init_process_group("gloo")
    assert sharding.world_size() == world and sharding.backend_name() == "gloo"
    orc = oracle_lib.load()
    tag = S.to_mont(15)
    per = n_leaves // world
    shard = torch.from_numpy(orc.gen_b(rank * per, per).view(np.int64).copy()).view(-1, 4)
    root = merkle.merkle4_root_sharded(shard, n_leaves, tag, 1)
    devs = sharding.gather_strings("rank %d of %d; pci 0000:%02x:00.0" % (rank, world, rank))
    q.put((rank, oracle_lib.int_of(root.numpy().view(np.uint64)), devs))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])          # 8 = the production split: 16 sub-trees, two per rank
def test_merkle_root_sharded_over_gloo(world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib
    import hades_spec as S
    n_leaves = 4 ** 6
    orc = oracle_lib.load()
    want = oracle_lib.int_of(orc.merkle4_root(orc.gen_b(0, n_leaves), S.to_mont(15), 1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() + 7 * world) % 300
    procs = [ctx.Process(target=_merkle_worker, args=(r, world, port, n_leaves, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in res) == list(range(world))
    for rank, root, devs in res:
        assert root == want, "rank %d: sharded root differs from the oracle's tree" % rank     # EVERY rank holds the root
        assert devs == ["rank %d of %d; pci 0000:%02x:00.0" % (r, world, r) for r in range(world)]      # rank order
        assert sharding.distinct_devices(devs)


def test_subtree_split_against_the_committed_sub_roots():
    """`subtree_split` for world sizes 1 .. 16 on BASELINE configs[3]'s 2^24-leaf tree, the split arithmetic checked on the
    oracle's committed nodes (tests/golden/kat.json merkle4_full_size: the root and the 16 nodes two levels below it): for
    every valid world size the ranks' sub-roots ARE a level of the committed tree, in rank order, and hashing them down
    with the oracle gives the committed root; the other world sizes are refused."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_lib
    import hades_spec as S
    from hades252_amd import merkle
    orc = oracle_lib.load()
    tag = S.to_mont(15)
    n = 1 << 24
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        gold = json.load(f)["merkle4_full_size"][str(n)]
    sub16 = np.array([l for h in gold["sub_roots_16"] for l in oracle_lib.limbs_of(int(h, 16))], dtype=np.uint64)
    level4 = orc.merkle4_level(sub16, tag, 1)                      # the 4 nodes one level below the root
    assert oracle_lib.int_of(orc.merkle4_level(level4, tag, 1)) == int(gold["root"], 16)
    levels = {1: None, 4: level4, 16: sub16}
    for world in range(1, 17):
        if world & (world - 1):
            with pytest.raises(ValueError):
                merkle.subtree_split(n, world)
            continue
        per_sub, subs = merkle.subtree_split(n, world)
        n_sub = world * subs
        assert n_sub in (1, 4, 16) and per_sub * n_sub == n and per_sub * subs == n // world
        # rank g owns sub-trees [g * subs, (g + 1) * subs) = leaves [g n / W, (g + 1) n / W): contiguous, in rank order
        for g in range(world):
            first_leaf = g * subs * per_sub
            assert first_leaf == g * (n // world)
        if n_sub > 1:
            # what the all_gather concatenates (rank order) is that level of the committed tree; finishing it gives the root
            gathered = np.concatenate([levels[n_sub][4 * g * subs:4 * (g + 1) * subs] for g in range(world)])
            assert (gathered == levels[n_sub]).all()
            assert oracle_lib.int_of(orc.merkle4_root(gathered, tag, 1)) == int(gold["root"], 16)
    with pytest.raises(ValueError):
        merkle.subtree_split(4 ** 2, 32)          # tree too small for the world size
    with pytest.raises(ValueError):
        merkle.subtree_split(3 * 4 ** 5, 2)       # not a power of 4


def test_device_identity_helpers():
    assert sharding.distinct_devices(["pci 0000:05:00.0; uuid a; AMD", "pci 0000:15:00.0; uuid b; AMD"])
    assert not sharding.distinct_devices(["pci 0000:05:00.0; AMD", "pci 0000:05:00.0; AMD"])
    assert not sharding.distinct_devices(["unknown device 0", "unknown device 0"])
    assert sharding.distinct_devices(["only one"])
    assert sharding.gather_strings("solo") == ["solo"] and sharding.world_size() == 1 and sharding.backend_name() == "none"
