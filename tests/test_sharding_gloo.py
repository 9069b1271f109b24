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


# ---------------------------------------------------------------------------------------------------------------
# Strong scaling (bench.py --total-perms / secondary.config5_2p30 at every N): T states in all, rank g owns
# [g T / N, (g + 1) T / N).  World size 8 over gloo with the ORACLE in place of the device call: the shards tile the
# range, every rank's digest of its outputs (global word indices) sums to the digest of the whole batch, and the
# pre-collective agreement (`ranks_agree`) makes all ranks skip together when one of them fails alone.
# ---------------------------------------------------------------------------------------------------------------
def _strong_worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import bench
    sharding.init_process_group("gloo")
    orc = oracle_lib.load()
    b, e = sharding.strong_shard(rank, world, total)
    out = orc.perm_batch(orc.gen_b(5 * b, 5 * (e - b)), 1)
    mine = oracle_lib.digest_ref(out, 20 * b)
    comb = sharding.combine_digests(mine)
    sizes = sharding.gather_floats(float(e - b))
    # one rank fails its local set-up: every rank must learn it BEFORE the record's first collective
    agree_all = bench.ranks_agree(sharding, True, "cpu")
    agree_one_down = bench.ranks_agree(sharding, rank != 5, "cpu")
    q.put((rank, (b, e), comb, sizes, agree_all, agree_one_down))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(8, 8 * 96), (8, 1001), (2, 640)])
def test_strong_scaling_shards_over_gloo(world, total):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.load()
    whole = oracle_lib.digest_ref(orc.perm_batch(orc.gen_b(0, 5 * total)), 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() + 13 * world + total) % 250
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    spans = [r[1] for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for rank, span, comb, sizes, agree_all, agree_one_down in res:
        assert comb == whole                                           # every rank holds the whole job's digest
        assert sizes == [float(e - b) for b, e in spans] and max(sizes) - min(sizes) <= 1
        assert agree_all is True and agree_one_down is (world <= 5)    # rank 5 exists only at world 8


def test_committed_oracle_digests_cover_every_strong_shard():
    """bench.golden_range_digest: the digest of ALL outputs of perm(generator-B states [b, b + n)) as a sum of committed
    oracle pieces.  For BASELINE configs[4] (2^30 states) at N = 1, 2, 4, 8 every rank's range is covered, the per-rank
    digests add up to the oracle's digest of the whole batch, and the two independent oracle runs (2^26-state blocks of the
    headline, 2^27-state shards of configs[4]) agree wherever both cover a range."""
    import json
    import bench
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        kat = json.load(f)
    total = 1 << 30
    whole = [int(h, 16) for h in kat["config5_2p30"]["oracle_digest"]]
    for world in (1, 2, 4, 8):
        acc = [0, 0, 0, 0]
        for rank in range(world):
            b, e = sharding.strong_shard(rank, world, total)
            d = bench.golden_range_digest(b, e - b)
            assert d is not None, (world, rank)
            acc = [(a + int(h, 16)) & sharding.M64 for a, h in zip(acc, d)]
        assert acc == whole, world
    assert bench.golden_range_digest(7 << 27, 1 << 27) == kat["config5_2p30"]["oracle_shard_digests"][7]
    # weak-scaling headline: rank g's block g; blocks 2k, 2k + 1 (one oracle run) == shard k (another oracle run)
    for g in range(8):
        assert bench.golden_range_digest(g << 26, 1 << 26) == kat["headline_2p26_blocks"]["blocks"][str(g)]
    for k in range(4):
        two = [(int(a, 16) + int(b, 16)) & sharding.M64 for a, b in zip(bench.golden_range_digest((2 * k) << 26, 1 << 26),
                                                                        bench.golden_range_digest((2 * k + 1) << 26, 1 << 26))]
        assert ["%016x" % x for x in two] == kat["config5_2p30"]["oracle_shard_digests"][k]
        assert bench.golden_range_digest(k << 27, 1 << 27) == kat["config5_2p30"]["oracle_shard_digests"][k]
    # not covered: unaligned, beyond the committed range, a world size that does not divide 8, empty
    for b, n in ((1, 1 << 26), (8 << 26, 1 << 26), (0, (1 << 30) // 3), (0, 0), (1 << 30, 1 << 27), (0, 1 << 20)):
        assert bench.golden_range_digest(b, n) is None, (b, n)


def test_lone_rank_picks_a_free_port_and_a_job_must_be_told_its_port(monkeypatch):
    """no hard-coded default MASTER_PORT."""
    src = open(os.path.join(ROOT, "hades252_amd", "sharding.py")).read()
    assert "29511" not in src
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        sharding.init_process_group("gloo")
    monkeypatch.setenv("WORLD_SIZE", "1")
    sharding.init_process_group("gloo")
    try:
        assert 1024 < int(os.environ["MASTER_PORT"]) < 65536 and sharding.world_size() == 1
        assert sharding.reduce_max(2.5) == 2.5 and sharding.combine_digests([1, 2, 3, sharding.M64]) == [1, 2, 3, sharding.M64]
    finally:
        dist.destroy_process_group()
        os.environ.pop("MASTER_PORT", None)
