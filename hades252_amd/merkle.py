"""Arity-4 Poseidon Merkle trees over the batched permutation (BASELINE config "Merkle 2^24").

node = perm([tag, c0, c1, c2, c3])[out_idx] -- the caller shape of dusk-poseidon (reference
README.md:9); tag and out_idx are parameters because that crate is not part of the reference tree
(defaults 15 and 1).

Single GPU: level by level, one launch per level (hades252_merkle4_root_dev).
Multi GPU: the leaves are sharded contiguously; with world size W = 4^k * m
every rank owns whole sub-trees, builds their roots locally (no communication), then the W-rank
job exchanges only the sub-roots -- 32 bytes each -- with one all_gather and every rank (or rank 0)
hashes the few remaining top levels.  That all_gather is the path's only real exchange step.
"""
from __future__ import annotations

from . import strategy as H


def subtree_split(n_leaves: int, world: int):
    """How a tree of n_leaves (a power of 4) splits over `world` ranks: returns
    (leaves_per_subtree, subtrees_per_rank).  world must divide the number of sub-trees at some
    level, i.e. world = 4^k or 2 * 4^k (1, 2, 4, 8, 16, 32, ..): rank g owns the sub-trees
    [g * subtrees_per_rank, (g + 1) * subtrees_per_rank) of that level, i.e. the leaves
    [g * n_leaves / world, (g + 1) * n_leaves / world).  BASELINE configs[3] on 8 GPUs: 2^24 leaves ->
    16 sub-trees of 2^20 leaves, two per rank."""
    if n_leaves < 4 or n_leaves & (n_leaves - 1) or (n_leaves.bit_length() - 1) % 2:
        raise ValueError("n_leaves must be a power of 4")
    if world < 1 or world & (world - 1):
        raise ValueError("world size must be a power of 2")
    n_sub = 1
    while n_sub % world != 0:
        n_sub *= 4                      # number of sub-trees at this depth
        if n_sub > n_leaves // 4:
            raise ValueError("tree too small for this world size")
    return n_leaves // n_sub, n_sub // world


def local_subroots(leaves_shard, n_leaves_total: int, world: int, tag_mont: int, out_idx: int = 1):
    """Roots of the sub-trees this rank owns (tensor [subtrees_per_rank, 4] int64)."""
    import torch
    per_sub, subs_per_rank = subtree_split(n_leaves_total, world)
    flat = leaves_shard.view(-1, 4)
    assert flat.shape[0] == per_sub * subs_per_rank, "shard does not hold whole sub-trees"
    roots = [H.merkle4_root(flat[i * per_sub:(i + 1) * per_sub], tag_mont, out_idx) for i in range(subs_per_rank)]
    return torch.stack(roots)


def finish_from_subroots(subroots, tag_mont: int, out_idx: int = 1):
    """Hash the gathered sub-roots (count a power of 4, or 1) down to the root."""
    n = subroots.view(-1, 4).shape[0]
    if n == 1:
        return subroots.view(4)
    return H.merkle4_root(subroots.contiguous(), tag_mont, out_idx)


def merkle4_root_sharded(leaves_shard, n_leaves_total: int, tag_mont: int, out_idx: int = 1):
    """Root of the whole tree from per-rank shards (one process per GPU, torch.distributed
    initialised with the nccl/RCCL backend -- or gloo, whose gather travels through host memory;
    world size 1 needs no process group).  Every rank returns the root."""
    import torch
    from . import sharding
    world = sharding.world_size()
    mine = local_subroots(leaves_shard, n_leaves_total, world, tag_mont, out_idx)
    if world == 1:
        return finish_from_subroots(mine, tag_mont, out_idx)
    gathered = sharding.all_gather_tensor(mine)     # world * subtrees_per_rank * 32 bytes in total, rank order
    return finish_from_subroots(torch.cat(gathered), tag_mont, out_idx)
