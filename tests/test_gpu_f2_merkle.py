"""GPU tier, SURVEY section 8 row f2 (+ BASELINE configs[3]): Merkle levels, trees of any leaf count and arity 1..4, openings,
batched verification, incremental updates, forests, the empty-subtree table -- against the oracle, and at 2^16 / 2^20 /
2^24 leaves against the oracle's committed roots."""
import ctypes
import hashlib
import json
import os
import random
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hades_spec as S  # noqa: E402,F401
from oracle_lib import P, R, limbs_of, int_of, digest_ref  # noqa: E402,F401
from gpu_common import *  # noqa: E402,F401,F403  (helpers shared by the GPU tier; fixtures torch_cuda / H: conftest.py)

pytestmark = pytest.mark.gpu


def test_merkle(torch_cuda, H, oracle, kat):
    torch = torch_cuda
    g = kat["merkle4_root_mont"]
    tag = S.to_mont(g["tag"])
    for n_str, root_hex in g["leaves_gen_b"].items():
        leaves = H.gen_b(int(n_str), "cuda")
        root = H.merkle4_root(leaves, tag, g["out_idx"])
        assert hex(int_of(to_host(root))) == root_hex
    # one level, ragged count, every output index
    n_par = 1000
    ch = oracle.gen_b(4242, 4 * n_par)
    for out_idx in range(5):
        par = H.merkle4_level(to_dev(torch, ch), tag, out_idx)
        assert (to_host(par) == oracle.merkle4_level(ch, tag, out_idx)).all()
    # 4^8 leaves: device root == oracle root
    n = 4 ** 8
    leaves = H.gen_b(n, "cuda")
    root = H.merkle4_root(leaves, tag, 1)
    assert (to_host(root) == oracle.merkle4_root(oracle.gen_b(0, n), tag, 1)).all()
    # 8 leaves are a valid (ragged) arity-4 tree since round 3: two parents, then a root over [p0, p1, 0, 0]
    l8 = oracle.gen_b(0, 8)
    assert (to_host(H.merkle4_root(H.gen_b(8, "cuda"), tag, 1)) == oracle.merkle_tree(l8, 4, tag, 1)[-1]).all()
    with pytest.raises(ValueError):
        H.merkle4_root(H.gen_b(1, "cuda"), tag, 1)


# ---------------------------------------------------------------------------------------------
# Merkle: arity 2 and 4, fused builder, every level, openings
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arity,depths", [(4, (1, 2, 3, 4, 5, 6, 7, 8, 9, 10)), (2, (1, 2, 3, 6, 7, 8, 13, 14, 15, 16, 17))])
def test_merkle_roots_and_levels_vs_oracle(torch_cuda, H, oracle, arity, depths):
    """Roots (root-only path) and EVERY level (build path) vs the oracle, from one-level trees through the
    single-block, two-launch fused, bulk + fused and two-levels-per-launch bulk regimes (the last at 4^10 / 2^17
    leaves)."""
    torch = torch_cuda
    tag = S.to_mont((1 << arity) - 1)
    for d in depths:
        n = arity ** d
        leaves = oracle.gen_b(1000 * d + arity, n)
        levels = oracle.merkle_tree(leaves, arity, tag, 1)
        dl = to_dev(torch, leaves).view(-1, 4)
        root = to_host(H.merkle_root(dl, arity, tag, 1))
        assert (root == levels[-1]).all(), (arity, d)
        tree = to_host(H.merkle_build(dl, arity, tag, 1))
        assert (tree == np.concatenate(levels)).all(), (arity, d)
    # another output word / tag
    leaves = oracle.gen_b(5, arity ** 4)
    got = to_host(H.merkle_root(to_dev(torch, leaves).view(-1, 4), arity, S.to_mont(77), 3))
    assert (got == oracle.merkle_tree(leaves, arity, S.to_mont(77), 3)[-1]).all()


@pytest.mark.parametrize("arity,depth", [(4, 6), (2, 11), (4, 9)])
def test_merkle_openings(torch_cuda, H, oracle, arity, depth):
    """Every sibling of every queried path vs the oracle's tree, and each opening re-verified by the oracle:
    leaf + path -> root."""
    torch = torch_cuda
    rng = random.Random(arity * 100 + depth)
    tag = S.to_mont((1 << arity) - 1)
    n = arity ** depth
    leaves = oracle.gen_b(99 + depth, n)
    dl = to_dev(torch, leaves).view(-1, 4)
    tree = H.merkle_build(dl, arity, tag, 1)
    idx = [0, 1, arity - 1, arity, n - 1, n // 2] + [rng.randrange(n) for _ in range(40)]
    paths = H.merkle_open(dl, tree, arity, to_dev(torch, np.array(idx, dtype=np.uint64)))
    host = paths.cpu().numpy().view(np.uint64).reshape(len(idx), depth, arity - 1, 4)
    levels = [leaves.reshape(-1, 4)] + [l.reshape(-1, 4) for l in oracle.merkle_tree(leaves, arity, tag, 1)]
    root = levels[-1].reshape(4)
    for t, i in enumerate(idx):
        node = i
        for l in range(depth):
            first, pos = node - node % arity, node % arity
            sib = [levels[l][first + c] for c in range(arity) if c != pos]
            assert (host[t, l] == np.array(sib)).all(), (i, l)
            node //= arity
        if t < 12:
            assert (oracle.merkle_verify_path(leaves[4 * i:4 * i + 4], i, host[t], arity, tag, 1) == root).all()
    with pytest.raises(IndexError):
        H.merkle_open(dl, tree, arity, to_dev(torch, np.array([n], dtype=np.uint64)))
    # the C ABI itself never reads outside the tree: an out-of-range index gives an all-zero path
    from hades252_amd import _lib
    bad = to_dev(torch, np.array([n + 5, 1], dtype=np.uint64))
    out = torch.full((2, depth, arity - 1, 4), -1, dtype=torch.int64, device="cuda")
    assert _lib.lib().hades252_merkle_open_dev(dl.data_ptr(), tree.data_ptr(), n, arity, bad.data_ptr(), 2,
                                                out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(out[0].abs().sum().item()) == 0 and torch.equal(out[1], paths[1])


def test_merkle_argument_errors(torch_cuda, H, hades_lib):
    torch = torch_cuda
    t = torch.zeros((48, 4), dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        H.merkle_root(t[:1], 4, 1)             # a tree needs at least two leaves
    with pytest.raises(ValueError):
        H.merkle_root(t[:25], 5, 1)            # arity 5 does not fit WIDTH = 5 (tag + children)
    with pytest.raises(ValueError):
        H.merkle_root(t[:9], 1, 1)             # arity 1 never shrinks: levels only
    tag = (ctypes.c_uint64 * 4)(1, 0, 0, 0)
    root = torch.zeros(4, dtype=torch.int64, device="cuda")
    # one-level tree: no scratch needed, NULL accepted
    assert hades_lib.hades252_merkle4_root_dev(t.data_ptr(), 4, None, 0, tag, 1, root.data_ptr(), None) == 0
    # scratch too small / missing
    assert hades_lib.hades252_merkle4_root_dev(t.data_ptr(), 16, None, 0, tag, 1, root.data_ptr(), None) == -5
    assert hades_lib.hades252_merkle4_root_dev(t.data_ptr(), 16, t.data_ptr(), 32, tag, 1, root.data_ptr(), None) == -5
    # misaligned root is rejected before anything is enqueued
    sc = torch.zeros(64, dtype=torch.int64, device="cuda")
    assert hades_lib.hades252_merkle4_root_dev(t.data_ptr(), 16, sc.data_ptr(), 512, tag, 1, root.data_ptr() + 8, None) == -1


@pytest.mark.parametrize("arity", [1, 2, 3, 4])
def test_merkle_single_levels_all_arities_and_kernels(torch_cuda, H, oracle, arity):
    """Full and ragged levels around every dispatch boundary (one parent per wave / five waves / per lane)."""
    torch = torch_cuda
    pad = oracle.gen_b(555, 1)
    dpad = to_dev(torch, pad).view(1, 4)
    for n_children in (1, arity, arity + 1, 5 * arity - 1, 64 * arity, 1024 * arity, 1024 * arity + 1, 1025 * arity - 1,
                       4096 * arity, 4096 * arity - 1 if arity > 1 else 4095, 4096 * arity + 1, 5000 * arity, (1 << 14) * arity + 3, 40000 * arity - 2):
        ch = oracle.gen_b(17 * n_children, n_children)
        exp = oracle.merkle_level_pad(ch, arity, TAG[arity], 1, pad)
        got = to_host(H.merkle_level(to_dev(torch, ch).view(-1, 4), arity, TAG[arity], 1, pad=dpad))
        assert (got == exp).all(), (arity, n_children)
    # zero padding when no table is given; out_idx other than 1
    ch = oracle.gen_b(3, 7 * arity + (1 if arity > 1 else 0))
    assert (to_host(H.merkle_level(to_dev(torch, ch).view(-1, 4), arity, TAG[arity], 3)) ==
            oracle.merkle_level_pad(ch, arity, TAG[arity], 3)).all()


@pytest.mark.parametrize("arity,n_leaves", [(3, 3 ** 9), (4, 4 ** 7 * 3), (2, 2), (2, 3), (3, 4), (4, 5), (2, 1000),
                                            (3, 2 ** 15 + 11), (4, 100001), (4, 4 ** 8 + 1), (2, 2 ** 16 - 1)])
def test_merkle_any_leaf_count_build_open_verify(torch_cuda, H, oracle, arity, n_leaves):
    """Trees over arbitrary leaf counts with the empty-subtree padding table: every level vs the oracle, root-only path,
    openings (incl. positions past the end of a level) and batched verification back to the root."""
    torch = torch_cuda
    tag = TAG[arity]
    depth = H.merkle_depth(n_leaves, arity)
    e0 = S.to_mont(0x5EED)
    pad = H.merkle_empty_digests(arity, depth, e0, tag, 1)
    opad = oracle.merkle_empty_digests(arity, depth, e0, tag, 1)
    assert (to_host(pad).reshape(-1, 4) == opad).all()
    leaves = oracle.gen_b(n_leaves, n_leaves)
    dl = to_dev(torch, leaves).view(-1, 4)
    levels = oracle.merkle_tree(leaves, arity, tag, 1, opad)
    assert [l.size // 4 for l in levels] == H.merkle_level_sizes(n_leaves, arity) and len(levels) == depth
    tree = H.merkle_build(dl, arity, tag, 1, pad=pad)
    assert (to_host(tree) == np.concatenate(levels)).all()
    assert (to_host(H.merkle_root(dl, arity, tag, 1, pad=pad)) == levels[-1]).all()
    # zero padding (no table) is a different, equally well-defined tree
    zl = oracle.merkle_tree(leaves, arity, tag, 1)
    assert (to_host(H.merkle_root(dl, arity, tag, 1)) == zl[-1]).all()
    # openings: first, last (its siblings are padding wherever the level is ragged), random
    rng = random.Random(n_leaves)
    idx = sorted(set([0, n_leaves - 1, n_leaves // 2] + [rng.randrange(n_leaves) for _ in range(61)]))
    didx = torch.tensor(idx, dtype=torch.int64, device="cuda")
    paths = H.merkle_open(dl, tree, arity, didx, pad=pad)
    hp = to_host(paths).reshape(len(idx), depth, arity - 1, 4)
    for q, i in enumerate(idx[:8] + idx[-8:]):
        qq = idx.index(i)
        assert (oracle.merkle_verify_path(rows(leaves)[i], i, hp[qq], arity, tag, 1) == levels[-1]).all(), i
    roots = H.merkle_verify(dl[didx].contiguous(), didx, paths, arity, tag, 1)
    assert bool((roots == to_dev(torch, levels[-1]).view(1, 4)).all())
    # a tampered sibling or leaf no longer verifies
    bad = paths.clone()
    t = min(3, len(idx) - 1)
    bad[t, depth - 1, 0, 0] ^= 1
    r2 = H.merkle_verify(dl[didx].contiguous(), didx, bad, arity, tag, 1)
    assert not bool((r2[t] == roots[t]).all()) and bool((r2[:t] == roots[:t]).all())


def test_merkle_verify_2pow16_queries(torch_cuda, H, oracle):
    torch = torch_cuda
    n, arity, tag = 4 ** 8, 4, TAG[4]
    leaves = H.gen_b(n, "cuda")
    tree = H.merkle_build(leaves, arity, tag, 1)
    g = torch.Generator(device="cpu")
    g.manual_seed(5)
    idx = torch.randint(0, n, (1 << 16,), generator=g, dtype=torch.int64).cuda()
    paths = H.merkle_open(leaves, tree, arity, idx)
    roots = H.merkle_verify(leaves[idx].contiguous(), idx, paths, arity, tag, 1)
    assert bool((roots == tree[-1:]).all())
    # root of the same tree by the oracle (2^16 leaves = 21 845 permutations)
    exp = oracle.merkle_tree(to_host(leaves), arity, tag, 1)[-1]
    assert (to_host(tree[-1]) == exp).all()
    # arity 1 chains: verify = depth successive single-child hashes
    chain = oracle.gen_b(9, 300)
    d1 = to_dev(torch, chain).view(-1, 4)
    r1 = H.merkle_verify(d1, torch.zeros(300, dtype=torch.int64, device="cuda"),
                         torch.zeros((300, 5, 0, 4), dtype=torch.int64, device="cuda"), 1, TAG[1], 1)
    cur = chain
    for _ in range(5):
        cur = oracle.merkle_level(cur, 1, TAG[1], 1)
    assert (to_host(r1) == cur).all()


@pytest.mark.parametrize("arity", [1, 2, 3, 4])
def test_small_batch_verify_one_query_per_wave(torch_cuda, H, oracle, arity):
    """The same openings verified one per wave (<= 1024 queries, both forms) and one per lane (> 1024) give the same roots;
    tampered siblings are caught."""
    torch = torch_cuda
    tag = TAG[arity]
    if arity == 1:
        chain = H.gen_b(17000, "cuda")
        z = torch.zeros(17000, dtype=torch.int64, device="cuda")
        e = torch.zeros((17000, 6, 0, 4), dtype=torch.int64, device="cuda")
        ref = H.merkle_verify(chain, z, e, 1, tag, 1)
        cur = to_host(chain[:50])
        for _ in range(6):
            cur = oracle.merkle_level(cur, 1, tag, 1)
        assert (to_host(ref[:50]) == cur).all()
        for m in (1, 3, 768, 769, 1024, 1025, 4096, 4097, 16384):
            assert torch.equal(H.merkle_verify(chain[:m].contiguous(), z[:m].contiguous(), e[:m].contiguous(), 1, tag, 1), ref[:m])
        return
    n_leaves = arity ** 7 + 5
    depth = H.merkle_depth(n_leaves, arity)
    pad = H.merkle_empty_digests(arity, depth, S.to_mont(3), tag, 1)
    leaves = H.gen_b(n_leaves, "cuda")
    tree = H.merkle_build(leaves, arity, tag, 1, pad=pad)
    g = torch.Generator(device="cpu")
    g.manual_seed(arity)
    idx = torch.randint(0, n_leaves, (17000,), generator=g, dtype=torch.int64).cuda()
    idx[0], idx[1] = n_leaves - 1, 0
    paths = H.merkle_open(leaves, tree, arity, idx, pad=pad)
    lv = leaves[idx].contiguous()
    ref = H.merkle_verify(lv, idx, paths, arity, tag, 1)                      # 17 000 queries: one per lane
    assert bool((ref == tree[-1:]).all())
    for m in (1, 2, 3, 4, 767, 768, 769, 1024, 1025, 1026, 4095, 4096, 4097, 5000, 16384, 16385):   # per wave / row / five waves / lane
        r = H.merkle_verify(lv[:m].contiguous(), idx[:m].contiguous(), paths[:m].contiguous(), arity, tag, 1)
        assert torch.equal(r, ref[:m]), m
    bad = paths[:5].clone()
    bad[2, depth // 2, 0, 1] ^= 4
    r = H.merkle_verify(lv[:5].contiguous(), idx[:5].contiguous(), bad, arity, tag, 1)
    assert not torch.equal(r[2], ref[2]) and torch.equal(r[:2], ref[:2]) and torch.equal(r[3:], ref[3:5])
    # out_idx other than 1, against the oracle's path walk
    hp = to_host(paths[:3]).reshape(3, depth, arity - 1, 4)
    r3 = to_host(H.merkle_verify(lv[:3].contiguous(), idx[:3].contiguous(), paths[:3].contiguous(), arity, tag, 3)).reshape(3, 4)
    hl = to_host(lv[:3]).reshape(3, 4)
    for q in range(3):
        assert (oracle.merkle_verify_path(hl[q], int(idx[q].item()), hp[q], arity, tag, 3) == r3[q]).all()


@pytest.mark.parametrize("arity,n_leaves,n_updates", [(4, 4 ** 7 * 3, 1), (4, 4 ** 7 * 3, 300), (4, 100001, 800), (2, 2 ** 15 + 11, 1024),
                                                      (3, 3 ** 9, 5000), (4, 4 ** 8 + 1, 20000), (2, 3, 2), (4, 5, 1),
                                                      (4, 4 ** 8 + 1, 1025), (2, 2 ** 15 + 11, 4096), (3, 3 ** 9, 4097)])
def test_merkle_update_equals_rebuild(torch_cuda, H, oracle, arity, n_leaves, n_updates):
    """Overwrite k leaves, re-hash their ancestors only: the tree equals the oracle's tree over the new leaves -- sorted and
    shuffled index lists with repeats, out-of-range indices ignored, every kernel form (wave / helped wave / lane / whole level)."""
    torch = torch_cuda
    tag = TAG[arity]
    depth = H.merkle_depth(n_leaves, arity)
    e0 = S.to_mont(0xE0)
    pad = H.merkle_empty_digests(arity, depth, e0, tag, 1)
    opad = oracle.merkle_empty_digests(arity, depth, e0, tag, 1)
    leaves = rows(oracle.gen_b(n_leaves + 1, n_leaves)).copy()
    dl = to_dev(torch, leaves.reshape(-1)).view(-1, 4)
    tree = H.merkle_build(dl, arity, tag, 1, pad=pad)
    rng = random.Random(n_leaves * 31 + n_updates)
    for order in ("sorted", "shuffled"):
        idx = [rng.randrange(n_leaves) for _ in range(n_updates)]
        idx[0] = n_leaves - 1                                  # the ragged end: its siblings are padding
        if n_updates > 2:
            idx[1] = idx[2]                                    # a repeat
        idx = sorted(idx) if order == "sorted" else idx
        fresh = rows(oracle.gen_b(rng.randrange(1 << 30), n_updates))
        for q, i in enumerate(idx):
            leaves[i] = fresh[q]
        dl.copy_(to_dev(torch, leaves.reshape(-1)).view(-1, 4))
        before = tree.clone()
        with_bogus = idx + [n_leaves, 2 ** 63 - 1]             # ignored, never read or written
        didx = torch.tensor(with_bogus, dtype=torch.int64, device="cuda")
        H.merkle_update(dl, tree, arity, didx, tag, 1, pad=pad)
        exp = np.concatenate(oracle.merkle_tree(leaves.reshape(-1), arity, tag, 1, opad))
        assert (to_host(tree) == exp).all(), order
        if n_updates * depth < sum(H.merkle_level_sizes(n_leaves, arity)) // 4:
            assert int((tree != before).any(dim=1).sum().item()) <= n_updates * depth      # nothing else was touched
    # no table = zero padding, and an empty update list is a no-op
    t0 = H.merkle_build(dl, arity, tag, 1)
    leaves[0] = rows(oracle.gen_b(77, 1))[0]
    dl.copy_(to_dev(torch, leaves.reshape(-1)).view(-1, 4))
    H.merkle_update(dl, t0, arity, torch.zeros(0, dtype=torch.int64, device="cuda"), tag, 1)
    H.merkle_update(dl, t0, arity, torch.zeros(1, dtype=torch.int64, device="cuda"), tag, 1)
    assert (to_host(t0) == np.concatenate(oracle.merkle_tree(leaves.reshape(-1), arity, tag, 1))).all()


def test_merkle_update_at_baseline_size(torch_cuda, H):
    """BASELINE configs[3] (arity 4, 2^24 leaves): k updated leaves, then the whole tree equals a fresh build (the build
    itself is checked against the oracle by decomposition in test_merkle_roots_and_levels_vs_oracle) -- k on both sides of every kernel choice."""
    torch = torch_cuda
    n, tag = 1 << 24, TAG[4]
    leaves = H.gen_b(n, "cuda")
    tree = H.merkle_build(leaves, 4, tag, 1)
    g = torch.Generator(device="cpu")
    g.manual_seed(2024)
    for k in (1, 700, 1000, 3000, 1 << 17):
        idx = torch.randint(0, n, (k,), generator=g, dtype=torch.int64).cuda()
        if k > 1:
            idx = idx[torch.randperm(k, generator=g).cuda()] if k == 3000 else torch.sort(idx)[0]
        leaves[idx] = H.gen_b(k, "cuda", first_elem=(1 << 40) + 7 * k)
        H.merkle_update(leaves, tree, 4, idx, tag, 1)
        assert bool((tree == H.merkle_build(leaves, 4, tag, 1)).all()), k


@pytest.mark.parametrize("arity,k,n_trees", [(4, 4, 10 ** 4), (4, 1, 1000), (2, 10, 333), (3, 5, 2000), (4, 6, 7)])
def test_merkle_forest_vs_oracle(torch_cuda, H, oracle, arity, k, n_trees):
    torch = torch_cuda
    per = arity ** k
    leaves = H.gen_b(n_trees * per, "cuda")
    roots = to_host(H.merkle_forest(leaves, n_trees, arity, TAG[arity], 1)).reshape(n_trees, 4)
    host = to_host(leaves)
    # the forest's levels are one big level each: oracle level by level over all trees at once
    cur = host
    for _ in range(k):
        cur = oracle.merkle_level(cur, arity, TAG[arity], 1)
    assert (roots.reshape(-1) == cur).all()
    # and a single tree of the forest equals merkle_root of its leaves
    t = n_trees // 2
    if per >= 2:
        one = to_host(H.merkle_root(leaves[t * per:(t + 1) * per], arity, TAG[arity], 1))
        assert (one == roots[t]).all()
    with pytest.raises(Exception):
        H.merkle_forest(leaves[: n_trees * per - 1], n_trees, arity, TAG[arity], 1)


def test_empty_digests_is_graph_capturable(torch_cuda, H, oracle):
    """hades252_merkle_empty_digests_dev takes e0 by value: captured once, replayed after the caller's host
    array is long gone."""
    torch = torch_cuda
    e0 = S.to_mont(123456789)
    exp = oracle.merkle_empty_digests(3, 7, e0, S.to_mont(7), 1)
    eager = H.merkle_empty_digests(3, 7, e0, S.to_mont(7), 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            pad = H.merkle_empty_digests(3, 7, e0, S.to_mont(7), 1)
    junk = [np.random.randint(0, 2 ** 62, size=1 << 16) for _ in range(8)]       # recycle host memory
    pad.zero_()
    g.replay()
    torch.cuda.synchronize()
    del junk
    assert torch.equal(pad, eager)
    assert (to_host(pad) == exp.reshape(-1)).all()


# ---------------------------------------------------------------------------------------------
# BASELINE configs[3] at full size against the oracle's committed roots (tests/golden/kat.json, merkle4_full_size:
# the C oracle applying src/strategies.rs:140 21 845 / 349 525 / 5 592 405 times)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log4", [8, 10, 12])
def test_merkle_root_equals_golden_at_full_size(torch_cuda, H, kat, log4):
    torch = torch_cuda
    n = 4 ** log4
    gold = kat["merkle4_full_size"][str(n)]
    leaves = H.gen_b(n, "cuda")
    assert hex_of(H.merkle4_root(leaves, TAG4, 1)) == gold["root"]
    # ... and the level two below the root = the 16 sub-roots of the multi-GPU decomposition:
    # every sub-tree on its own, then the built tree's own copy of that level
    q = n // 16
    subs = [hex_of(H.merkle4_root(leaves[i * q:(i + 1) * q], TAG4, 1)) for i in range(16)]
    assert subs == gold["sub_roots_16"]
    if log4 <= 10:
        tree = H.merkle_build(leaves, 4, TAG4, 1)
        flat = to_host(tree)
        lvl = flat[-(16 + 4 + 1) * 4:-(4 + 1) * 4]
        assert [hex(int_of(lvl[4 * i:4 * i + 4])) for i in range(16)] == gold["sub_roots_16"]
        assert hex(int_of(flat[-4:])) == gold["root"]


def test_config4_merkle_2pow24(torch_cuda, H, oracle, kat):
    """BASELINE config[3]: arity-4 tree over 2^24 leaves, level by level (5 592 405 permutations).
    The root equals the ORACLE's root at full size (tests/golden/kat.json `merkle4_full_size`: the C oracle on all host
    cores, committed -- SURVEY section 8(d) config 4); it also equals the root of the 4 sub-tree roots (the multi-GPU
    decomposition of SURVEY section 8(e)), and a 2^20-leaf sub-tree root is recomputed by the oracle live."""
    torch = torch_cuda
    tag = S.to_mont(15)
    n = 1 << 24
    leaves = H.gen_b(n, "cuda")
    root = to_host(H.merkle4_root(leaves, tag, 1))
    assert hex(int_of(root)) == kat["merkle4_full_size"][str(n)]["root"]
    q = n // 4
    subs = torch.cat([H.merkle4_root(leaves[i * q:(i + 1) * q], tag, 1) for i in range(4)])
    top = to_host(H.merkle4_level(subs, tag, 1))
    assert (top == root).all()
    sub = 1 << 20
    exp = oracle.merkle4_root(oracle.gen_b(0, sub), tag, 1)
    assert (to_host(H.merkle4_root(leaves[:sub], tag, 1)) == exp).all()
