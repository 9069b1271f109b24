"""GPU tier, round 3: the wire format pinned to reference-held bytes, the page-locked host path, the in-process
multi-worker path with more workers than devices, the lane-split low-latency kernel, general Merkle trees
(any arity 1..4, any leaf count, forests, path verification) and the streaming / bucketed sponge.
Everything goes through the C ABI."""
import ctypes
import hashlib
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hades_spec as S  # noqa: E402
from oracle_lib import P, R, limbs_of, int_of  # noqa: E402
from test_blob_kat import blob_bytes  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def H(hades_lib):
    from hades252_amd import strategy
    return strategy


def to_dev(torch, arr):
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    return torch.from_numpy((a if a.flags.writeable else a.copy()).view(np.int64)).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint64).reshape(-1)


# ---------------------------------------------------------------------------------------------
# f3 pinned to reference-held bytes: the reference's own test_round_constants
# (src/round_constants.rs:55-65) executed on the HIP path, made absolute with assets/ark.bin / mds.bin
# ---------------------------------------------------------------------------------------------
def test_wire_format_pinned_to_reference_blobs(torch_cuda, H):
    torch = torch_cuda
    strat = H.ScalarStrategy()
    # ROUND_CONSTANTS as the device holds it: zero states + add_round_key at every cursor = the table itself
    zeros = torch.zeros((192, 5, 4), dtype=torch.int64, device="cuda")
    for r in range(192):
        strat.add_round_key(H.RoundConstantsIter(5 * r), zeros[r])
    table = zeros.view(960, 4)
    assert bool((table != 0).any(dim=1).all())                      # every constant is non-zero (:58)
    ark = blob_bytes("ark")                                         # sha256-pinned; == the reference's file here
    got = to_host(H.to_bytes(table)).tobytes()
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(ark).hexdigest()
    assert got == ark                                               # to_bytes(ROUND_CONSTANTS[i]) == chunk i
    back = H.from_bytes(to_dev(torch, np.frombuffer(ark, dtype=np.uint64)).view(960, 4))
    assert bool((back == table).all())                              # from_bytes(chunk i) == ROUND_CONSTANTS[i] (:61-62)
    # MDS_MATRIX as the device applies it: mul_matrix of the unit vector e_j (Montgomery one in word j) = column j
    one = np.array(limbs_of(R), dtype=np.uint64)
    units = np.zeros((5, 5, 4), dtype=np.uint64)
    for j in range(5):
        units[j, j] = one
    cols = to_dev(torch, units.reshape(-1)).view(5, 5, 4)
    strat.mul_matrix(H.RoundConstantsIter(), cols)
    mds_dev = cols.permute(1, 0, 2).contiguous().view(25, 4)         # [i][j] = column j, word i
    mds = blob_bytes("mds")
    assert to_host(H.to_bytes(mds_dev)).tobytes() == mds
    assert bool((H.from_bytes(to_dev(torch, np.frombuffer(mds, dtype=np.uint64)).view(25, 4)) == mds_dev).all())


# ---------------------------------------------------------------------------------------------
# the drop-in boundary: page-locked host buffers, the three-stream chunk pipeline, many workers on one device
# ---------------------------------------------------------------------------------------------
def test_host_alloc_register_roundtrip(torch_cuda, hades_lib, H, oracle):
    n = 3 * (1 << 16) + 777                                  # several chunks + a ragged tail
    inp = oracle.gen_b(5 * 9000, 5 * n)
    exp = oracle.perm_batch(inp)
    # (a) memory allocated by the library
    with H.HostBuffer(n) as hb:
        assert H.host_is_pinned(hb.array)
        hb.array[:] = inp
        H.ScalarStrategy().perm(hb.array)
        assert (hb.array == exp).all()
        # a sub-range of a pinned buffer is pinned too (what a worker of perm_multi gets)
        assert hades_lib.hades252_host_is_pinned(ctypes.c_void_p(hb.ptr.value + 160 * 5), 160 * 100) == 1
        assert hades_lib.hades252_host_unregister(hb.ptr) == -1       # not a registered range
    # (b) the caller's own allocation, registered once, used for several calls
    mine = inp.copy()
    assert not H.host_is_pinned(mine)
    H.host_register(mine)
    assert H.host_is_pinned(mine)
    H.ScalarStrategy().perm(mine)
    assert (mine == exp).all()
    mine[:] = inp
    H.ScalarStrategy().perm(mine)
    assert (mine == exp).all()
    assert hades_lib.hades252_host_free(mine.ctypes.data_as(ctypes.c_void_p)) == -1   # not from host_alloc
    H.host_unregister(mine)
    assert not H.host_is_pinned(mine)
    assert hades_lib.hades252_host_unregister(mine.ctypes.data_as(ctypes.c_void_p)) == -1   # already gone
    # (c) pageable memory, per-call registration refused / disabled: same bits
    plain = inp.copy()
    H.ScalarStrategy().perm(plain)
    assert (plain == exp).all()
    # argument errors
    assert hades_lib.hades252_host_alloc(None, 100) == -1
    out = ctypes.c_void_p()
    assert hades_lib.hades252_host_alloc(ctypes.byref(out), 0) == -1
    assert hades_lib.hades252_host_register(None, 10) == -1
    assert hades_lib.hades252_host_free(None) == 0 and hades_lib.hades252_host_unregister(None) == 0


@pytest.mark.parametrize("n_chunks", [2, 3, 6, 7, 13])
def test_host_pipeline_slot_reuse(torch_cuda, H, oracle, monkeypatch, n_chunks):
    """Chunk counts around the number of pipeline slots (6): every slot-reuse pattern, ragged last chunk."""
    n = (n_chunks - 1) * (1 << 16) + 4321
    inp = oracle.gen_b(12345, 5 * n)
    exp = oracle.perm_batch(inp)
    with H.HostBuffer(n) as hb:
        hb.array[:] = inp
        H.ScalarStrategy().perm(hb.array)
        assert (hb.array == exp).all()


def test_host_bytes_format_through_pipeline(torch_cuda, hades_lib, H, oracle):
    n = 2 * (1 << 16) + 99
    inp = oracle.gen_b(777, 5 * n)
    exp = oracle.perm_batch(inp)
    canon_in = to_host(H.to_bytes(to_dev(torch_cuda, inp)))
    canon_exp = to_host(H.to_bytes(to_dev(torch_cuda, exp)))
    buf = canon_in.copy()
    assert hades_lib.hades252_perm_batch_bytes(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
    assert (buf == canon_exp).all()


@pytest.mark.parametrize("workers", [2, 3, 8, 64])
def test_multi_more_workers_than_devices(torch_cuda, hades_lib, H, oracle, workers):
    """hades252_perm_batch_multi with worker g on device g % (visible devices): the hipSetDevice threads, the shard
    arithmetic and the register-once path of an 8-GPU node, run on whatever this box has."""
    for n in (70001, 1 << 17, 5):                            # not divisible by the worker count; n < workers for 8, 64
        inp = oracle.gen_b(4242 + n, 5 * n)
        exp = oracle.perm_batch(inp)
        a = inp.copy()
        H.perm_multi(a, workers, virtual=True)               # pageable: >= 8 MiB is registered once for all workers
        assert (a == exp).all(), (workers, n)
    with H.HostBuffer(70001) as hb:                          # caller-pinned memory shared by all workers
        inp = oracle.gen_b(99, 5 * 70001)
        hb.array[:] = inp
        H.perm_multi(hb.array, workers, virtual=True)
        assert (hb.array == oracle.perm_batch(inp)).all()
    ndev = hades_lib.hades252_device_count()
    tiny = oracle.gen_b(0, 5 * 4)
    p = tiny.ctypes.data_as(ctypes.c_void_p)
    assert hades_lib.hades252_perm_batch_multi_ex(p, 4, ndev + 1, 0) == -1          # real devices only
    assert hades_lib.hades252_perm_batch_multi_ex(p, 4, 65, 1) == -1                # worker cap
    assert hades_lib.hades252_perm_batch_multi_ex(p, 4, 2, 2) == -1                 # unknown flag
    assert hades_lib.hades252_perm_batch_multi_ex(p, 0, 2, 1) == 0


def test_multi_workers_concurrent_with_host_calls(torch_cuda, H, oracle):
    """Workers sharing a device while other host threads call perm: the pipe pool under contention."""
    import threading
    n = 1 << 17
    inp = oracle.gen_b(31, 5 * n)
    exp = oracle.perm_batch(inp)
    results = {}

    def run(tag, fn):
        a = inp.copy()
        fn(a)
        results[tag] = bool((a == exp).all())

    threads = [threading.Thread(target=run, args=("multi%d" % w, lambda a, w=w: H.perm_multi(a, w, virtual=True)))
               for w in (2, 5)]
    threads += [threading.Thread(target=run, args=("host%d" % i, lambda a: H.ScalarStrategy().perm(a))) for i in range(3)]
    threads += [threading.Thread(target=run, args=("small%d" % i, lambda a: [H.ScalarStrategy().perm(a[20 * j * 200:20 * (j + 1) * 200]) for j in range(n // 200 + 1)]))
                for i in range(1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(results.values()) and len(results) == len(threads), results


# ---------------------------------------------------------------------------------------------
# lane-split kernel: dispatch thresholds (768: one state per wave + a helper wave per three states; 1 024: one state per
# wave; 16 384: five waves per state; above: per lane)
# ---------------------------------------------------------------------------------------------
def test_default_dispatch_across_both_thresholds(torch_cuda, H, oracle):
    torch = torch_cuda
    for n in (1, 2, 3, 4, 5, 6, 7, 15, 16, 17, 767, 768, 769, 1023, 1024, 1025, 2048, 4095, 4096, 4097, (1 << 14), (1 << 14) + 1):
        inp = oracle.gen_b(11 * n, 5 * n)
        guard = np.full(40, 0xDEADBEEFCAFEF00D, dtype=np.uint64)
        exp = oracle.perm_batch(inp)
        for kernel in (0, 4, 5):
            if kernel in (4, 5) and n > 4097:
                continue
            buf = to_dev(torch, np.concatenate([guard, inp, guard]))
            H.ScalarStrategy(kernel).perm(buf[40:40 + 20 * n])
            got = to_host(buf)
            assert (got[:40] == guard).all() and (got[-40:] == guard).all(), "wrote outside the batch"
            assert (got[40:-40] == exp).all(), (n, kernel)


def test_lanes_kernel_2pow18_vs_fast(torch_cuda, H):
    torch = torch_cuda
    a = H.gen_b(5 << 18, "cuda")
    b = a.clone()
    H.ScalarStrategy(2).perm(a)
    H.ScalarStrategy(4).perm(b)
    assert torch.equal(a, b)
    c = H.gen_b(5 << 18, "cuda")
    H.ScalarStrategy(5).perm(c)                      # one state per row
    assert torch.equal(a, c)


# ---------------------------------------------------------------------------------------------
# general Merkle trees: arity 1..4, any number of leaves (padding table), openings, verification, forests
# ---------------------------------------------------------------------------------------------
TAG = {1: S.to_mont(1), 2: S.to_mont(3), 3: S.to_mont(7), 4: S.to_mont(15)}


def rows(a):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)


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
    itself is checked against the oracle by decomposition in test_gpu_round2) -- k on both sides of every kernel choice."""
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


# ---------------------------------------------------------------------------------------------
# sponge: device-side sort of ragged batches, streaming absorb / squeeze
# ---------------------------------------------------------------------------------------------
CAP = S.to_mont(1 << 64)


@pytest.mark.parametrize("pad", [0, 1])
def test_sponge_sorted_equals_unsorted_and_oracle(torch_cuda, H, oracle, pad):
    """Ragged lengths (0 .. 70 scalars, a few very long, one beyond the last sort bucket): the device-sorted run gives
    the same digests in message order as the plain run and the oracle."""
    torch = torch_cuda
    rng = random.Random(77 + pad)
    n = 5000
    lens = [rng.choice([0, 1, 3, 4, 5, 8, 9, 17, 33, 70]) if rng.random() < 0.8 else rng.randrange(0, 40) for _ in range(n)]
    lens[123] = 4 * 1030                                     # > 1023 blocks: clamps into the last bucket
    lens[4000] = 600
    offs = np.cumsum([0] + lens[:-1]).astype(np.uint64)
    pool = oracle.gen_b(8, int(sum(lens)) + 1)
    lens_a = np.array(lens, dtype=np.uint64)
    exp = oracle.sponge_var(pool, offs, lens_a, CAP, pad)
    dp, do, dl = to_dev(torch, pool).view(-1, 4), to_dev(torch, offs), to_dev(torch, lens_a)
    plain = to_host(H.sponge_hash_var(dp, do, dl, CAP, pad))
    srt = to_host(H.sponge_hash_var(dp, do, dl, CAP, pad, sort=True))
    assert (plain == exp).all() and (srt == exp).all()
    # tiny batches and n not a multiple of the block size
    for m in (1, 2, 63, 65, 257):
        e = oracle.sponge_var(pool, offs[:m], lens_a[:m], CAP, pad)
        assert (to_host(H.sponge_hash_var(dp, do[:m].contiguous(), dl[:m].contiguous(), CAP, pad, sort=True)) == e).all()


def test_sponge_sort_argument_errors(torch_cuda, hades_lib, H):
    torch = torch_cuda
    pool = H.gen_b(64, "cuda")
    off = torch.zeros(8, dtype=torch.int64, device="cuda")
    ln = torch.full((8,), 4, dtype=torch.int64, device="cuda")
    out = torch.zeros((8, 4), dtype=torch.int64, device="cuda")
    cap = (ctypes.c_uint64 * 4)(1, 0, 0, 0)
    small = torch.zeros(8, dtype=torch.int64, device="cuda")
    need = hades_lib.hades252_sponge_sort_scratch_bytes(8)
    assert need >= (1024 + 8) * 4
    assert hades_lib.hades252_sponge_hash_var_ex_dev(pool.data_ptr(), 64, off.data_ptr(), ln.data_ptr(), 8, cap, 1,
                                                     out.data_ptr(), None, small.data_ptr(), 64, None) == -5
    assert hades_lib.hades252_sponge_hash_var_ex_dev(pool.data_ptr(), 64, off.data_ptr(), ln.data_ptr(), 8, cap, 1,
                                                     out.data_ptr(), None, small.data_ptr() + 8, need, None) == -1


def test_streaming_sponge_absorb_squeeze(torch_cuda, H, oracle):
    """init + absorb (in one call, in two calls, block by block) + squeeze == the one-shot sponge without padding, and
    the full state after each absorb == the oracle's add-then-permute."""
    torch = torch_cuda
    n, t = 3000, 5
    msgs = oracle.gen_b(21, n * t * 4)                       # n messages of 4 t scalars
    exp = oracle.sponge(msgs, 4 * t, CAP, 0)
    dm = to_dev(torch, msgs).view(n, t, 4, 4)
    a = H.SpongeStates(n, CAP)
    a.absorb(dm)
    assert (to_host(a.squeeze()) == exp).all()
    b = H.SpongeStates(n, CAP)
    b.absorb(dm[:, :2].contiguous())
    b.absorb(dm[:, 2:].contiguous())
    assert torch.equal(a.states, b.states)
    c = H.SpongeStates(n, CAP)
    for i in range(t):
        c.absorb(dm[:, i].contiguous())
    assert torch.equal(a.states, c.states)
    # the whole state, not only the digest word: one absorb of one block vs oracle arithmetic
    d = H.SpongeStates(7, CAP)
    blk = oracle.gen_b(99, 7 * 4)
    d.absorb(to_dev(torch, blk).view(7, 1, 4, 4))
    st = np.zeros((7, 5, 4), dtype=np.uint64)
    st[:, 0] = np.array(limbs_of(CAP), dtype=np.uint64)
    st[:, 1:] = blk.reshape(7, 4, 4)                          # 0 + block
    assert (to_host(d.states) == oracle.perm_batch(st.reshape(-1))).all()
    for w in range(5):
        assert (to_host(d.squeeze(w)).reshape(7, 4) == to_host(d.states).reshape(7, 5, 4)[:, w]).all()
    # the C ABI equivalence promised in the header: pad_mode 0 one-shot == streaming
    assert (to_host(H.sponge_hash(to_dev(torch, msgs).view(-1, 4), 4 * t, CAP, 0)) == exp).all()


# ---------------------------------------------------------------------------------------------
# small batches: one message / state / query per wave (the low-latency forms of sponge, absorb and verification)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pad", [0, 1])
def test_small_batch_sponge_one_message_per_wave(torch_cuda, hades_lib, H, oracle, pad):
    """Batches on both sides of the two dispatch thresholds (768: helper wave, 1024: one message per lane), ragged lengths
    inside a block of three / four waves (the helped form runs every wave to the block's maximum), empty messages,
    overlapping messages, one LONG message alone, a message outside the pool."""
    torch = torch_cuda
    rng = random.Random(5 + pad)
    pool = oracle.gen_b(1234, 3000)
    dp = to_dev(torch, pool).view(-1, 4)
    for n in (1, 2, 3, 4, 5, 100, 767, 768, 769, 1023, 1024, 1025, 1027, 1100, 4095, 4096, 4097, 5000, 16383, 16384, 16385):
        lens = [rng.choice([0, 1, 2, 3, 4, 5, 7, 8, 9, 13, 40]) for _ in range(n)]
        offs = [rng.randrange(0, 3000 - l + 1) for l in lens]              # anywhere in the pool: messages overlap
        la, oa = np.array(lens, dtype=np.uint64), np.array(offs, dtype=np.uint64)
        exp = oracle.sponge_var(pool, oa, la, CAP, pad)
        got = to_host(H.sponge_hash_var(dp, to_dev(torch, oa), to_dev(torch, la), CAP, pad))
        assert (got == exp).all(), n
        if n in (3, 768, 1024, 5000, 16385):
            assert (to_host(H.sponge_hash_var(dp, to_dev(torch, oa), to_dev(torch, la), CAP, pad, sort=True)) == exp).all()
    # one long message (750 blocks): the chain of dependent permutations the low-latency form is for
    one = oracle.sponge_var(pool, np.array([0], dtype=np.uint64), np.array([2999], dtype=np.uint64), CAP, pad)
    assert (to_host(H.sponge_hash_var(dp, to_dev(torch, np.array([0], dtype=np.uint64)),
                                      to_dev(torch, np.array([2999], dtype=np.uint64)), CAP, pad)) == one).all()
    # fixed length, few messages
    for n, ln in ((1, 9), (7, 4), (770, 3), (1024, 1), (1025, 5), (4096, 3), (4097, 3), (16384, 2), (16385, 2)):
        msgs = oracle.gen_b(n + ln, n * ln)
        e = oracle.sponge(msgs, ln, CAP, pad)
        assert (to_host(H.sponge_hash(to_dev(torch, msgs).view(-1, 4), ln, CAP, pad)) == e).all(), (n, ln)
    # a message that does not lie inside the pool is hashed as the empty message and counted, never read
    la = np.array([4, 8, 4, 3000], dtype=np.uint64)
    oa = np.array([0, 2995, 3001, 1], dtype=np.uint64)                     # #1 runs past the end, #2 starts past it, #3 too long
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = torch.zeros((4, 4), dtype=torch.int64, device="cuda")
    cap = (ctypes.c_uint64 * 4)(*limbs_of(CAP))
    assert hades_lib.hades252_sponge_hash_var_dev(dp.data_ptr(), 3000, to_dev(torch, oa).data_ptr(),
                                                  to_dev(torch, la).data_ptr(), 4, cap, pad, out.data_ptr(),
                                                  bad.data_ptr(), None) == 0
    torch.cuda.synchronize()
    empty = oracle.sponge_var(pool, np.array([0], dtype=np.uint64), np.array([0], dtype=np.uint64), CAP, pad)
    good = oracle.sponge_var(pool, oa[:1], la[:1], CAP, pad)
    got = to_host(out).reshape(4, 4)
    assert int(bad.item()) == 3 and (got[0] == good).all() and all((got[i] == empty).all() for i in (1, 2, 3))


def test_small_batch_streaming_absorb(torch_cuda, H, oracle):
    torch = torch_cuda
    for n, t in ((1, 1), (1, 40), (3, 2), (4, 3), (767, 2), (769, 2), (1024, 1), (1025, 1), (1030, 3), (4095, 2), (4096, 1), (4097, 1),
                 (16384, 1), (16385, 1)):
        msgs = oracle.gen_b(31 * n + t, n * t * 4)
        exp = oracle.sponge(msgs, 4 * t, CAP, 0)
        st = H.SpongeStates(n, CAP)
        st.absorb(to_dev(torch, msgs).view(n, t, 4, 4))
        assert (to_host(st.squeeze()) == exp).all(), (n, t)
        # the whole state equals the per-lane kernel's (forced by a batch above the threshold sharing the first n states)
        if n <= 4:
            big = H.SpongeStates(20000, CAP)
            blocks = torch.zeros((20000, t, 4, 4), dtype=torch.int64, device="cuda")
            blocks[:n] = to_dev(torch, msgs).view(n, t, 4, 4)
            big.absorb(blocks)
            assert torch.equal(big.states[:n], st.states)


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


def test_device_entry_points_are_graph_capturable(torch_cuda, H, oracle):
    """The _dev entry points only enqueue work on the caller's stream (no allocation, no synchronisation, no host read-back),
    so a launch-bound chain -- here a whole 4^6-leaf tree (six dependent levels), a small sponge batch and an in-place
    permutation -- can be captured once in a hipGraph and replayed on new data."""
    torch = torch_cuda
    tag = TAG[4]
    n = 4 ** 6
    leaves = H.gen_b(n, "cuda")
    states = H.gen_b(5 * 100, "cuda").view(100, 5, 4)
    scratch = torch.empty(max(H._lib.lib().hades252_merkle_scratch_bytes(n, 4) // 8, 2), dtype=torch.int64, device="cuda")
    strat = H.ScalarStrategy()
    root_e = H.merkle_root(leaves, 4, tag, 1, scratch)                       # eager, also warms everything up
    dig_e = H.sponge_hash(leaves[:400], 4, CAP, 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            root_g = H.merkle_root(leaves, 4, tag, 1, scratch)
            dig_g = H.sponge_hash(leaves[:400], 4, CAP, 1)
            strat.perm(states)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(root_g, root_e) and torch.equal(dig_g, dig_e)
    # new data in the same buffers, replayed: equals the oracle
    fresh = oracle.gen_b(4242, n)
    leaves.copy_(to_dev(torch, fresh).view(-1, 4))
    st0 = oracle.gen_b(777, 500)
    states.copy_(to_dev(torch, st0).view(100, 5, 4))
    g.replay()
    torch.cuda.synchronize()
    assert (to_host(root_g) == oracle.merkle_tree(fresh, 4, tag, 1)[-1]).all()
    assert (to_host(dig_g) == oracle.sponge(fresh[: 400 * 4], 4, CAP, 1)).all()
    assert (to_host(states) == oracle.perm_batch(st0)).all()


# ---------------------------------------------------------------------------------------------
# the callers of perm on HOST memory, and device memory for callers without HIP bindings
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("chunk_bytes", [None, "4096", "100000"])
def test_host_merkle_root_and_sponge(torch_cuda, H, oracle, monkeypatch, chunk_bytes):
    """hades252_merkle_root / hades252_sponge_hash: host memory in, 32 bytes per tree / message out; chunked upload behind
    the hashing (tiny chunks force many slot reuses and ragged last chunks).  The chunk size is latched at first use, so
    the forced sizes run in child interpreters."""
    import subprocess, textwrap
    if chunk_bytes is not None:
        code = textwrap.dedent('''
            import os, sys
            sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, os.path.join(%r, "oracle"))
            import numpy as np
            from hades252_amd import strategy as H
            import hades_spec as S
            import oracle_lib
            o = oracle_lib.load()
            tag = S.to_mont(15); cap = S.to_mont(1 << 64)
            for arity, n in ((4, 1000), (3, 5000), (2, 777), (4, 4), (4, 5)):
                lv = o.gen_b(n, n)
                assert (H.merkle_root_host(lv, arity, S.to_mont(2 ** arity - 1)) == o.merkle_tree(lv, arity, S.to_mont(2 ** arity - 1), 1)[-1]).all(), (arity, n)
            for n, ln in ((1000, 3), (50, 40), (3, 1000)):
                m = o.gen_b(n + ln, n * ln)
                assert (H.sponge_hash_host(m, n, ln, cap, 1).reshape(-1) == o.sponge(m, ln, cap, 1)).all(), (n, ln)
            print("child ok")
        ''') % (ROOT, ROOT, ROOT)
        env = dict(os.environ, HADES252_HOST_CHUNK_BYTES=chunk_bytes)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "child ok" in r.stdout, r.stderr[-3000:]
        return
    for arity, n in ((4, 4 ** 8), (4, 100001), (3, 3 ** 9), (2, 2), (2, 3), (4, 4 ** 10 + 17), (2, 2 ** 20)):
        tag = TAG[arity]
        lv = oracle.gen_b(n + arity, n)
        depth = H.merkle_depth(n, arity)
        opad = oracle.merkle_empty_digests(arity, depth, S.to_mont(9), tag, 1)
        dev = to_dev(torch_cuda, lv).view(-1, 4)
        dpad = to_dev(torch_cuda, opad).view(depth, 4)
        assert (H.merkle_root_host(lv, arity, tag) == to_host(H.merkle_root(dev, arity, tag, 1))).all(), (arity, n)
        assert (H.merkle_root_host(lv, arity, tag, 3, pad=opad.reshape(depth, 4).copy()) ==
                to_host(H.merkle_root(dev, arity, tag, 3, pad=dpad))).all(), (arity, n)
        if n <= 100001:
            assert (H.merkle_root_host(lv, arity, tag, 1, pad=opad.reshape(depth, 4).copy()) ==
                    oracle.merkle_tree(lv, arity, tag, 1, opad)[-1]).all(), (arity, n)
    for n, ln, pad in ((1, 1, 1), (5, 0, 1), (1000, 7, 0), (70000, 4, 1), (3, 3000, 1), (1 << 18, 3, 1)):
        m = oracle.gen_b(3 * n + ln, n * ln)
        got = H.sponge_hash_host(m, n, ln, CAP, pad)
        if ln:
            exp = to_host(H.sponge_hash(to_dev(torch_cuda, m).view(-1, 4), ln, CAP, pad))
            assert (got.reshape(-1) == exp).all(), (n, ln)
        if ln == 0:                                                           # n empty messages
            z = np.zeros(n, dtype=np.uint64)
            assert (got.reshape(-1) == oracle.sponge_var(np.zeros(4, dtype=np.uint64), z, z, CAP, pad)).all()
        elif n * ln <= 300000:
            assert (got.reshape(-1) == oracle.sponge(m, ln, CAP, pad)).all(), (n, ln)
    # ragged messages in host memory (sorted on the device above 16 384 messages), one of them outside the pool
    rng = random.Random(31)
    for n in (1, 700, 5000, 40000):
        lens = [rng.choice([0, 1, 3, 4, 5, 9, 17, 40]) for _ in range(n)]
        pool = oracle.gen_b(n, sum(lens) + 8)
        la = np.array(lens, dtype=np.uint64)
        oa = (np.cumsum(la) - la).astype(np.uint64)
        if n >= 700:
            oa[5], la[5] = np.uint64(pool.size // 4 - 2), np.uint64(3)            # runs past the end of the pool
        got, bad = H.sponge_hash_var_host(pool, oa, la, CAP, 1)
        exp_l = la.copy()
        if n >= 700:
            exp_l[5] = 0
        assert bad == (1 if n >= 700 else 0) and (got.reshape(-1) == oracle.sponge_var(pool, oa, exp_l, CAP, 1)).all(), n
    with pytest.raises(Exception):
        H.merkle_root_host(oracle.gen_b(1, 1), 4, TAG[4])                         # one leaf is not a tree
    with pytest.raises(Exception):
        H.merkle_root_host(oracle.gen_b(1, 8), 5, TAG[4])


def test_device_memory_helpers_without_torch_allocations(hades_lib, oracle):
    """A caller with no HIP bindings: allocate, upload, permute on its own stream, download -- only through the library."""
    n = 3000
    inp = oracle.gen_b(606, 5 * n)
    out = np.zeros_like(inp)
    d, s = ctypes.c_void_p(), ctypes.c_void_p()
    assert hades_lib.hades252_dev_alloc(ctypes.byref(d), inp.nbytes) == 0 and d.value
    assert hades_lib.hades252_stream_create(ctypes.byref(s)) == 0 and s.value
    assert hades_lib.hades252_dev_upload(d, inp.ctypes.data_as(ctypes.c_void_p), inp.nbytes, s) == 0
    assert hades_lib.hades252_perm_batch_dev(d, n, s) == 0
    assert hades_lib.hades252_dev_download(out.ctypes.data_as(ctypes.c_void_p), d, out.nbytes, s) == 0
    assert hades_lib.hades252_stream_sync(s) == 0
    assert (out == oracle.perm_batch(inp)).all()
    assert hades_lib.hades252_stream_destroy(s) == 0 and hades_lib.hades252_dev_free(d) == 0
    assert hades_lib.hades252_dev_free(None) == 0 and hades_lib.hades252_stream_destroy(None) == 0
    assert hades_lib.hades252_dev_alloc(None, 16) == -1 and hades_lib.hades252_dev_alloc(ctypes.byref(d), 0) == -1
    assert hades_lib.hades252_dev_upload(None, inp.ctypes.data_as(ctypes.c_void_p), 16, None) == -1
    assert hades_lib.hades252_stream_sync(None) == 0


@pytest.mark.parametrize("workers", [1, 2, 3, 8, 16, 64])
def test_merkle_root_multi_workers_on_one_device(torch_cuda, hades_lib, H, oracle, workers):
    """The sub-tree sharding of SURVEY 8(e) behind the C ABI, with more workers than devices: the hipSetDevice threads, the
    sub-tree arithmetic (worker counts that are no power of the arity, more workers than sub-trees) and the final small tree."""
    for arity, k in ((4, 7), (2, 12), (3, 6), (4, 1), (2, 2)):
        n = arity ** k
        lv = oracle.gen_b(100 + n, n)
        exp = H.merkle_root_host(lv, arity, TAG[arity])
        assert (H.merkle_root_multi(lv, arity, TAG[arity], 1, workers, virtual=True) == exp).all(), (arity, k, workers)
    big = oracle.gen_b(5, 4 ** 10)                                  # >= 8 MiB: page-locked once for all workers
    assert (H.merkle_root_multi(big, 4, TAG[4], 3, workers, virtual=True) == H.merkle_root_host(big, 4, TAG[4], 3)).all()
    with pytest.raises(Exception):
        H.merkle_root_multi(oracle.gen_b(1, 100), 4, TAG[4], 1, workers, virtual=True)        # not a full tree
    ndev = hades_lib.hades252_device_count()
    if workers > ndev:
        with pytest.raises(Exception):
            H.merkle_root_multi(oracle.gen_b(1, 64), 4, TAG[4], 1, workers)                   # real devices only
