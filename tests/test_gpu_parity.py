"""GPU tier: the HIP path, called through the C ABI, against the CPU oracle -- bit exact."""
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
from oracle_lib import P, R, limbs_of, int_of, digest_ref  # noqa: E402

pytestmark = pytest.mark.gpu

KERNELS = [1, 2, 3, 4, 5]   # HADES252_KERNEL_LITERAL, _FAST (one state per lane), _COOP (five waves per state), _LANES (one
                            # state per wave, elements spread over 16-lane rows), _ROWS (one state per row, four per wave)


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
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64)).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint64).reshape(-1)


def kernel_available(hades_lib, torch, k):
    t = torch.zeros(20, dtype=torch.int64, device="cuda")
    return hades_lib.hades252_perm_batch_dev_ex(t.data_ptr(), 1, None, k) == 0


@pytest.mark.parametrize("kernel", KERNELS)
def test_single_perm_kats(torch_cuda, hades_lib, H, kat, kernel):
    """config 0/1: README-style single permutation, golden vectors."""
    torch = torch_cuda
    if not kernel_available(hades_lib, torch, kernel):
        pytest.skip("kernel %d not built" % kernel)
    strat = H.ScalarStrategy(kernel)
    for s in kat["single"]:
        st = to_dev(torch, sum([limbs_of(int(x, 16)) for x in s["in_mont"]], []))
        strat.perm(st)
        exp = sum([limbs_of(int(x, 16)) for x in s["out_mont"]], [])
        assert list(map(int, to_host(st))) == exp


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097])
def test_ragged_batches(torch_cuda, hades_lib, H, oracle, kernel, n):
    torch = torch_cuda
    if not kernel_available(hades_lib, torch, kernel):
        pytest.skip("kernel %d not built" % kernel)
    inp = oracle.gen_b(1000 * n, 5 * n)
    guard = np.full(40, 0xDEADBEEFCAFEF00D, dtype=np.uint64)
    buf = to_dev(torch, np.concatenate([guard, inp, guard]))
    view = buf[40:40 + 20 * n]
    H.ScalarStrategy(kernel).perm(view)
    got = to_host(buf)
    assert (got[:40] == guard).all() and (got[-40:] == guard).all(), "wrote outside the batch"
    assert (got[40:-40] == oracle.perm_batch(inp)).all()


def test_empty_batch(torch_cuda, H):
    t = torch_cuda.zeros(0, dtype=torch_cuda.int64, device="cuda")
    H.ScalarStrategy().perm(t)


def test_bad_length_rejected(torch_cuda, H):
    # reference: copy_from_slice panics for len != WIDTH (scalar.rs:48)
    t = torch_cuda.zeros(19, dtype=torch_cuda.int64, device="cuda")
    with pytest.raises(ValueError):
        H.ScalarStrategy().perm(t)


@pytest.mark.parametrize("kernel", KERNELS)
def test_batch_digests_golden(torch_cuda, hades_lib, H, kat, kernel):
    torch = torch_cuda
    if not kernel_available(hades_lib, torch, kernel):
        pytest.skip("kernel %d not built" % kernel)
    for name in ("gen_a", "gen_b"):
        n = kat[name]["n"]
        buf = H.gen_a(5 * n, "cuda") if name == "gen_a" else H.gen_b(5 * n, "cuda")
        assert hashlib.sha256(to_host(buf).tobytes()).hexdigest() == kat[name]["sha256_in"]
        H.ScalarStrategy(kernel).perm(buf)
        assert hashlib.sha256(to_host(buf).tobytes()).hexdigest() == kat[name]["sha256_out"]


@pytest.mark.parametrize("kernel", KERNELS)
def test_edge_values(torch_cuda, hades_lib, H, oracle, kernel):
    """0, 1, p-1, R, all-ones-ish limbs in every word position, plus random."""
    torch = torch_cuda
    if not kernel_available(hades_lib, torch, kernel):
        pytest.skip("kernel %d not built" % kernel)
    rng = random.Random(1)
    edge = [0, 1, 2, P - 1, P - 2, R, P - R, (1 << 255) % P, (1 << 254) - 1, 0xFFFFFFFF, P - (1 << 32),
            0xFFFFFFFF00000000, (P - 1) // 2, 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF]
    states = []
    for _ in range(2048):
        states.append([rng.choice(edge) if rng.random() < 0.7 else rng.randrange(P) for _ in range(5)])
    inp = np.array([l for st in states for v in st for l in limbs_of(v)], dtype=np.uint64)
    buf = to_dev(torch, inp)
    H.ScalarStrategy(kernel).perm(buf)
    assert (to_host(buf) == oracle.perm_batch(inp)).all()


@pytest.mark.parametrize("kernel", KERNELS)
def test_config2_2pow20_bit_exact(torch_cuda, hades_lib, H, oracle, kernel):
    """BASELINE config[1]: 2^20 independent permutations, every output bit compared."""
    torch = torch_cuda
    if not kernel_available(hades_lib, torch, kernel):
        pytest.skip("kernel %d not built" % kernel)
    n = 1 << 20
    buf = H.gen_b(5 * n, "cuda")
    inp = to_host(buf).copy()
    assert (inp[:20 * 4096] == oracle.gen_b(0, 5 * 4096)).all()
    H.ScalarStrategy(kernel).perm(buf)
    got = to_host(buf)
    exp = oracle.perm_batch(inp)
    assert (got == exp).all()


def test_per_op_kernels(torch_cuda, H, oracle):
    """Strategy::add_round_key / quintic_s_box / mul_matrix / apply_*_round vs the oracle."""
    torch = torch_cuda
    n = 777
    inp = oracle.gen_b(31337, 5 * n)
    strat = H.ScalarStrategy()
    for rnd in (0, 3, 4, 35, 62, 63, 66):
        buf = to_dev(torch, inp)
        it = H.RoundConstantsIter(5 * rnd)
        strat.add_round_key(it, buf)
        assert it.pos == 5 * rnd + 5
        assert (to_host(buf) == oracle.add_round_key(inp, rnd)).all()
        buf = to_dev(torch, inp)
        strat.apply_full_round(H.RoundConstantsIter(5 * rnd), buf)
        assert (to_host(buf) == oracle.full_round(inp, rnd)).all()
        buf = to_dev(torch, inp)
        strat.apply_partial_round(H.RoundConstantsIter(5 * rnd), buf)
        assert (to_host(buf) == oracle.partial_round(inp, rnd)).all()
    buf = to_dev(torch, inp)
    strat.quintic_s_box(buf)
    assert (to_host(buf) == oracle.quintic_s_box(inp)).all()
    buf = to_dev(torch, inp)
    strat.mul_matrix(H.RoundConstantsIter(), buf)
    assert (to_host(buf) == oracle.mul_matrix(inp)).all()
    with pytest.raises(RuntimeError, match="out of ARK constants"):
        strat.add_round_key(H.RoundConstantsIter(956), buf)


def test_perm_is_the_composition_of_rounds(torch_cuda, H, oracle):
    """The trait's provided perm (strategies.rs:140-157) replayed over the per-round entry
    points equals the fused kernel."""
    torch = torch_cuda
    inp = oracle.gen_b(5, 5 * 300)
    a, b = to_dev(torch, inp), to_dev(torch, inp)
    H.ScalarStrategy().perm(a)
    H.ScalarStrategy().perm_stepwise(b)
    assert torch.equal(a, b)


def test_kernels_agree_at_scale(torch_cuda, hades_lib, H):
    """Two independent device implementations, 2^22 permutations, digest of all outputs."""
    torch = torch_cuda
    if not kernel_available(hades_lib, torch, 2):
        pytest.skip("fast kernel not built")
    n = 1 << 22
    a = H.gen_b(5 * n, "cuda")
    b = a.clone()
    H.ScalarStrategy(1).perm(a)
    H.ScalarStrategy(2).perm(b)
    assert H.digest(a) == H.digest(b)
    assert torch.equal(a, b)


def test_split_invariance(torch_cuda, H):
    """Permuting a batch in one call or in ragged pieces gives the same bytes (no cross-lane state)."""
    torch = torch_cuda
    n = 100003
    a = H.gen_b(5 * n, "cuda")
    b = a.clone()
    H.ScalarStrategy().perm(a)
    flat = b.view(-1)
    cuts = [0, 1, 64, 1000, 65537, n]
    for lo, hi in zip(cuts, cuts[1:]):
        H.ScalarStrategy().perm(flat[20 * lo:20 * hi])
    assert torch.equal(a, b)


def test_bytes_wire_format(torch_cuda, hades_lib, H, oracle):
    torch = torch_cuda
    rng = random.Random(3)
    vals = [0, 1, P - 1, R] + [rng.randrange(P) for _ in range(996)]
    raw = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint64).copy()
    dev = to_dev(torch, raw)
    limbs = H.from_bytes(dev)
    got = to_host(limbs).reshape(-1, 4)
    for k in (0, 1, 2, 3, 500, 999):
        assert int_of(got[k]) == vals[k] * R % P
    back = H.to_bytes(limbs)
    assert (to_host(back) == raw).all()
    # host entry point: whole permutation on canonical bytes
    n = 200
    host = raw[:n * 20].copy()
    rc = hades_lib.hades252_perm_batch_bytes(host.ctypes.data_as(ctypes.c_void_p), n)
    assert rc == 0
    for i in (0, 7, 199):
        exp = S.perm(vals[5 * i:5 * i + 5])
        got_vals = [int_of(host[20 * i + 4 * w:20 * i + 4 * w + 4]) for w in range(5)]
        assert got_vals == exp
    # non-canonical input (>= p) is rejected and the buffer is left untouched
    bad = raw[:40].copy()
    bad[4:8] = np.array(limbs_of(P), dtype=np.uint64)
    keep = bad.copy()
    assert hades_lib.hades252_perm_batch_bytes(bad.ctypes.data_as(ctypes.c_void_p), 2) == -3
    assert (bad == keep).all()
    with pytest.raises(ValueError):
        H.from_bytes(to_dev(torch, bad))


def test_host_entry_points(torch_cuda, hades_lib, H, oracle):
    n = 5000
    inp = oracle.gen_b(99, 5 * n)
    exp = oracle.perm_batch(inp)
    a = inp.copy()
    H.ScalarStrategy().perm(a)                       # numpy -> hades252_perm_batch
    assert (a == exp).all()
    b = inp.copy()
    assert hades_lib.hades252_perm_batch_multi(b.ctypes.data_as(ctypes.c_void_p), n, 1) == 0
    assert (b == exp).all()
    ndev = hades_lib.hades252_device_count()
    assert ndev >= 1
    c = inp.copy()
    assert hades_lib.hades252_perm_batch_multi(c.ctypes.data_as(ctypes.c_void_p), n, 0) == 0
    assert (c == exp).all()
    assert hades_lib.hades252_perm_batch_multi(c.ctypes.data_as(ctypes.c_void_p), n, ndev + 1) == -1


def test_host_path_chunked(torch_cuda, hades_lib, H, oracle):
    """Several chunks: exercises the event-chained copy-in / kernel / copy-out pipeline (pageable caller memory)."""
    n = (1 << 18) * 2 + 12345
    buf = H.gen_b(5 * n, "cuda")
    inp = to_host(buf).copy()
    H.ScalarStrategy().perm(buf)
    host = inp.copy()
    H.ScalarStrategy().perm(host)
    assert (host == to_host(buf)).all()


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


def test_generators_and_digest(torch_cuda, H, oracle):
    a = H.gen_a(1000, "cuda", first_elem=17)
    assert (to_host(a) == oracle.gen_a(17, 1000)).all()
    b = H.gen_b(100001, "cuda", first_elem=12345)
    hb = to_host(b)
    assert (hb == oracle.gen_b(12345, 100001)).all()
    assert H.digest(b) == digest_ref(hb)
    assert H.digest(b, first_index=6) == digest_ref(hb, 6)
    # additivity over a split (how shards combine)
    flat = b.view(-1)
    cut = 4 * 5003
    d1, d2 = H.digest(flat[:cut], 0), H.digest(flat[cut:], cut)
    assert [(x + y) & 0xFFFFFFFFFFFFFFFF for x, y in zip(d1, d2)] == H.digest(b)


# ---- BASELINE full sizes: size-independent properties ----------------------------------------
def test_config3_2pow26_properties(torch_cuda, hades_lib, H, oracle):
    """BASELINE config[2]: 2^26 permutations on one GPU.  The oracle cannot replay 2^26, so:
    (1) a strided sample (every 2^10-th state, 65 536 states: SURVEY section 8(d) config 3) is compared bit for bit with the oracle,
    (2) the two independent device implementations agree on the digest of ALL outputs,
    (3) one call == two ragged calls (no cross-lane / cross-launch state)."""
    torch = torch_cuda
    n = 1 << 26
    a = H.gen_b(5 * n, "cuda")
    stride = 1 << 10
    sample_in = a.view(n, 20)[::stride].contiguous()
    host_in = to_host(sample_in).copy()
    assert (host_in[:20] == oracle.gen_b(0, 5)).all()
    H.ScalarStrategy(2).perm(a)
    got = to_host(a.view(n, 20)[::stride].contiguous())
    assert (got == oracle.perm_batch(host_in)).all()
    d_fast = H.digest(a)
    # literal kernel on the same inputs
    H.gen_b(5 * n, "cuda", out=a.view(-1, 4))
    H.ScalarStrategy(1).perm(a)
    assert H.digest(a) == d_fast
    # split invariance with the shipped kernel
    H.gen_b(5 * n, "cuda", out=a.view(-1, 4))
    flat = a.view(-1)
    cut = 20 * ((n // 3) + 7)
    H.ScalarStrategy(2).perm(flat[:cut])
    H.ScalarStrategy(2).perm(flat[cut:])
    assert H.digest(a) == d_fast


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


def test_config5_sharding_arithmetic(torch_cuda, H):
    """BASELINE config[4] decomposition on one device: 8 shards generated and permuted
    independently (global element offsets as bench.py computes them) combine, by digest
    addition, to the digest of the unsharded batch."""
    from hades252_amd import sharding
    n_total = 1 << 23
    whole = H.gen_b(5 * n_total, "cuda")
    H.ScalarStrategy().perm(whole)
    ref = H.digest(whole)
    acc = [0, 0, 0, 0]
    for rank in range(8):
        b, e = sharding.shard_range(rank, 8, n_total)
        shard = H.gen_b(5 * (e - b), "cuda", first_elem=5 * b)
        H.ScalarStrategy().perm(shard)
        d = H.digest(shard, first_index=20 * b)
        acc = [(x + y) & 0xFFFFFFFFFFFFFFFF for x, y in zip(acc, d)]
    assert acc == ref


def test_perm_trace(torch_cuda, H, oracle):
    """Per-round states (Gadget witness pre-computation) vs the oracle's trace."""
    torch = torch_cuda
    n = 300
    inp = oracle.gen_b(777, 5 * n)
    dev = to_dev(torch, inp)
    tr = H.perm_trace(dev)
    assert (to_host(dev) == inp).all()                      # input untouched
    host = tr.cpu().numpy().view(np.uint64).reshape(67, n, 20)
    for i in (0, 1, 63, 64, 299):
        _, otr = oracle.perm_trace(inp[20 * i:20 * i + 20])
        assert (host[:, i, :] == otr.reshape(67, 20)).all()
    out = dev.clone()
    H.ScalarStrategy().perm(out)
    assert torch.equal(tr[66].reshape(-1), out.reshape(-1))


def test_host_path_concurrent_threads(torch_cuda, hades_lib, oracle):
    """The library is re-entrant (the reference strategy is a stateless ZST): many host threads
    permuting their own buffers at once, small and chunked sizes mixed."""
    import threading
    sizes = [1, 7, 300, 5000, (1 << 18) + 77, 64, 1000, 3]
    bufs = [oracle.gen_b(1000 * i, 5 * n) for i, n in enumerate(sizes)]
    exp = [oracle.perm_batch(b) for b in bufs]
    rcs = [None] * len(sizes)

    def work(i):
        for _ in range(2 if sizes[i] > 10000 else 6):
            x = bufs[i].copy()
            rcs[i] = hades_lib.hades252_perm_batch(x.ctypes.data_as(ctypes.c_void_p), sizes[i])
            if rcs[i] != 0 or not (x == exp[i]).all():
                rcs[i] = -99
                return

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(sizes))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert rcs == [0] * len(sizes)


def test_merkle_sharded_emulated(torch_cuda, H, oracle):
    """Multi-GPU Merkle decomposition emulated on one device: every 'rank' builds
    its sub-tree roots, the gathered sub-roots are finished, result == single-device root."""
    torch = torch_cuda
    from hades252_amd import merkle
    tag = S.to_mont(15)
    n = 1 << 16
    leaves = H.gen_b(n, "cuda")
    ref = H.merkle4_root(leaves, tag, 1)
    for world in (1, 2, 4, 8):
        per_rank = n // world
        parts = [merkle.local_subroots(leaves[r * per_rank:(r + 1) * per_rank], n, world, tag, 1)
                 for r in range(world)]
        root = merkle.finish_from_subroots(torch.cat(parts), tag, 1)
        assert torch.equal(root.view(-1), ref.view(-1))
    assert torch.equal(merkle.merkle4_root_sharded(leaves, n, tag, 1).view(-1), ref.view(-1))


def test_per_op_kernels_edge_values(torch_cuda, H, oracle):
    """Field-operation edge cases through the per-op kernels: operands 0, 1, p-1, p-2, R, values
    that make word + constant cross p, etc. (saturated 8x32 arithmetic of csrc/fr32.hpp)."""
    torch = torch_cuda
    rng = random.Random(77)
    edge = [0, 1, 2, P - 1, P - 2, R, P - R, (1 << 255) % P, P - (1 << 32), 0xFFFFFFFF, 0xFFFFFFFF00000000,
            (P - 1) // 2, (P + 1) // 2, (1 << 224) - 1]
    ark = S.round_constants()
    # complements of the first round constants (Montgomery domain): w + c == 0, p - 1, 1 (mod p)
    for c in ark[:10]:
        cm = S.to_mont(c)
        edge += [(P - cm) % P, (P - cm - 1) % P, (P - cm + 1) % P]
    n = 1024
    vals = [rng.choice(edge) if rng.random() < 0.8 else rng.randrange(P) for _ in range(5 * n)]
    inp = np.array([l for v in vals for l in limbs_of(v)], dtype=np.uint64)
    strat = H.ScalarStrategy()
    for rnd in (0, 1, 33, 66):
        buf = to_dev(torch, inp)
        strat.add_round_key(H.RoundConstantsIter(5 * rnd), buf)
        assert (to_host(buf) == oracle.add_round_key(inp, rnd)).all()
    buf = to_dev(torch, inp)
    strat.quintic_s_box(buf)
    assert (to_host(buf) == oracle.quintic_s_box(inp)).all()
    buf = to_dev(torch, inp)
    strat.mul_matrix(H.RoundConstantsIter(), buf)
    assert (to_host(buf) == oracle.mul_matrix(inp)).all()
    # wire format on the same edge values
    canon = np.frombuffer(b"".join(S.from_mont(v).to_bytes(32, "little") for v in vals), dtype=np.uint64).copy()
    assert (to_host(H.to_bytes(to_dev(torch, inp))) == canon).all()
    assert (to_host(H.from_bytes(to_dev(torch, canon))) == inp).all()


def test_misaligned_device_pointer_rejected(torch_cuda, hades_lib):
    t = torch_cuda.zeros(64, dtype=torch_cuda.int64, device="cuda")
    assert hades_lib.hades252_perm_batch_dev(t.data_ptr() + 8, 1, None) == -1
    assert hades_lib.hades252_perm_batch_dev(t.data_ptr() + 32, 1, None) == 0


def test_exhaustive_edge_tuples(torch_cuda, H, oracle):
    """Every 5-tuple over 14 edge values (0, 1, p-1, R, 2^255 mod p, all-ones limbs, ...) = 537 824
    states, shipped kernel vs the CPU oracle, all bits."""
    import itertools
    torch = torch_cuda
    edge = [0, 1, 2, P - 1, P - 2, R, P - R, (1 << 255) % P, (1 << 254) - 1, 0xFFFFFFFF, P - (1 << 32),
            0xFFFFFFFF00000000, (P - 1) // 2, (1 << 128) - 1]
    tab = np.array([limbs_of(v) for v in edge], dtype=np.uint64)
    idx = np.array(list(itertools.product(range(len(edge)), repeat=5)), dtype=np.int64)
    inp = np.ascontiguousarray(tab[idx]).reshape(-1)
    buf = to_dev(torch, inp)
    H.ScalarStrategy(2).perm(buf)
    assert (to_host(buf) == oracle.perm_batch(inp)).all()


def test_sponge_hash(torch_cuda, H, oracle):
    """Batched fixed-length sponge over the permutation vs the oracle (convention parameters;
    dusk-poseidon itself is outside the reference tree)."""
    torch = torch_cuda
    cap = S.to_mont(1 << 64)
    for length in (1, 2, 3, 4, 5, 7, 8, 9, 16):
        for pad in (0, 1):
            n = 1000 if length < 9 else 130
            msgs = oracle.gen_b(length * 977 + pad, n * length)
            got = H.sponge_hash(to_dev(torch, msgs), length, cap, pad)
            assert (to_host(got) == oracle.sponge(msgs, length, cap, pad)).all(), (length, pad)
    with pytest.raises(ValueError):
        H.sponge_hash(to_dev(torch, oracle.gen_b(0, 10)), 3, cap, 1)
