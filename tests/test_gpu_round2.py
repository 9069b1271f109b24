"""GPU tier, round 2: blob KATs through the device arithmetic, free constant cursor, fast trace,
variable-length sponge, and BASELINE config 5 at its real size.  Everything goes through the C ABI."""
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
from test_blob_kat import ARK_SHA256, MDS_SHA256  # noqa: E402

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
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64)).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint64).reshape(-1)


def scalars_dev(torch, ints):
    return to_dev(torch, np.array([l for v in ints for l in limbs_of(v)], dtype=np.uint64)).view(-1, 4)


# ---------------------------------------------------------------------------------------------
# Pin #0 on the device: both device arithmetics regenerate the reference's blobs
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl", [0, 1])
def test_mds_blob_through_device_field_ops(torch_cuda, H, impl):
    """assets/mds.bin (written by the real dusk-bls12_381, HOWTO.md:71-108) regenerated with the DEVICE
    field arithmetic: x = from(i) + from(j+5) via from_raw and add, x^(p-2) by ~380 device squarings /
    products.  impl 0 = fr32.hpp (literal kernels), impl 1 = to_f29 / mont_fips / finalize (shipped kernel)."""
    torch = torch_cuda
    xs = H.fr_op(H.FR_FROM_RAW, scalars_dev(torch, [i for i in range(5) for _ in range(5)]), impl=impl)
    ys = H.fr_op(H.FR_FROM_RAW, scalars_dev(torch, [j + 5 for _ in range(5) for j in range(5)]), impl=impl)
    x = H.fr_op(H.FR_ADD, xs, ys, impl=impl)
    acc = x
    for bit in bin(P - 2)[3:]:
        acc = H.fr_op(H.FR_SQUARE, acc, impl=impl)
        if bit == "1":
            acc = H.fr_op(H.FR_MUL, acc, x, impl=impl)
    blob = to_host(acc).tobytes()
    assert hashlib.sha256(blob).hexdigest() == MDS_SHA256
    # x * x^-1 == one
    one = to_host(H.fr_op(H.FR_MUL, acc, x, impl=impl)).reshape(-1, 4)
    assert all(int_of(r) == R for r in one)


@pytest.mark.parametrize("impl", [0, 1])
def test_ark_blob_through_device_field_ops(torch_cuda, H, impl):
    """assets/ark.bin (HOWTO.md:21-48): from_bytes_wide(SHA-512 chain) = lo*R^2 + hi*R^3 and the running
    sum, all on the device arithmetic (the sum as a 10-step scan of batched adds)."""
    torch = torch_cuda
    data, lo, hi = b"poseidon-for-plonk", [], []
    for _ in range(960):
        data = hashlib.sha512(data).digest()
        lo.append(int.from_bytes(data[:32], "little"))
        hi.append(int.from_bytes(data[32:], "little"))
    # lo / hi are arbitrary 256-bit integers (possibly >= p), exactly what the crate's from_u512 multiplies
    r2 = H.fr_op(H.FR_FROM_RAW, scalars_dev(torch, [R]), impl=impl)               # R * R^2 / R
    assert int_of(to_host(r2)) == R * R % P
    r3 = H.fr_op(H.FR_SQUARE, r2, impl=impl)                                      # R^4 / R
    wide = H.fr_op(H.FR_ADD,
                   H.fr_op(H.FR_MUL, scalars_dev(torch, lo), r2.expand(960, 4).contiguous(), impl=impl),
                   H.fr_op(H.FR_MUL, scalars_dev(torch, hi), r3.expand(960, 4).contiguous(), impl=impl), impl=impl)
    # inclusive prefix sums (Hillis-Steele), then + one
    acc, d = wide, 1
    while d < 960:
        nxt = acc.clone()
        nxt[d:] = H.fr_op(H.FR_ADD, acc[d:].contiguous(), acc[:-d].contiguous(), impl=impl)
        acc, d = nxt, 2 * d
    one = H.fr_op(H.FR_FROM_RAW, scalars_dev(torch, [1]), impl=impl)
    acc = H.fr_op(H.FR_ADD, acc, one.expand(960, 4).contiguous(), impl=impl)
    assert hashlib.sha256(to_host(acc).tobytes()).hexdigest() == ARK_SHA256


def test_fr_ops_vs_oracle_edge_values(torch_cuda, H, oracle):
    torch = torch_cuda
    rng = random.Random(5)
    edge = [0, 1, 2, P - 1, P - 2, R, P - R, (1 << 255) % P, (1 << 254) - 1, 0xFFFFFFFF, P - (1 << 32),
            0xFFFFFFFF00000000, (P - 1) // 2, (P + 1) // 2, (1 << 128) - 1]
    a = [rng.choice(edge) if rng.random() < 0.6 else rng.randrange(P) for _ in range(700)]
    b = [rng.choice(edge) if rng.random() < 0.6 else rng.randrange(P) for _ in range(700)]
    da, db = scalars_dev(torch, a), scalars_dev(torch, b)
    for impl in (0, 1):
        got = to_host(H.fr_op(H.FR_ADD, da, db, impl=impl)).reshape(-1, 4)
        assert [int_of(r) for r in got] == [(x + y) % P for x, y in zip(a, b)]
        got = to_host(H.fr_op(H.FR_MUL, da, db, impl=impl)).reshape(-1, 4)
        assert [int_of(r) for r in got] == [oracle.fr2("mul", x, y) for x, y in zip(a, b)]
        got = to_host(H.fr_op(H.FR_SQUARE, da, impl=impl)).reshape(-1, 4)
        assert [int_of(r) for r in got] == [oracle.fr1("square", x) for x in a]
        got = to_host(H.fr_op(H.FR_FROM_RAW, da, impl=impl)).reshape(-1, 4)
        assert [int_of(r) for r in got] == [x * R % P for x in a]


# ---------------------------------------------------------------------------------------------
# the trait's constants iterator: any cursor, all 960 constants
# ---------------------------------------------------------------------------------------------
def test_free_cursor_all_constants(torch_cuda, H, oracle, hades_lib):
    torch = torch_cuda
    inp = oracle.gen_b(31337, 5 * 200)
    strat = H.ScalarStrategy()
    for cur in (0, 1, 3, 7, 334, 335, 336, 700, 955):
        buf = to_dev(torch, inp)
        it = H.RoundConstantsIter(cur)
        strat.add_round_key(it, buf)
        assert it.pos == cur + 5
        assert (to_host(buf) == oracle.add_round_key_at(inp, cur)).all(), cur
    for cur in (2, 336, 951):
        buf = to_dev(torch, inp)
        strat.apply_full_round(H.RoundConstantsIter(cur), buf)
        assert (to_host(buf) == oracle.full_round_at(inp, cur)).all(), cur
        buf = to_dev(torch, inp)
        strat.apply_partial_round(H.RoundConstantsIter(cur), buf)
        assert (to_host(buf) == oracle.partial_round_at(inp, cur)).all(), cur
    # every one of the 960 constants: state of zeros + constants = the table itself
    zeros = torch.zeros((192, 5, 4), dtype=torch.int64, device="cuda")
    for r in range(192):
        strat.add_round_key(H.RoundConstantsIter(5 * r), zeros[r])
    table = to_host(zeros).reshape(960, 4)
    assert [int_of(t) for t in table] == [oracle.round_constant(i) for i in range(960)]
    # exhaustion: the reference panics "Hades252 out of ARK constants" (src/strategies.rs:40)
    buf = to_dev(torch, inp)
    with pytest.raises(RuntimeError, match="out of ARK constants"):
        strat.add_round_key(H.RoundConstantsIter(956), buf)
    assert hades_lib.hades252_add_round_key_at_dev(buf.data_ptr(), 200, 956, None) == -6
    assert hades_lib.hades252_apply_full_round_dev(buf.data_ptr(), 200, 192, None) == -6
    assert hades_lib.hades252_add_round_key_at_dev(buf.data_ptr(), 200, -1, None) == -1
    assert (to_host(buf) == inp).all()


# ---------------------------------------------------------------------------------------------
# per-round trace: shipped (scale-tracked) kernel == literal kernel == oracle
# ---------------------------------------------------------------------------------------------
def test_perm_trace_fast_vs_literal_vs_oracle(torch_cuda, H, oracle):
    torch = torch_cuda
    rng = random.Random(9)
    edge = [0, 1, P - 1, R, P - R, (1 << 254) - 1, 0xFFFFFFFF]
    n = 1500
    vals = [rng.choice(edge) if rng.random() < 0.3 else rng.randrange(P) for _ in range(5 * n)]
    inp = np.array([l for v in vals for l in limbs_of(v)], dtype=np.uint64)
    dev = to_dev(torch, inp)
    fast = H.perm_trace(dev, kernel=2)
    lit = H.perm_trace(dev, kernel=1)
    assert (to_host(dev) == inp).all()
    assert torch.equal(fast, lit)
    host = fast.cpu().numpy().view(np.uint64).reshape(67, n, 20)
    for i in (0, 1, 63, 64, 777, n - 1):
        _, otr = oracle.perm_trace(inp[20 * i:20 * i + 20])
        assert (host[:, i, :] == otr.reshape(67, 20)).all()


def test_perm_trace_fast_2pow16_digest(torch_cuda, H):
    """Round-major trace of 2^16 states: last slice == perm output; literal and fast agree by digest."""
    torch = torch_cuda
    n = 1 << 16
    st = H.gen_b(5 * n, "cuda")
    fast = H.perm_trace(st, kernel=2)
    lit = H.perm_trace(st, kernel=1)
    assert H.digest(fast) == H.digest(lit)
    out = st.clone()
    H.ScalarStrategy().perm(out)
    assert torch.equal(fast[66].reshape(-1), out.reshape(-1))


def test_perm_witness_all_gadget_wires(torch_cuda, H, oracle):
    """hades252_perm_witness_dev: all 972 gate outputs of the reference's GadgetStrategy per state
    (src/strategies/gadget.rs:41-133) vs the spec oracle's restatement of that schedule, on edge and random
    states; then batch-wide identities on 5 000 states: r2 of the last round == perm output, and every S-box
    triple satisfies v4 == v2^2 through the device field ops."""
    torch = torch_cuda
    rng = random.Random(17)
    cases = [[5000] * 5, [0] * 5, [P - 1] * 5, [1, 2, 3, 4, 5]] + [[rng.randrange(P) for _ in range(5)] for _ in range(6)]
    n_pad = 70                                  # more than one wave, ragged
    vals = cases + [[rng.randrange(P) for _ in range(5)] for _ in range(n_pad - len(cases))]
    inp = np.array([l for st in vals for v in st for l in limbs_of(S.to_mont(v))], dtype=np.uint64)
    dev = to_dev(torch, inp)
    wires = H.perm_witness(dev)
    assert tuple(wires.shape) == (972, n_pad, 4)
    assert (to_host(dev) == inp).all()                       # input untouched
    host = wires.cpu().numpy().view(np.uint64).reshape(972, n_pad, 4)
    for i in list(range(len(cases))) + [63, 64, 69]:
        spec = []
        S.perm_gadget(vals[i], spec)
        got = [int_of(host[g, i]) for g in range(972)]
        bad = [g for g in range(972) if got[g] != S.to_mont(spec[g])]
        assert not bad, (i, bad[:8])
    # batch-wide identities
    n = 5000
    st = H.gen_b(5 * n, "cuda")
    w = H.perm_witness(st)
    out = st.clone()
    H.ScalarStrategy().perm(out)
    last = torch.stack([w[962 + 2 * j + 1] for j in range(5)], dim=1)       # r2[j] of round 66
    assert torch.equal(last.reshape(-1), out.reshape(-1))
    for g in (5, 8, 17, 20 + 10, 5 + 15 + 10 + 15 + 10):                    # some v2 wires (rounds 0, 0, 0, 1, 2)
        v2, v4 = w[g].contiguous(), w[g + 1].contiguous()
        assert torch.equal(H.fr_op(H.FR_SQUARE, v2), v4)


def test_witness_rows_equal_trace_plus_next_round_key(torch_cuda, H, oracle):
    """Two independent kernels at scale: for every round r and word j, the gadget's row wire r2[r][j] must equal the
    per-round trace state + the NEXT round's constant (src/strategies/gadget.rs:102-129), on 2^14 states."""
    torch = torch_cuda
    n = 1 << 14
    st = H.gen_b(5 * n, "cuda", first_elem=12345)
    wires = H.perm_witness(st)
    trace = H.perm_trace(st)                                   # [67, n, 5, 4]
    base = 5                                                   # wires of round 0's key additions
    for r in range(67):
        full = r < 4 or r >= 63
        base += 15 if full else 3                              # S-box wires of this round
        for j in range(5):
            row = wires[base + 2 * j + 1]
            state = trace[r, :, j, :].contiguous()
            if r < 66:
                c = scalars_dev(torch, [oracle.round_constant(5 * (r + 1) + j)]).expand(n, 4).contiguous()
                state = H.fr_op(H.FR_ADD, state, c)
            assert torch.equal(row, state), (r, j)
        base += 10
    assert base == 972


def test_witness_every_gate_identity_at_scale(torch_cuda, H, oracle):
    """All 972 wires of 4 096 states against the GATES themselves (src/strategies/gadget.rs:59-69, :102-129), evaluated by
    the device field ops on the kernel's own outputs: every S-box triple (v2 = v v, v4 = v2 v2, v5 = v4 v with v the wire
    that feeds it), every r1 = M[j][0] z0 + M[j][1] z1 + M[j][2] z2 and r2 = r1 + M[j][3] z3 + M[j][4] z4 + c.  Together
    with the first five wires (input + round key) this pins every wire to the input by induction."""
    torch = torch_cuda
    n = 1 << 12
    st = H.gen_b(5 * n, "cuda", first_elem=99)
    w = H.perm_witness(st)
    mul = lambda a, b: H.fr_op(H.FR_MUL, a.contiguous(), b.contiguous())
    add = lambda a, b: H.fr_op(H.FR_ADD, a.contiguous(), b.contiguous())
    const = lambda v: scalars_dev(torch, [S.to_mont(v)]).expand(n, 4).contiguous()   # v: canonical integer
    mds = [[const(v) for v in row] for row in S.mds_matrix()]
    ark = S.round_constants()
    state = []
    for j in range(5):                                                          # wires 0..4: input + first round key
        state.append(w[j])
        assert torch.equal(w[j], add(st.view(n, 5, 4)[:, j, :], const(ark[j]))), j
    g = 5
    for r in range(67):
        full = r < 4 or r >= 63
        z = list(state)
        for word in (range(5) if full else (4,)):
            v = state[word]
            assert torch.equal(w[g], mul(v, v)) and torch.equal(w[g + 1], mul(w[g], w[g])), (r, word)
            assert torch.equal(w[g + 2], mul(w[g + 1], v)), (r, word)
            z[word] = w[g + 2]
            g += 3
        nxt = []
        for j in range(5):
            r1 = add(add(mul(mds[j][0], z[0]), mul(mds[j][1], z[1])), mul(mds[j][2], z[2]))
            assert torch.equal(w[g], r1), (r, j)
            r2 = add(add(mul(mds[j][3], z[3]), mul(mds[j][4], z[4])), w[g])
            if r < 66:
                r2 = add(r2, const(ark[5 * (r + 1) + j]))
            assert torch.equal(w[g + 1], r2), (r, j)
            nxt.append(w[g + 1])
            g += 2
        state = nxt
    assert g == 972


# ---------------------------------------------------------------------------------------------
# variable-length sponge
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pad", [0, 1])
def test_sponge_var_ragged_lengths(torch_cuda, H, oracle, pad):
    """Ragged lengths 0..33 (every residue mod 4, zero-length messages, one long outlier in a wave of
    short ones), shuffled offsets, gaps and overlaps -- vs the oracle, both padding modes."""
    torch = torch_cuda
    rng = random.Random(11 + pad)
    cap = S.to_mont((1 << 64) + 7)
    n = 1000
    lengths = [rng.randrange(0, 34) for _ in range(n)]
    lengths[5] = 0
    lengths[64:128] = [1] * 63 + [33]          # a wave of short messages with one long one
    lengths[300:364] = [0] * 64                # a wave of empty messages
    pool = oracle.gen_b(4242, 40000)
    offsets = [rng.randrange(0, 40000 - 34) for _ in range(n)]      # arbitrary: overlaps and gaps
    got = H.sponge_hash_var(to_dev(torch, pool), to_dev(torch, np.array(offsets, dtype=np.uint64)),
                            to_dev(torch, np.array(lengths, dtype=np.uint64)), cap, pad)
    exp = oracle.sponge_var(pool, offsets, lengths, cap, pad)
    assert (to_host(got) == exp).all()


def test_sponge_var_equals_fixed_and_packed(torch_cuda, H, oracle):
    torch = torch_cuda
    cap = S.to_mont(1 << 64)
    n, length = 777, 6
    msgs = oracle.gen_b(99, n * length)
    fixed = H.sponge_hash(to_dev(torch, msgs), length, cap, 1)
    off = np.arange(n, dtype=np.uint64) * np.uint64(length)
    var = H.sponge_hash_var(to_dev(torch, msgs), to_dev(torch, off), to_dev(torch, np.full(n, length, dtype=np.uint64)),
                            cap, 1)
    assert torch.equal(fixed, var)
    assert (to_host(fixed) == oracle.sponge(msgs, length, cap, 1)).all()
    # packed ragged (CSR-style offsets)
    lens = np.array([(i * 7) % 13 for i in range(500)], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    pool = oracle.gen_b(5, int(lens.sum()) + 1)
    got = H.sponge_hash_var(to_dev(torch, pool), to_dev(torch, offs), to_dev(torch, lens), cap, 1)
    assert (to_host(got) == oracle.sponge_var(pool, offs, lens, cap, 1)).all()
    # a message reaching outside the pool is never read: counted, raised by the mirror
    lens_bad = lens.copy()
    lens_bad[7] = np.uint64(1 << 40)
    with pytest.raises(IndexError):
        H.sponge_hash_var(to_dev(torch, pool), to_dev(torch, offs), to_dev(torch, lens_bad), cap, 1)
    offs_bad = offs.copy()
    offs_bad[9] = np.uint64((1 << 64) - 3)                      # offset + length would wrap around
    with pytest.raises(IndexError):
        H.sponge_hash_var(to_dev(torch, pool), to_dev(torch, offs_bad), to_dev(torch, lens), cap, 1)


# ---------------------------------------------------------------------------------------------
# low-latency kernel (five waves per state) and the dispatch on batch size
# ---------------------------------------------------------------------------------------------
def test_coop_kernel_and_default_dispatch(torch_cuda, hades_lib, H, oracle):
    """HADES252_KERNEL_COOP == HADES252_KERNEL_FAST == oracle on ragged sizes around the block size (64) and the
    dispatch threshold (2^14), with guard words; DEFAULT must give the same bits on both sides of the threshold."""
    torch = torch_cuda
    for n in (1, 5, 63, 64, 65, 127, 128, 129, 1000, (1 << 14) - 1, 1 << 14, (1 << 14) + 1, 40000):
        inp = oracle.gen_b(7 * n, 5 * n)
        guard = np.full(40, 0xDEADBEEFCAFEF00D, dtype=np.uint64)
        exp = oracle.perm_batch(inp)
        for kernel in (3, 0):
            buf = to_dev(torch, np.concatenate([guard, inp, guard]))
            H.ScalarStrategy(kernel).perm(buf[40:40 + 20 * n])
            got = to_host(buf)
            assert (got[:40] == guard).all() and (got[-40:] == guard).all(), "wrote outside the batch"
            assert (got[40:-40] == exp).all(), (n, kernel)


def test_coop_kernel_2pow20_vs_fast(torch_cuda, H):
    torch = torch_cuda
    a = H.gen_b(5 << 20, "cuda")
    b = a.clone()
    H.ScalarStrategy(2).perm(a)
    H.ScalarStrategy(3).perm(b)
    assert torch.equal(a, b)


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


# ---------------------------------------------------------------------------------------------
# BASELINE config 5 at its real size
# ---------------------------------------------------------------------------------------------
def _record(name, text):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "a") as f:
        f.write(text + "\n")


def test_config5_one_rank_shard_2pow27(torch_cuda, H, oracle):
    """BASELINE config[4]: 2^30 permutations over 8 GPUs = 2^27 (20 GiB) per GPU.  This is rank 7's shard
    exactly as bench.py generates it (global element offsets of rank 7), permuted in one call, then a strided
    sample (every 2^14-th state + the first and last 64) compared bit for bit with the CPU oracle."""
    torch = torch_cuda
    from hades252_amd import sharding
    n_total, world, rank = 1 << 30, 8, 7
    b, e = sharding.shard_range(rank, world, n_total)
    n = e - b
    assert n == 1 << 27
    st = torch.empty((n, 5, 4), dtype=torch.int64, device="cuda")
    H.gen_b(5 * n, "cuda", first_elem=5 * b, out=st.view(-1, 4))
    idx = np.unique(np.concatenate([np.arange(0, n, 1 << 14), np.arange(64), np.arange(n - 64, n)]))
    tidx = torch.from_numpy(idx).cuda()
    inp = st[tidx].cpu().numpy().view(np.uint64).reshape(-1)
    # the sampled inputs are what generator B defines for those global indices
    for k in (0, 1, len(idx) // 2, len(idx) - 1):
        assert (inp[20 * k:20 * k + 20] == oracle.gen_b(5 * (b + int(idx[k])), 5)).all()
    H.ScalarStrategy().perm(st)
    got = st[tidx].cpu().numpy().view(np.uint64).reshape(-1)
    assert (got == oracle.perm_batch(inp)).all()
    d = H.digest(st, first_index=20 * b)
    _record("config5_r2.txt", "rank7_shard 2^27 perms first_perm=%d sample=%d states bit-exact vs oracle; digest %s"
            % (b, len(idx), " ".join("%016x" % x for x in d)))


def test_config5_whole_2pow30_on_one_device(torch_cuda, H, oracle):
    """The whole 2^30-permutation config on ONE device (160 GiB of the 288 GB), two launches inside the
    library.  Size-independent properties: (1) split invariance -- the digest of the whole batch equals
    the wrapping sum of the 8 shard digests computed from independently generated + permuted shards;
    (2) a strided oracle sample across the whole range."""
    torch = torch_cuda
    from hades252_amd import sharding
    free, _ = torch.cuda.mem_get_info()
    n_total = 1 << 30
    if free < n_total * 160 + (22 << 30):
        pytest.skip("needs %d GiB free HBM" % ((n_total * 160 + (22 << 30)) >> 30))
    whole = torch.empty((n_total, 5, 4), dtype=torch.int64, device="cuda")
    H.gen_b(5 * n_total, "cuda", out=whole.view(-1, 4))
    idx = np.unique(np.concatenate([np.arange(0, n_total, 1 << 17), np.arange(n_total - 64, n_total),
                                    np.arange((1 << 30) - (1 << 29) - 32, (1 << 30) - (1 << 29) + 32)]))
    tidx = torch.from_numpy(idx).cuda()
    inp = whole[tidx].cpu().numpy().view(np.uint64).reshape(-1)
    H.ScalarStrategy().perm(whole)
    got = whole[tidx].cpu().numpy().view(np.uint64).reshape(-1)
    assert (got == oracle.perm_batch(inp)).all()
    ref = H.digest(whole)
    del whole
    torch.cuda.empty_cache()
    acc = [0, 0, 0, 0]
    for rank in range(8):
        b, e = sharding.shard_range(rank, 8, n_total)
        shard = torch.empty((e - b, 5, 4), dtype=torch.int64, device="cuda")
        H.gen_b(5 * (e - b), "cuda", first_elem=5 * b, out=shard.view(-1, 4))
        H.ScalarStrategy().perm(shard)
        d = H.digest(shard, first_index=20 * b)
        acc = [(x + y) & 0xFFFFFFFFFFFFFFFF for x, y in zip(acc, d)]
        del shard
    assert acc == ref
    _record("config5_r2.txt", "whole 2^30 perms on one device: digest %s == sum of 8 shard digests; %d sampled states "
            "bit-exact vs oracle" % (" ".join("%016x" % x for x in ref), len(idx)))
