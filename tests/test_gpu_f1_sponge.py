"""GPU tier, SURVEY section 8 row f1: the sponge hash over the batched permutation (fixed and variable length, device-side
sort by block count, streaming absorb / squeeze, the one-message-per-wave forms).  Convention (capacity, padding) is a
parameter: dusk-poseidon is outside the reference tree -- UNPINNED."""
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


def test_sponge_golden_vectors_on_the_device(torch_cuda, H, oracle, kat):
    """The committed sponge vectors (tests/golden/kat.json `sponge`, the ones rust/tests/kat_scalar.rs hands to the real
    crate's `perm`) through the HIP path: one ragged batch per (capacity, padding rule)."""
    torch = torch_cuda
    vecs = kat["sponge"]["vectors"]
    groups = {}
    for v in vecs:
        groups.setdefault((v["capacity"], v["pad_mode"]), []).append(v)
    assert len(groups) == 4
    for (cap_hex, pad), vs in groups.items():
        pool = np.concatenate([oracle.gen_b(v["first_elem"], v["len"]) for v in vs] + [np.zeros(4, dtype=np.uint64)])
        lengths = [v["len"] for v in vs]
        offsets = [sum(lengths[:i]) for i in range(len(vs))]
        got = H.sponge_hash_var(to_dev(torch, pool), to_dev(torch, np.array(offsets, dtype=np.uint64)),
                                to_dev(torch, np.array(lengths, dtype=np.uint64)), S.to_mont(int(cap_hex, 16)), pad)
        host = to_host(got).reshape(-1, 4)
        for i, v in enumerate(vs):
            assert int_of(host[i]) == int(v["digest_mont"], 16), v
