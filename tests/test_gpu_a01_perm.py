"""GPU tier, SURVEY section 8 rows a1 / a8 / a12 (+ the configs of section 8(d)): `Strategy::perm` through the C ABI -- known
answers, ragged batches with guard words, edge values, 2^20 full compares for all five kernels, the default dispatch and
its thresholds, kernels against each other at scale, BASELINE configs[1], [2] and [4] at full size, the multi-launch loop.
Under `pytest -x` this file runs first: a failure here is a failure of the hot path itself."""
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
def test_hades_det_like_reference(torch_cuda, hades_lib, H, kernel):
    """The reference's own test of this path, src/strategies/scalar.rs:62-74, on the device: perm([17; 5]) twice gives the
    same state, and differs from perm([19; 5]).  (The reference pins nothing more; the known answers are in
    test_single_perm_kats.)"""
    torch = torch_cuda
    def perm(start):
        x = scalars_dev(torch, [S.to_mont(start)] * 5).view(-1)            # perm(start): BlsScalar::from(u64)
        H.ScalarStrategy(kernel).perm(x)
        return to_host(x)
    x, y, z = perm(17), perm(17), perm(19)
    assert (x == y).all() and (x != z).any()
    assert int_of(x[:4]) == S.to_mont(S.perm([17] * 5)[0])


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
# the helped lane-split form under timing disturbance (the exchange is double-buffered by round parity, so
# correctness no longer depends on the peer finishing its read within one S-box)
# ---------------------------------------------------------------------------------------------
def test_lanes_helped_form_is_timing_independent(torch_cuda, H, oracle):
    torch = torch_cuda
    from hades252_amd import _lib
    big = H.gen_b(5 << 20, "cuda")
    side = torch.cuda.Stream()
    n = 768                                                       # helped form, one block per CU
    inp = oracle.gen_b(4040, 5 * n)
    exp = oracle.perm_batch(inp)
    bufs = [to_dev(torch, inp) for _ in range(40)]
    torch.cuda.synchronize()
    with torch.cuda.stream(side):                                 # a throughput kernel hogging every SIMD beside them
        for _ in range(3):
            H.ScalarStrategy(_lib.KERNEL_FAST).perm(big)
    for b in bufs:
        H.ScalarStrategy(_lib.KERNEL_LANES).perm(b)
    torch.cuda.synchronize()
    for b in bufs:
        assert (to_host(b) == exp).all()
    # chains in the helped form (sponge: 30 dependent permutations per message) beside the same disturbance
    msgs = oracle.gen_b(77, 500 * 119)
    dexp = oracle.sponge(msgs, 119, CAP, 1)
    dm = to_dev(torch, msgs).view(-1, 4)
    with torch.cuda.stream(side):
        H.ScalarStrategy(_lib.KERNEL_FAST).perm(big)
    got = [H.sponge_hash(dm, 119, CAP, 1) for _ in range(4)]
    torch.cuda.synchronize()
    for g in got:
        assert (to_host(g) == dexp).all()


# ---------------------------------------------------------------------------------------------
# one dispatch rule, exported
# ---------------------------------------------------------------------------------------------
def test_dispatch_rule_is_exported_and_consistent(torch_cuda, H, hades_lib, oracle):
    from hades252_amd import _lib
    names = {_lib.KERNEL_LITERAL: "k_states_literal", _lib.KERNEL_FAST: "k_perm_fast", _lib.KERNEL_COOP: "k_perm_coop",
             _lib.KERNEL_LANES: "k_perm_lanes", _lib.KERNEL_ROWS: "k_perm_rows"}
    for k, nm in names.items():
        assert H.kernel_name(k, 12345) == nm
    assert hades_lib.hades252_kernel_name(99, 1) is None
    assert H.kernel_for(1) == _lib.KERNEL_LANES and H.kernel_for(1 << 26) == _lib.KERNEL_FAST
    # monotone: the selector sequence over growing n never returns to an earlier form
    order, last = [], None
    for n in [1, 2, 700, 768, 769, 1024, 1025, 4096, 4097, 8192, 16384, 16385, 32768, 65536, 65537, 1 << 20]:
        k = H.kernel_for(n)
        assert k in names and k != _lib.KERNEL_LITERAL
        assert H.kernel_name(0, n) == names[k] and H.chain_form_for(n) in names
        if k != last:
            order.append(k)
            last = k
    assert len(order) == len(set(order)) and order[0] == _lib.KERNEL_LANES and order[-1] == _lib.KERNEL_FAST
    # the default dispatch and the forced selector it reports give the same bits (and the oracle's) around every switch
    sizes = sorted({1, 768, 769, 1024, 1025, 4096, 4097, 16384, 16385, 20000})
    for n in sizes:
        inp = oracle.gen_b(31 * n, 5 * n)
        a, b = to_dev(torch_cuda, inp), to_dev(torch_cuda, inp)
        H.ScalarStrategy().perm(a)
        H.ScalarStrategy(H.kernel_for(n)).perm(b)
        exp = oracle.perm_batch(inp)
        assert (to_host(a) == exp).all() and (to_host(b) == exp).all(), n


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


def test_misaligned_device_pointer_rejected(torch_cuda, hades_lib):
    t = torch_cuda.zeros(64, dtype=torch_cuda.int64, device="cuda")
    assert hades_lib.hades252_perm_batch_dev(t.data_ptr() + 8, 1, None) == -1
    assert hades_lib.hades252_perm_batch_dev(t.data_ptr() + 32, 1, None) == 0


# ---- BASELINE full sizes: size-independent properties ----------------------------------------
def test_config3_2pow26_properties(torch_cuda, hades_lib, H, oracle, kat):
    """BASELINE config[2]: 2^26 permutations on one GPU.  (0) the digest of ALL outputs equals the one the oracle computed over
    the same 2^26 states (25 min of CPU, once: tools/oracle_block_digests.py), and, independent of that record:
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
    # (0) ALL 2^26 outputs against the oracle: its committed digest of the same states (kat.json headline_2p26_blocks, block 0)
    assert ["%016x" % x for x in d_fast] == kat["headline_2p26_blocks"]["blocks"]["0"]
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


def test_config5_one_rank_shard_2pow27(torch_cuda, H, oracle, kat):
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
    assert ["%016x" % x for x in d] == kat["config5_2p30"]["rank7_shard_digest"]
    _record("config5_r2.txt", "rank7_shard 2^27 perms first_perm=%d sample=%d states bit-exact vs oracle; digest %s"
            % (b, len(idx), " ".join("%016x" % x for x in d)))


def test_config5_whole_2pow30_on_one_device(torch_cuda, H, oracle, kat):
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
    # ... and the committed record bench.py checks an 8-GPU run against (tests/golden/kat.json config5_2p30)
    assert ["%016x" % x for x in ref] == kat["config5_2p30"]["digest"] == kat["config5_2p30"]["oracle_digest"]
    _record("config5_r2.txt", "whole 2^30 perms on one device: digest %s == sum of 8 shard digests; %d sampled states "
            "bit-exact vs oracle" % (" ".join("%016x" % x for x in ref), len(idx)))


# ---------------------------------------------------------------------------------------------
# the multi-launch loop of hades252_perm_batch_dev_ex (more than 2^30 states per call in production): its second and
# later trips, on a batch of a few thousand states, with the per-launch cap lowered by the test hook
# ---------------------------------------------------------------------------------------------
_MULTI_LAUNCH_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import torch
import oracle_lib
from hades252_amd import build, strategy as H
build.build(verbose=False)
orc = oracle_lib.load()
cap = int(os.environ["HADES252_TEST_MAX_LAUNCH"])
GUARD = 0x5A5A5A5A5A5A5A5A
for n in (3 * cap + 17, 2 * cap, cap + 1, cap, cap - 1):
    inp = orc.gen_b(5 * 77, 5 * n)
    exp = orc.perm_batch(inp)
    for k in (1, 2, 3, 4, 5):
        buf = np.full(20 * n + 40, GUARD, dtype=np.uint64)
        buf[20:20 + 20 * n] = inp
        t = torch.from_numpy(buf.view(np.int64)).cuda()
        H.ScalarStrategy(k).perm(t[20:20 + 20 * n])
        got = t.cpu().numpy().view(np.uint64)
        assert (got[:20] == GUARD).all() and (got[-20:] == GUARD).all(), ("guard words", n, k)
        bad = np.nonzero((got[20:-20] != exp).reshape(-1, 20).any(axis=1))[0]
        assert bad.size == 0, ("kernel %d, n %d: first wrong state %d (launch boundary every %d)" % (k, n, bad[0], cap))
print("MULTI_LAUNCH_OK")
"""


def test_multi_launch_loop_with_lowered_cap(torch_cuda, hades_lib):
    """the `for (off = 0; off < n; off += cap)` loop had only ever run one trip (2^30 states are exactly
    one launch).  HADES252_TEST_MAX_LAUNCH = 4096 (read once by the library, hence a child process): n = 3 cap + 17, 2 cap,
    cap + 1, cap, cap - 1, all five kernels, guard words on both sides, every state against the oracle."""
    import subprocess
    env = dict(os.environ, HADES252_TEST_MAX_LAUNCH="4096")
    r = subprocess.run([sys.executable, "-c", _MULTI_LAUNCH_CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MULTI_LAUNCH_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
