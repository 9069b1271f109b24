"""GPU tier, round 4: BASELINE configs pinned to the oracle's golden roots at full size, the failure contract of the
host-pointer entry points under injected faults, the bounded pool of pipes, the library's one dispatch rule, and the
helped lane-split kernel under timing disturbance.  Everything goes through the C ABI."""
import json
import os
import sys
import random
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hades_spec as S  # noqa: E402
from oracle_lib import int_of  # noqa: E402

pytestmark = pytest.mark.gpu

TAG4 = S.to_mont(15)
CAP = S.to_mont(1 << 64)


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


def hex_of(t):
    return hex(int_of(to_host(t)))


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


def test_host_and_sharded_roots_equal_golden(torch_cuda, H, kat, oracle):
    """hades252_merkle_root (host memory, chunked upload) and hades252_merkle_root_multi (8 and 16 virtual workers: the
    8-GPU decomposition on one device) on 2^20 leaves against the committed root."""
    n = 4 ** 10
    gold = kat["merkle4_full_size"][str(n)]["root"]
    leaves = oracle.gen_b(0, n)
    assert hex(int_of(H.merkle_root_host(leaves, 4, TAG4, 1))) == gold
    for w in (8, 16, 3):
        assert hex(int_of(H.merkle_root_multi(leaves, 4, TAG4, 1, n_workers=w, virtual=True))) == gold


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


# ---------------------------------------------------------------------------------------------
# failure contract of the host-pointer entry points (include/hades252.h), under injected faults
# ---------------------------------------------------------------------------------------------
def each_state_is_input_or_output(got, inp, exp):
    g, i, e = got.reshape(-1, 20), inp.reshape(-1, 20), exp.reshape(-1, 20)
    is_in, is_out = (g == i).all(axis=1), (g == e).all(axis=1)
    return bool((is_in | is_out).all()), int(is_out.sum())


SITES_PERM = ["malloc", "hostmalloc", "hostregister", "memcpy", "streamcreate", "eventcreate", "sync"]


@pytest.mark.parametrize("n", [1, 200, 3000, 70000, 300000])
def test_perm_batch_under_injected_faults(torch_cuda, H, hades_lib, oracle, n):
    """Every wrapped HIP call of hades252_perm_batch fails once, in turn (nth = 1, 2, ... until the call no longer reaches
    that occurrence): negative return code (or success where the failing call is optional: page-locking), every state of
    the caller's buffer is its input or its output, the next call succeeds with the right bits, and the pool does not grow
    without bound."""
    inp = oracle.gen_b(17 * n + 3, 5 * n)
    exp = oracle.perm_batch(inp)
    lib = hades_lib
    failures = 0
    for site in SITES_PERM:
        for nth in range(1, 40):
            H.trim()                                            # a fresh pipe: the creation calls are reached again
            buf = inp.copy()
            H.fault_inject("%s:%d" % (site, nth))
            rc = lib.hades252_perm_batch(buf.ctypes.data, n)
            H.fault_inject(None)
            if rc == 0:
                assert (buf == exp).all(), (site, nth)
                if site != "hostregister":
                    break                                       # the nth occurrence does not exist: done with this site
                if nth >= 2:
                    break
                continue
            failures += 1
            assert rc == -2 and lib.hades252_last_hip_error() != 0, (site, nth, rc)
            ok, done = each_state_is_input_or_output(buf, inp, exp)
            assert ok, (site, nth)
            if n <= 65536:
                assert done in (0, n), "a one-chunk call leaves the buffer untouched or completely written (%s:%d)" % (site, nth)
            # the next call works and is right
            again = inp.copy()
            assert lib.hades252_perm_batch(again.ctypes.data, n) == 0
            assert (again == exp).all(), (site, nth)
    assert failures >= 3
    assert H.pool_bytes() <= (1 << 30)


def test_big_pageable_batch_goes_through_staging_threads(torch_cuda, H, hades_lib, oracle):
    """A big batch in ORDINARY memory (more than two 2^16-state chunks): helper threads copy it chunk by chunk into
    page-locked staging buffers and the results back (perm_batch_host_staged) -- the caller's pages are never locked.
    Right bits from an unaligned base address and ragged sizes; nothing is locked afterwards; a failing copy, event wait
    or staging allocation leaves whole states and a working library."""
    lib = hades_lib
    for n in (131073, 600000, 4 * 65536, 4 * 65536 + 1, 5 * 65536 - 1):
        inp = oracle.gen_b(12345 + n, 5 * n)
        exp = oracle.perm_batch(inp)
        buf = np.empty(5 * 4 * n + 3, dtype=np.uint64)[3:]         # pageable, base address = 8 (mod 32)
        buf[:] = inp
        assert buf.ctypes.data % 32 != 0
        assert lib.hades252_perm_batch(buf.ctypes.data, n) == 0
        assert (buf == exp).all(), n
        assert not H.host_is_pinned(buf)
    n = 600000                                                     # ten chunks
    inp = oracle.gen_b(777, 5 * n)
    exp = oracle.perm_batch(inp)
    buf = np.empty(5 * 4 * n + 3, dtype=np.uint64)[3:]
    failures = 0
    for spec in ("hostmalloc:1", "memcpy:1", "memcpy:4", "memcpy:9", "memcpy:17", "memcpy:20", "sync:1", "sync:3", "sync:11"):
        if spec.startswith("hostmalloc"):
            H.trim()                                               # a fresh pipe: the staging buffer is allocated again
        buf[:] = inp
        H.fault_inject(spec)
        rc = lib.hades252_perm_batch(buf.ctypes.data, n)
        H.fault_inject(None)
        assert rc in (0, -2), spec
        failures += rc != 0
        ok, done = each_state_is_input_or_output(buf, inp, exp)
        assert ok, spec
        assert rc != 0 or done == n
        assert done % 65536 == 0 or done == n, "results come back in whole chunks (%s: %d)" % (spec, done)
        buf[:] = inp
        assert lib.hades252_perm_batch(buf.ctypes.data, n) == 0 and (buf == exp).all(), spec
    assert failures >= 6
    # the canonical-bytes entry point takes the same road (wire conversions on the device around the permutation)
    k = 200000
    torch = torch_cuda
    b = to_host(H.to_bytes(to_dev(torch, inp[: 20 * k]).view(-1, 4))).view(np.uint8).copy()
    assert lib.hades252_perm_batch_bytes(b.ctypes.data, k) == 0
    want = to_host(H.to_bytes(to_dev(torch, exp[: 20 * k]).view(-1, 4))).view(np.uint8)
    assert (b == want).all()
    # ... and rejects a batch with ONE value >= p deep inside it up front (the check runs on several threads), untouched
    b2 = to_host(H.to_bytes(to_dev(torch, inp[: 20 * k]).view(-1, 4))).view(np.uint8).copy()
    pos = 32 * (5 * k - 12345)
    b2[pos:pos + 32] = np.frombuffer(S.P.to_bytes(32, "little"), dtype=np.uint8)
    keep = b2.copy()
    assert lib.hades252_perm_batch_bytes(b2.ctypes.data, k) == -3 and (b2 == keep).all()


def test_host_callers_under_injected_faults(torch_cuda, H, hades_lib, oracle):
    """hades252_merkle_root / _sponge_hash / _sponge_hash_var: a failing call returns a code, leaves the root untouched,
    and the same call then succeeds."""
    lib = hades_lib
    n = 40000
    leaves = oracle.gen_b(5, n)
    root_exp = oracle.merkle_tree(leaves, 4, TAG4, 1)[-1]
    m = oracle.gen_b(99, 3000 * 6)
    dig_exp = oracle.sponge(m, 6, CAP, 1)
    for site in ("malloc", "memcpy", "sync", "streamcreate", "eventcreate"):
        for nth in (1, 2, 3):
            H.trim()
            root = np.full(4, 0xABCDEF, dtype=np.uint64)
            H.fault_inject("%s:%d" % (site, nth))
            rc = lib.hades252_merkle_root(leaves.ctypes.data, n, 4, H._tag_arr(TAG4), 1, None, root.ctypes.data)
            H.fault_inject(None)
            if rc != 0:
                assert rc == -2 and (root == 0xABCDEF).all(), (site, nth)
            else:
                assert (root == root_exp).all()
            assert (H.merkle_root_host(leaves, 4, TAG4, 1) == root_exp).all()
            H.trim()
            H.fault_inject("%s:%d" % (site, nth))
            dig = np.zeros(3000 * 4, dtype=np.uint64)
            rc = lib.hades252_sponge_hash(m.ctypes.data, 3000, 6, H._tag_arr(CAP), 1, dig.ctypes.data)
            H.fault_inject(None)
            assert rc in (0, -2)
            if rc == 0:
                assert (dig == dig_exp).all()
            assert (H.sponge_hash_host(m, 3000, 6, CAP, 1).reshape(-1) == dig_exp).all()


def test_big_inputs_of_the_one_shot_callers_travel_through_staging(torch_cuda, H, hades_lib, oracle, kat):
    """Merkle root / sponge / variable-length sponge on big inputs in ORDINARY memory (>= 8 MiB): uploaded through the
    staging threads (StagedSource), nothing of the caller's page-locked; right results, also with ragged last chunks and
    from an unaligned base; a failing copy or event wait gives a return code, leaves the outputs alone where the header
    says so, and the next call is right."""
    lib = hades_lib
    n = 4 ** 10 + 4 ** 9 + 12345                                 # 42 MiB of leaves, ragged tree, ragged chunks
    base = oracle.gen_b(9, n + 1)
    leaves = base[4:]                                            # base address = 32 B past an allocation start
    want = oracle.merkle_tree(leaves, 4, TAG4, 1)[-1]
    assert (H.merkle_root_host(leaves, 4, TAG4, 1) == want).all()
    assert not H.host_is_pinned(leaves)
    assert hex(int_of(H.merkle_root_host(oracle.gen_b(0, 4 ** 10), 4, TAG4, 1))) == kat["merkle4_full_size"][str(4 ** 10)]["root"]
    fails = 0
    for spec in ("memcpy:1", "memcpy:2", "memcpy:3", "sync:1", "sync:2", "hostmalloc:1"):
        if spec.startswith("hostmalloc"):
            H.trim()
        root = np.full(4, 0xABCDEF, dtype=np.uint64)
        H.fault_inject(spec)
        rc = lib.hades252_merkle_root(leaves.ctypes.data, n, 4, H._tag_arr(TAG4), 1, None, root.ctypes.data)
        H.fault_inject(None)
        assert rc in (0, -2), spec
        fails += rc != 0
        assert (root == (want if rc == 0 else 0xABCDEF)).all(), spec
        assert (H.merkle_root_host(leaves, 4, TAG4, 1) == want).all(), spec
    assert fails >= 4
    # fixed-length sponge: 2^18 messages of 5 scalars (40 MiB)
    nm, ln = 1 << 18, 5
    msgs = oracle.gen_b(31, nm * ln)
    dig = oracle.sponge(msgs, ln, CAP, 1)
    assert (H.sponge_hash_host(msgs, nm, ln, CAP, 1).reshape(-1) == dig).all()
    # variable-length sponge: a 24 MiB pool, messages of 0 .. 9 scalars anywhere in it
    rng = np.random.default_rng(5)
    pool = oracle.gen_b(77, 750000)
    lens = rng.integers(0, 10, size=60000).astype(np.uint64)
    offs = rng.integers(0, 750000 - 10, size=60000).astype(np.uint64)
    got, bad = H.sponge_hash_var_host(pool, offs, lens, CAP, 1)
    assert bad == 0 and (np.asarray(got).reshape(-1) == oracle.sponge_var(pool, offs, lens, CAP, 1)).all()
    for spec in ("memcpy:2", "sync:1"):
        H.fault_inject(spec)
        out = np.zeros(4 * 60000, dtype=np.uint64)
        rc = lib.hades252_sponge_hash_var(pool.ctypes.data, 750000, offs.ctypes.data, lens.ctypes.data, 60000, H._tag_arr(CAP), 1,
                                          out.ctypes.data, None)
        H.fault_inject(None)
        assert rc in (0, -2), spec
        got, bad = H.sponge_hash_var_host(pool, offs, lens, CAP, 1)
        assert bad == 0 and (np.asarray(got).reshape(-1) == oracle.sponge_var(pool, offs, lens, CAP, 1)).all()


def test_multi_entry_points_on_ordinary_memory(torch_cuda, H, hades_lib, oracle):
    """hades252_perm_batch_multi_ex / hades252_merkle_root_multi on a big buffer in ORDINARY memory: no registration of
    the whole buffer any more; every worker stages its own shard (shards share boundary pages).  2, 3 and 8 virtual
    workers on this one device."""
    n = 900000                                                   # 144 MB: 8 shards of 18 MB -> each through its staging threads
    inp = oracle.gen_b(2024, 5 * n)
    exp = oracle.perm_batch(inp)
    for w in (2, 3, 8):
        buf = np.empty(20 * n + 1, dtype=np.uint64)[1:]
        buf[:] = inp
        H.perm_multi(buf, w, virtual=True)
        assert (buf == exp).all(), w
        assert not H.host_is_pinned(buf)


def test_concurrent_big_callers_on_ordinary_memory(torch_cuda, H, hades_lib, oracle):
    """Four host threads, each with its own big batch in ordinary memory, at the same time: every call gets its own pipe,
    staging buffer and helper threads; results right, the pool stays within its budget and keeps at most two staging
    buffers per device."""
    n = 300000
    inps = [oracle.gen_b(100 + t, 5 * n) for t in range(4)]
    exps = [oracle.perm_batch(x) for x in inps]
    bufs = [x.copy() for x in inps]
    rcs = [None] * 4

    def work(t):
        rcs[t] = hades_lib.hades252_perm_batch(bufs[t].ctypes.data, n)
    for _ in range(2):
        for t in range(4):
            bufs[t][:] = inps[t]
        ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for th in ts:
            th.start()
        for th in ts:
            th.join()
        assert rcs == [0] * 4
        for t in range(4):
            assert (bufs[t] == exps[t]).all(), t
    assert H.pool_bytes() <= (1 << 30)


def test_multi_worker_failure_is_reported_and_survivable(torch_cuda, H, hades_lib, oracle):
    """One worker of hades252_perm_batch_multi_ex cannot select its device: the call reports it, the other workers' shards
    are whole states (input or output), nothing hangs, and the next call is right."""
    lib = hades_lib
    n = 50000
    inp = oracle.gen_b(1, 5 * n)
    exp = oracle.perm_batch(inp)
    for nth in (1, 3, 8):
        buf = inp.copy()
        H.fault_inject("worker:%d" % nth)
        rc = lib.hades252_perm_batch_multi_ex(buf.ctypes.data, n, 8, 1)
        H.fault_inject(None)
        assert rc == -2
        ok, done = each_state_is_input_or_output(buf, inp, exp)
        assert ok and done == n - (n * 8 // 8 - n * 7 // 8), (nth, done)      # exactly one shard of 8 stayed behind
        again = inp.copy()
        assert lib.hades252_perm_batch_multi_ex(again.ctypes.data, n, 8, 1) == 0 and (again == exp).all()
    # merkle_root_multi: same hook
    lv = oracle.gen_b(0, 4 ** 8)
    gold = oracle.merkle_tree(lv, 4, TAG4, 1)[-1]
    root = np.full(4, 7, dtype=np.uint64)
    H.fault_inject("worker:2")
    rc = lib.hades252_merkle_root_multi(lv.ctypes.data, 4 ** 8, 4, H._tag_arr(TAG4), 1, 4, 1, root.ctypes.data)
    H.fault_inject(None)
    assert rc == -2 and (root == 7).all()
    assert (H.merkle_root_multi(lv, 4, TAG4, 1, n_workers=4, virtual=True) == gold).all()


def test_helper_thread_that_cannot_start_is_an_error_not_a_crash(torch_cuda, H, hades_lib, oracle):
    """The system refuses a helper thread (hook site `thread`): staging copies of a big pageable batch, the staged upload of
    a Merkle call, one worker of the _multi entry points, one slice of the canonical-bytes check.  A return code (or, for
    the check, the slice done on the calling thread), whole states, no hang, and the next call is right."""
    lib = hades_lib
    n = 300000
    inp = oracle.gen_b(4242, 5 * n)
    exp = oracle.perm_batch(inp)
    for nth in (1, 2, 6):
        buf = inp.copy()
        H.fault_inject("thread:%d" % nth)
        rc = lib.hades252_perm_batch(buf.ctypes.data, n)
        H.fault_inject(None)
        assert rc == -2 and lib.hades252_last_hip_error() != 0, nth
        ok, _ = each_state_is_input_or_output(buf, inp, exp)
        assert ok, nth
        assert lib.hades252_perm_batch(buf.ctypes.data, n) in (0,)      # (a half-done buffer permuted again: only the rc counts)
        again = inp.copy()
        assert lib.hades252_perm_batch(again.ctypes.data, n) == 0 and (again == exp).all(), nth
    for nth in (1, 5):                                                   # _multi: the nth worker never starts
        buf = inp.copy()
        H.fault_inject("thread:%d" % nth)
        rc = lib.hades252_perm_batch_multi_ex(buf.ctypes.data, n, 8, 1)
        H.fault_inject(None)
        assert rc == -2, nth
        ok, done = each_state_is_input_or_output(buf, inp, exp)
        assert ok and done < n, (nth, done)
    leaves = oracle.gen_b(3, 4 ** 9 + 77)                                # 8 MiB + : staged upload
    want = oracle.merkle_tree(leaves, 4, TAG4, 1)[-1]
    for nth in (1, 3):
        root = np.full(4, 0xABCDEF, dtype=np.uint64)
        H.fault_inject("thread:%d" % nth)
        rc = lib.hades252_merkle_root(leaves.ctypes.data, 4 ** 9 + 77, 4, H._tag_arr(TAG4), 1, None, root.ctypes.data)
        H.fault_inject(None)
        assert rc == -2 and (root == 0xABCDEF).all(), nth
        assert (H.merkle_root_host(leaves, 4, TAG4, 1) == want).all()
    k = 60000                                                            # 300 000 scalars: the check runs on several threads
    b = to_host(H.to_bytes(to_dev(torch_cuda, inp[: 20 * k]).view(-1, 4))).view(np.uint8).copy()
    H.fault_inject("thread:2")
    rc = lib.hades252_perm_batch_bytes(b.ctypes.data, k)
    H.fault_inject(None)
    want_b = to_host(H.to_bytes(to_dev(torch_cuda, exp[: 20 * k]).view(-1, 4))).view(np.uint8)
    assert rc == 0 and (b == want_b).all()


def test_warm_up_prepays_the_first_call(torch_cuda, H, hades_lib, oracle):
    """hades252_warm_up(hint): the pool holds the pipe a batch of that size takes (device chunk buffers; staging buffers
    for the pageable path), the next call of that size allocates nothing more, results are right; a failing allocation
    during warm-up is a return code and leaves nothing behind."""
    lib = hades_lib
    H.trim()
    assert H.pool_bytes() == 0
    H.warm_up(0)
    assert H.pool_bytes() == 0                                    # a small call's pipe has no device chunk buffers
    n = 300000
    H.warm_up(n)
    held = H.pool_bytes()
    assert held > 0
    inp = oracle.gen_b(31337, 5 * n)
    buf = inp.copy()
    assert lib.hades252_perm_batch(buf.ctypes.data, n) == 0
    assert (buf == oracle.perm_batch(inp)).all()
    assert H.pool_bytes() == held                                 # the warmed pipe served the call
    H.trim()
    H.fault_inject("malloc:1")
    rc = lib.hades252_warm_up(n)
    H.fault_inject(None)
    assert rc == -2 and H.pool_bytes() == 0
    H.warm_up(n)
    assert H.pool_bytes() == held


def test_concurrent_callers_with_faults_flying(torch_cuda, H, hades_lib, oracle):
    """Chaos run of the host boundary: four threads call hades252_perm_batch with batches of every path's size (staging
    buffer, one chunk, chunk pipeline, staging threads) while the main thread keeps arming the fault hook at random sites
    and trims the pool under their feet.  Whichever call a fault lands in: return code 0 or -2, every state its input or
    its output (all outputs when 0), no hang, no crash -- and afterwards, hook disarmed, every size is right again."""
    lib = hades_lib
    sizes = [1, 200, 3000, 70000, 140000, 300000]
    data = {n: (oracle.gen_b(1000 + n, 5 * n),) for n in sizes}
    data = {n: (inp, oracle.perm_batch(inp)) for n, (inp,) in data.items()}
    stop = threading.Event()
    problems, calls, failures = [], [0], [0]

    def worker(seed):
        rng = random.Random(seed)
        while not stop.is_set():
            n = rng.choice(sizes)
            inp, exp = data[n]
            buf = inp.copy()
            rc = lib.hades252_perm_batch(buf.ctypes.data, n)
            calls[0] += 1
            if rc == 0:
                if not (buf == exp).all():
                    problems.append(("wrong result with rc 0", n))
            elif rc == -2:
                failures[0] += 1
                ok, _ = each_state_is_input_or_output(buf, inp, exp)
                if not ok:
                    problems.append(("torn state", n))
            else:
                problems.append(("return code", rc, n))

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    rng = random.Random(99)
    t_end = time.time() + float(os.environ.get("HADES252_CHAOS_SECONDS", "12"))
    sites = ["malloc", "hostmalloc", "memcpy", "streamcreate", "eventcreate", "sync", "thread", "hostregister"]
    while time.time() < t_end:
        H.fault_inject("%s:%d" % (rng.choice(sites), rng.randint(1, 12)))
        time.sleep(rng.random() * 0.02)
        if rng.random() < 0.2:
            H.trim()
    H.fault_inject(None)
    stop.set()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive(), "a caller hangs"
    assert not problems, problems[:5]
    assert calls[0] > 50 and failures[0] > 5, (calls, failures)
    for n in sizes:                                                       # hook disarmed: everything works again
        inp, exp = data[n]
        buf = inp.copy()
        assert lib.hades252_perm_batch(buf.ctypes.data, n) == 0 and (buf == exp).all(), n
    assert H.pool_bytes() <= (1 << 30)


def test_concurrent_one_shot_callers_with_faults_flying(torch_cuda, H, hades_lib, oracle):
    """The same chaos for the one-shot callers: three threads build Merkle roots (resident-size and staged uploads) and hash
    sponge batches from host memory while faults are armed at random and the pool is trimmed.  A root is written on success
    only and is then right; digests of a successful call are right; nothing hangs; afterwards everything works."""
    lib = hades_lib
    small = oracle.gen_b(5, 40000)
    big = oracle.gen_b(6, 4 ** 9 + 77)                                    # > 8 MiB: staged upload
    roots = {40000: oracle.merkle_tree(small, 4, TAG4, 1)[-1], 4 ** 9 + 77: oracle.merkle_tree(big, 4, TAG4, 1)[-1]}
    leaves = {40000: small, 4 ** 9 + 77: big}
    msgs = oracle.gen_b(99, 3000 * 6)
    dig_exp = oracle.sponge(msgs, 6, CAP, 1)
    stop = threading.Event()
    problems, calls, failures = [], [0], [0]

    def worker(seed):
        rng = random.Random(seed)
        while not stop.is_set():
            calls[0] += 1
            if rng.random() < 0.6:
                n = rng.choice(list(roots))
                root = np.full(4, 0xABCDEF, dtype=np.uint64)
                rc = lib.hades252_merkle_root(leaves[n].ctypes.data, n, 4, H._tag_arr(TAG4), 1, None, root.ctypes.data)
                if rc == 0 and not (root == roots[n]).all():
                    problems.append(("wrong root", n))
                if rc != 0 and not (root == 0xABCDEF).all():
                    problems.append(("root written by a failing call", n, rc))
            else:
                dig = np.zeros(3000 * 4, dtype=np.uint64)
                rc = lib.hades252_sponge_hash(msgs.ctypes.data, 3000, 6, H._tag_arr(CAP), 1, dig.ctypes.data)
                if rc == 0 and not (dig == dig_exp).all():
                    problems.append(("wrong digests",))
            if rc not in (0, -2):
                problems.append(("return code", rc))
            failures[0] += rc != 0

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(3)]
    for t in threads:
        t.start()
    rng = random.Random(7)
    t_end = time.time() + float(os.environ.get("HADES252_CHAOS_SECONDS", "10"))
    sites = ["malloc", "hostmalloc", "memcpy", "streamcreate", "eventcreate", "sync", "thread"]
    while time.time() < t_end:
        H.fault_inject("%s:%d" % (rng.choice(sites), rng.randint(1, 10)))
        time.sleep(rng.random() * 0.02)
        if rng.random() < 0.2:
            H.trim()
    H.fault_inject(None)
    stop.set()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive(), "a caller hangs"
    assert not problems, problems[:5]
    assert calls[0] > 30 and failures[0] > 3, (calls, failures)
    for n in roots:
        assert (H.merkle_root_host(leaves[n], 4, TAG4, 1) == roots[n]).all()
    assert (H.sponge_hash_host(msgs, 3000, 6, CAP, 1).reshape(-1) == dig_exp).all()


def test_fault_hook_argument_checking(hades_lib):
    lib = hades_lib
    assert lib.hades252_fault_inject(b"nosuchsite:1") == -1
    assert lib.hades252_fault_inject(b"malloc:0") == -1
    assert lib.hades252_fault_inject(b"") == 0 and lib.hades252_fault_inject(None) == 0


# ---------------------------------------------------------------------------------------------
# the pool of pipes is bounded and can be emptied
# ---------------------------------------------------------------------------------------------
def test_pool_is_bounded_and_trim_gives_memory_back(torch_cuda, H, hades_lib, oracle):
    torch = torch_cuda
    H.trim()
    assert H.pool_bytes() == 0
    free0 = torch.cuda.mem_get_info()[0]
    n = 1 << 22                                                   # 128 MiB of leaves: arena ~ 45 MiB, chunk buffers 6 x 32 MiB
    leaves = oracle.gen_b(0, n)
    exp = hex(int_of(H.merkle_root_host(leaves, 4, TAG4, 1)))
    held = H.pool_bytes()
    budget = int(os.environ.get("HADES252_POOL_MAX_BYTES", 1 << 30))
    assert held <= budget and (held > 0 or budget < (1 << 28))
    # many concurrent large calls: every one gets its own pipe; what returns to the pool stays under the budget
    out = [None] * 6

    def work(i):
        out[i] = hex(int_of(H.merkle_root_host(leaves, 4, TAG4, 1)))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out == [exp] * 6
    assert H.pool_bytes() <= (1 << 30)
    H.trim()
    assert H.pool_bytes() == 0
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 >= free0 - (64 << 20), "trim did not give the device memory back (%d -> %d)" % (free0, free1)
    # and the library works afterwards
    small = leaves[: 4 * 4 ** 8]
    assert (H.merkle_root_host(small, 4, TAG4, 1) == oracle.merkle_tree(small, 4, TAG4, 1)[-1]).all()


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
