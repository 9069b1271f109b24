"""GPU tier, SURVEY section 8 row b: the drop-in boundary on HOST memory -- `hades252_perm_batch*` (page-locked, pageable,
staged, chunk pipeline, `_multi` workers), the one-shot Merkle / sponge host callers, device-memory helpers, graph capture,
the failure contract under injected faults, the bounded pool, warm-up."""
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
