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
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).view(np.int64)).cuda()


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
