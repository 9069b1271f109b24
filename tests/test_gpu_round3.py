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
