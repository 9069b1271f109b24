"""GPU tier, SURVEY section 8 row f3: the canonical-bytes wire format (`to_bytes` / `from_bytes`) on the device, pinned byte for
byte to the reference's `ark.bin` / `mds.bin` (pin #0b: the reference's own `test_round_constants` made absolute)."""
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
