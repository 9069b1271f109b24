"""Helpers shared by the GPU-tier test files (tests/test_gpu_*.py, one file per SURVEY section 8 row).  The fixtures
`torch_cuda` and `H` live in conftest.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hades_spec as S  # noqa: E402
from oracle_lib import limbs_of, int_of  # noqa: E402
from test_blob_kat import ARK_SHA256, MDS_SHA256, blob_bytes  # noqa: E402,F401

__all__ = ["KERNELS", "TAG", "TAG4", "CAP", "SITES_PERM", "to_dev", "to_host", "hex_of", "scalars_dev", "rows",
           "kernel_available", "each_state_is_input_or_output", "_record", "ARK_SHA256", "MDS_SHA256", "blob_bytes"]

KERNELS = [1, 2, 3, 4, 5]   # HADES252_KERNEL_LITERAL, _FAST (one state per lane), _COOP (five waves per state), _LANES (one
                            # state per wave, elements spread over 16-lane rows), _ROWS (one state per row, four per wave)
TAG = {1: S.to_mont(1), 2: S.to_mont(3), 3: S.to_mont(7), 4: S.to_mont(15)}      # per-arity domain tags of the Merkle tests
TAG4 = S.to_mont(15)
CAP = S.to_mont(1 << 64)
SITES_PERM = ["malloc", "hostmalloc", "hostregister", "memcpy", "streamcreate", "eventcreate", "sync"]


def to_dev(torch, arr):
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    return torch.from_numpy((a if a.flags.writeable else a.copy()).view(np.int64)).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint64).reshape(-1)


def hex_of(t):
    return hex(int_of(to_host(t)))


def scalars_dev(torch, ints):
    return to_dev(torch, np.array([l for v in ints for l in limbs_of(v)], dtype=np.uint64)).view(-1, 4)


def rows(a):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)


def kernel_available(hades_lib, torch, k):
    t = torch.zeros(20, dtype=torch.int64, device="cuda")
    return hades_lib.hades252_perm_batch_dev_ex(t.data_ptr(), 1, None, k) == 0


def each_state_is_input_or_output(got, inp, exp):
    g, i, e = got.reshape(-1, 20), inp.reshape(-1, 20), exp.reshape(-1, 20)
    is_in, is_out = (g == i).all(axis=1), (g == e).all(axis=1)
    return bool((is_in | is_out).all()), int(is_out.sum())


def _record(name, text):
    """Append a line to gpurun_out/<name> (travels back from the GPU box: evidence of the full-size runs)."""
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "a") as f:
        f.write(text + "\n")
