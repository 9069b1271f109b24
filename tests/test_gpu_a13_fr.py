"""GPU tier, SURVEY section 8 rows a9 / a10 / a13: the device field arithmetics (`BlsScalar` add / mul / square / from_raw on the
saturated 8 x 32 and the radix-2^29 path) regenerate every byte of the reference's constant blobs (pin #0), and agree with the
oracle on edge values."""
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
