"""GPU tier, SURVEY section 8 rows a2 - a7: the trait's per-operation methods (`add_round_key`, `quintic_s_box`, `mul_matrix`,
`apply_full_round`, `apply_partial_round`), the constant cursor `next_c` over all 960 constants, `perm` as the composition of
the per-round kernels, and the per-round trace."""
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


def test_per_op_kernels(torch_cuda, H, oracle):
    """Strategy::add_round_key / quintic_s_box / mul_matrix / apply_*_round vs the oracle."""
    torch = torch_cuda
    n = 777
    inp = oracle.gen_b(31337, 5 * n)
    strat = H.ScalarStrategy()
    for rnd in (0, 3, 4, 35, 62, 63, 66):
        buf = to_dev(torch, inp)
        it = H.RoundConstantsIter(5 * rnd)
        strat.add_round_key(it, buf)
        assert it.pos == 5 * rnd + 5
        assert (to_host(buf) == oracle.add_round_key(inp, rnd)).all()
        buf = to_dev(torch, inp)
        strat.apply_full_round(H.RoundConstantsIter(5 * rnd), buf)
        assert (to_host(buf) == oracle.full_round(inp, rnd)).all()
        buf = to_dev(torch, inp)
        strat.apply_partial_round(H.RoundConstantsIter(5 * rnd), buf)
        assert (to_host(buf) == oracle.partial_round(inp, rnd)).all()
    buf = to_dev(torch, inp)
    strat.quintic_s_box(buf)
    assert (to_host(buf) == oracle.quintic_s_box(inp)).all()
    buf = to_dev(torch, inp)
    strat.mul_matrix(H.RoundConstantsIter(), buf)
    assert (to_host(buf) == oracle.mul_matrix(inp)).all()
    with pytest.raises(RuntimeError, match="out of ARK constants"):
        strat.add_round_key(H.RoundConstantsIter(956), buf)


def test_per_op_kernels_edge_values(torch_cuda, H, oracle):
    """Field-operation edge cases through the per-op kernels: operands 0, 1, p-1, p-2, R, values
    that make word + constant cross p, etc. (saturated 8x32 arithmetic of csrc/fr32.hpp)."""
    torch = torch_cuda
    rng = random.Random(77)
    edge = [0, 1, 2, P - 1, P - 2, R, P - R, (1 << 255) % P, P - (1 << 32), 0xFFFFFFFF, 0xFFFFFFFF00000000,
            (P - 1) // 2, (P + 1) // 2, (1 << 224) - 1]
    ark = S.round_constants()
    # complements of the first round constants (Montgomery domain): w + c == 0, p - 1, 1 (mod p)
    for c in ark[:10]:
        cm = S.to_mont(c)
        edge += [(P - cm) % P, (P - cm - 1) % P, (P - cm + 1) % P]
    n = 1024
    vals = [rng.choice(edge) if rng.random() < 0.8 else rng.randrange(P) for _ in range(5 * n)]
    inp = np.array([l for v in vals for l in limbs_of(v)], dtype=np.uint64)
    strat = H.ScalarStrategy()
    for rnd in (0, 1, 33, 66):
        buf = to_dev(torch, inp)
        strat.add_round_key(H.RoundConstantsIter(5 * rnd), buf)
        assert (to_host(buf) == oracle.add_round_key(inp, rnd)).all()
    buf = to_dev(torch, inp)
    strat.quintic_s_box(buf)
    assert (to_host(buf) == oracle.quintic_s_box(inp)).all()
    buf = to_dev(torch, inp)
    strat.mul_matrix(H.RoundConstantsIter(), buf)
    assert (to_host(buf) == oracle.mul_matrix(inp)).all()
    # wire format on the same edge values
    canon = np.frombuffer(b"".join(S.from_mont(v).to_bytes(32, "little") for v in vals), dtype=np.uint64).copy()
    assert (to_host(H.to_bytes(to_dev(torch, inp))) == canon).all()
    assert (to_host(H.from_bytes(to_dev(torch, canon))) == inp).all()


def test_perm_is_the_composition_of_rounds(torch_cuda, H, oracle):
    """The trait's provided perm (strategies.rs:140-157) replayed over the per-round entry
    points equals the fused kernel."""
    torch = torch_cuda
    inp = oracle.gen_b(5, 5 * 300)
    a, b = to_dev(torch, inp), to_dev(torch, inp)
    H.ScalarStrategy().perm(a)
    H.ScalarStrategy().perm_stepwise(b)
    assert torch.equal(a, b)


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


def test_perm_trace(torch_cuda, H, oracle):
    """Per-round states (Gadget witness pre-computation) vs the oracle's trace."""
    torch = torch_cuda
    n = 300
    inp = oracle.gen_b(777, 5 * n)
    dev = to_dev(torch, inp)
    tr = H.perm_trace(dev)
    assert (to_host(dev) == inp).all()                      # input untouched
    host = tr.cpu().numpy().view(np.uint64).reshape(67, n, 20)
    for i in (0, 1, 63, 64, 299):
        _, otr = oracle.perm_trace(inp[20 * i:20 * i + 20])
        assert (host[:, i, :] == otr.reshape(67, 20)).all()
    out = dev.clone()
    H.ScalarStrategy().perm(out)
    assert torch.equal(tr[66].reshape(-1), out.reshape(-1))


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
