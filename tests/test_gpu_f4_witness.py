"""GPU tier, SURVEY section 8 row f4: every gate output of the reference's `GadgetStrategy` (972 wires per state,
src/strategies/gadget.rs:41-133) against the oracle and against the gates themselves."""
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


def test_perm_witness_all_gadget_wires(torch_cuda, H, oracle):
    """hades252_perm_witness_dev: all 972 gate outputs of the reference's GadgetStrategy per state
    (src/strategies/gadget.rs:41-133) vs the spec oracle's restatement of that schedule, on edge and random
    states; then batch-wide identities on 5 000 states: r2 of the last round == perm output, and every S-box
    triple satisfies v4 == v2^2 through the device field ops."""
    torch = torch_cuda
    rng = random.Random(17)
    cases = [[5000] * 5, [0] * 5, [P - 1] * 5, [1, 2, 3, 4, 5]] + [[rng.randrange(P) for _ in range(5)] for _ in range(6)]
    n_pad = 70                                  # more than one wave, ragged
    vals = cases + [[rng.randrange(P) for _ in range(5)] for _ in range(n_pad - len(cases))]
    inp = np.array([l for st in vals for v in st for l in limbs_of(S.to_mont(v))], dtype=np.uint64)
    dev = to_dev(torch, inp)
    wires = H.perm_witness(dev)
    assert tuple(wires.shape) == (972, n_pad, 4)
    assert (to_host(dev) == inp).all()                       # input untouched
    host = wires.cpu().numpy().view(np.uint64).reshape(972, n_pad, 4)
    for i in list(range(len(cases))) + [63, 64, 69]:
        spec = []
        S.perm_gadget(vals[i], spec)
        got = [int_of(host[g, i]) for g in range(972)]
        bad = [g for g in range(972) if got[g] != S.to_mont(spec[g])]
        assert not bad, (i, bad[:8])
    # batch-wide identities
    n = 5000
    st = H.gen_b(5 * n, "cuda")
    w = H.perm_witness(st)
    out = st.clone()
    H.ScalarStrategy().perm(out)
    last = torch.stack([w[962 + 2 * j + 1] for j in range(5)], dim=1)       # r2[j] of round 66
    assert torch.equal(last.reshape(-1), out.reshape(-1))
    for g in (5, 8, 17, 20 + 10, 5 + 15 + 10 + 15 + 10):                    # some v2 wires (rounds 0, 0, 0, 1, 2)
        v2, v4 = w[g].contiguous(), w[g + 1].contiguous()
        assert torch.equal(H.fr_op(H.FR_SQUARE, v2), v4)


def test_witness_rows_equal_trace_plus_next_round_key(torch_cuda, H, oracle):
    """Two independent kernels at scale: for every round r and word j, the gadget's row wire r2[r][j] must equal the
    per-round trace state + the NEXT round's constant (src/strategies/gadget.rs:102-129), on 2^14 states."""
    torch = torch_cuda
    n = 1 << 14
    st = H.gen_b(5 * n, "cuda", first_elem=12345)
    wires = H.perm_witness(st)
    trace = H.perm_trace(st)                                   # [67, n, 5, 4]
    base = 5                                                   # wires of round 0's key additions
    for r in range(67):
        full = r < 4 or r >= 63
        base += 15 if full else 3                              # S-box wires of this round
        for j in range(5):
            row = wires[base + 2 * j + 1]
            state = trace[r, :, j, :].contiguous()
            if r < 66:
                c = scalars_dev(torch, [oracle.round_constant(5 * (r + 1) + j)]).expand(n, 4).contiguous()
                state = H.fr_op(H.FR_ADD, state, c)
            assert torch.equal(row, state), (r, j)
        base += 10
    assert base == 972


def gate_violations(torch, H, st, w):
    """The gates of the reference's GadgetStrategy (src/strategies/gadget.rs:59-69, :102-129) evaluated by the device field
    ops on a claimed witness `w` (972 wires x n scalars) for the input states `st`: returns the list of gate indices whose
    output wire does not satisfy its gate for at least one state -- empty for a valid witness.  Every S-box triple
    (v2 = v v, v4 = v2 v2, v5 = v4 v with v the wire that feeds it), every r1 = M[j][0] z0 + M[j][1] z1 + M[j][2] z2 and
    r2 = r1 + M[j][3] z3 + M[j][4] z4 + c; with the first five wires (input + round key) this pins every wire to the
    input by induction."""
    n = st.numel() // 20
    mul = lambda a, b: H.fr_op(H.FR_MUL, a.contiguous(), b.contiguous())
    add = lambda a, b: H.fr_op(H.FR_ADD, a.contiguous(), b.contiguous())
    const = lambda v: scalars_dev(torch, [S.to_mont(v)]).expand(n, 4).contiguous()   # v: canonical integer
    mds = [[const(v) for v in row] for row in S.mds_matrix()]
    ark = S.round_constants()
    bad = []

    def gate(idx, expect):
        if not torch.equal(w[idx], expect):
            bad.append(idx)

    state = []
    for j in range(5):                                                          # wires 0..4: input + first round key
        state.append(w[j])
        gate(j, add(st.view(n, 5, 4)[:, j, :], const(ark[j])))
    g = 5
    for r in range(67):
        full = r < 4 or r >= 63
        z = list(state)
        for word in (range(5) if full else (4,)):
            v = state[word]
            gate(g, mul(v, v))
            gate(g + 1, mul(w[g], w[g]))
            gate(g + 2, mul(w[g + 1], v))
            z[word] = w[g + 2]
            g += 3
        nxt = []
        for j in range(5):
            gate(g, add(add(mul(mds[j][0], z[0]), mul(mds[j][1], z[1])), mul(mds[j][2], z[2])))
            r2 = add(add(mul(mds[j][3], z[3]), mul(mds[j][4], z[4])), w[g])
            if r < 66:
                r2 = add(r2, const(ark[5 * (r + 1) + j]))
            gate(g + 1, r2)
            nxt.append(w[g + 1])
            g += 2
        state = nxt
    assert g == 972
    return bad


def test_witness_every_gate_identity_at_scale(torch_cuda, H, oracle):
    """All 972 wires of 4 096 states against the GATES themselves, evaluated by the device field ops on the kernel's own
    outputs (gate_violations above)."""
    torch = torch_cuda
    n = 1 << 12
    st = H.gen_b(5 * n, "cuda", first_elem=99)
    w = H.perm_witness(st)
    assert gate_violations(torch, H, st, w) == []


# ---------------------------------------------------------------------------------------------
# the reference's own gadget tests (src/strategies/gadget.rs:207-271), read on the device: `preimage` (random input: the
# ScalarStrategy output equals the gadget's output wires, and the witness satisfies every gate -- what prove + verify
# establish there), `preimage_constant` (input [5000; 5]), `preimage_fails` (a wrong witness must NOT satisfy the gates)
# ---------------------------------------------------------------------------------------------
def _gadget_output_rows(w):
    nw = w.shape[0]
    return [w[nw - 10 + 2 * j + 1] for j in range(5)]          # r2 of the last round = the permutation's output words


def test_preimage_like_reference(torch_cuda, H):
    torch = torch_cuda
    rng = random.Random(20240207)
    vals = [rng.randrange(P) for _ in range(5 * 64)]                     # gadget.rs:182-195: BlsScalar::random
    st = scalars_dev(torch, [S.to_mont(v) for v in vals]).view(-1)
    w = H.perm_witness(st)
    out = st.clone()
    H.ScalarStrategy().perm(out)                                         # gadget.rs:192: ScalarStrategy::new().perm(&mut output)
    for j, row in enumerate(_gadget_output_rows(w)):                     # gadget.rs:166-175: assert_equal on every output word
        assert torch.equal(row, out.view(-1, 5, 4)[:, j, :]), j
    assert gate_violations(torch, H, st, w) == []


def test_preimage_constant_like_reference(torch_cuda, H):
    torch = torch_cuda
    st = scalars_dev(torch, [S.to_mont(5000)] * 5).view(-1)              # gadget.rs:230
    w = H.perm_witness(st)
    out = st.clone()
    H.ScalarStrategy().perm(out)
    assert [hex_of(r) for r in _gadget_output_rows(w)] == [hex(S.to_mont(v)) for v in S.perm([5000] * 5)]
    for j, row in enumerate(_gadget_output_rows(w)):
        assert torch.equal(row, out.view(-1, 5, 4)[:, j, :]), j
    assert gate_violations(torch, H, st, w) == []


def test_preimage_fails_like_reference(torch_cuda, H):
    """gadget.rs:246-271: the claimed output is the permutation of a DIFFERENT input -- the circuit must not be satisfiable.
    Here: the witness of [5000; 5] with its output rows replaced by those of [5001; 5] violates exactly the five output
    gates; one flipped bit in one wire of one state violates that wire's gate and the gates that consume it."""
    torch = torch_cuda
    st = scalars_dev(torch, [S.to_mont(5000)] * 5).view(-1)
    other = scalars_dev(torch, [S.to_mont(5001)] * 5).view(-1)
    w, w_other = H.perm_witness(st), H.perm_witness(other)
    nw = w.shape[0]
    forged = w.clone()
    for j in range(5):
        forged[nw - 10 + 2 * j + 1] = w_other[nw - 10 + 2 * j + 1]
    assert gate_violations(torch, H, st, forged) == [nw - 10 + 2 * j + 1 for j in range(5)]
    n = 256
    stn = H.gen_b(5 * n, "cuda", first_elem=7)
    wn = H.perm_witness(stn)
    assert gate_violations(torch, H, stn, wn) == []
    wn[300, 17, 0] ^= 1                                                  # wire 300 of state 17: one bit
    bad = gate_violations(torch, H, stn, wn)
    assert 300 in bad and len(bad) <= 12, bad                           # its own gate + the few gates that consume it


# ---------------------------------------------------------------------------------------------------------------
# The per-round trace in SCALED form (round 6; hades252_perm_trace_scaled_dev + hades252_perm_trace_scale_table): what the
# throughput kernel holds after every round, fully reduced; true[r][w] = scaled[r][w] * mul[r] + add[r][w].
# ---------------------------------------------------------------------------------------------------------------
def _unscale(torch, H, scaled, mul, add):
    """The consumer's side, on the device with the library's own BlsScalar operations (hades252_fr_op_dev)."""
    rounds, n = scaled.shape[0], scaled.shape[1]
    out = torch.empty_like(scaled)
    for r in range(rounds):
        m = to_dev(torch, np.tile(mul[r], n * 5)).view(-1, 4)
        a = to_dev(torch, np.tile(add[r].reshape(-1), n)).view(-1, 4)
        prod = H.fr_op(H.FR_MUL, scaled[r].reshape(-1, 4).contiguous(), m)
        out[r] = H.fr_op(H.FR_ADD, prod, a).view(n, 5, 4)
    return out


def test_scaled_trace_times_table_is_the_oracle_trace(torch_cuda, H, oracle):
    torch = torch_cuda
    mul, add = H.trace_scale_table()
    rng = random.Random(606)
    edge = [0, 1, P - 1, R, P - R, (1 << 255) % P, (1 << 254) - 1]
    cases = [[1] * 5, [0] * 5, [P - 1] * 5, [17] * 5, [5000] * 5] + [[rng.choice(edge) for _ in range(5)] for _ in range(11)]
    cases += [[rng.randrange(P) for _ in range(5)] for _ in range(48)]
    st = scalars_dev(torch, [S.to_mont(v) for c in cases for v in c]).view(-1, 5, 4)
    keep = st.clone()
    scaled = H.perm_trace_scaled(st)
    assert torch.equal(st, keep)                                        # the input is left untouched
    assert tuple(scaled.shape) == (67, len(cases), 5, 4)
    got = _unscale(torch, H, scaled, mul, add)
    for i, c in enumerate(cases):                                       # every round, every word, against the ORACLE's trace
        _, tr = oracle.perm_trace(np.array([l for v in c for l in limbs_of(S.to_mont(v))], dtype=np.uint64))
        assert (to_host(got[:, i]).reshape(67, 5, 4) == tr).all(), i
    # every stored value is a fully reduced field element
    host = to_host(scaled).reshape(-1, 4)
    assert all(int_of(row) < P for row in host[:: max(1, len(host) // 500)])
    # scale 1 and no deferred constants after the last round?  No: the last round still carries the schedule's final scale
    assert not torch.equal(scaled[66], got[66]) and (add[66] == 0).all() and (add[0] == 0).all() and add[30].any()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 4097])
def test_scaled_trace_ragged_sizes_equal_true_trace(torch_cuda, H, n):
    """Any n (ragged last wave / block), guard rows around the output, against the shipped true-form trace kernel."""
    torch = torch_cuda
    mul, add = H.trace_scale_table()
    st = H.gen_b(5 * n, "cuda", first_elem=5 * 1234).view(n, 5, 4)
    guard = 0x5A5A5A5A5A5A5A5A
    buf = torch.full((67 * n * 20 + 40,), guard, dtype=torch.int64, device="cuda")
    out = buf[20:20 + 67 * n * 20].view(67, n, 5, 4)
    H.perm_trace_scaled(st, out=out)
    assert (buf[:20] == guard).all() and (buf[-20:] == guard).all()
    assert torch.equal(_unscale(torch, H, out, mul, add), H.perm_trace(st))


def test_scaled_trace_at_scale_last_round_is_perm(torch_cuda, H):
    """2^18 states: the un-scaled last round equals the permutation itself; a strided sample of rounds equals the true trace."""
    torch = torch_cuda
    n = 1 << 18
    mul, add = H.trace_scale_table()
    st = H.gen_b(5 * n, "cuda").view(n, 5, 4)
    scaled = H.perm_trace_scaled(st)
    true = H.perm_trace(st)
    for r in (0, 3, 4, 30, 62, 63, 66):
        got = _unscale(torch, H, scaled[r:r + 1], mul[r:r + 1], add[r:r + 1])
        assert torch.equal(got[0], true[r]), r
    out = st.clone().view(-1)
    H.ScalarStrategy().perm(out)
    assert torch.equal(_unscale(torch, H, scaled[66:67], mul[66:67], add[66:67])[0].reshape(-1), out)


def test_finalize_window_both_rare_sides_on_the_device(torch_cuda, H):
    """The exit routine of the scaled trace (finalize_window: pack, add p, a wave-uniform fix-up branch) driven directly
    through hades252_fr_op_dev(HADES252_FR_REDUCE_SIGNED): the two sides real states reach a few times per thousand words
    (x < -p) or practically never (x >= 0), the window's edges, waves where no / one / every lane takes the fix-up."""
    torch = torch_cuda
    rng = random.Random(77)
    edge = [0, 1, -1, -P, -P + 1, -P - 1, 1 << 250, -P - (1 << 250) + 1, -(P >> 1), (1 << 249) + 12345, -P - (1 << 249) - 7,
            (1 << 250) - 1, -P - 2, 2]
    common = [-rng.randrange(1, P) for _ in range(64 * 6)]                     # [-p, 0): the fix-up branch is skipped
    one_lane = list(common[:64]); one_lane[37] = 5                              # one lane of a wave needs it
    all_lanes = [rng.randrange(0, 1 << 250) for _ in range(32)] + [-P - rng.randrange(1, 1 << 250) for _ in range(32)]
    xs = common + one_lane + all_lanes + edge + [rng.choice(edge) for _ in range(50)]
    a = scalars_dev(torch, [x & ((1 << 256) - 1) for x in xs])
    got = to_host(H.fr_op(H.FR_REDUCE_SIGNED, a)).reshape(-1, 4)
    for x, row in zip(xs, got):
        assert int_of(row) == x % P, hex(x)
    with pytest.raises(Exception):
        H.fr_op(H.FR_REDUCE_SIGNED, a, impl=H.FR_IMPL_SATURATED32)              # a radix-2^29 routine only
