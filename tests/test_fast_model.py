"""CPU tier: a limb-exact Python replay of the scale-tracked kernel (hades_fast.hpp) with the
machine-word bounds asserted on every intermediate, checked against the spec oracle.

Random GPU tests cannot show that a 64-bit column never overflows; this model asserts it on real
inputs AND on adversarial maximal-limb inputs (which are not reachable, but bound the reachable)."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hades_spec as S  # noqa: E402
from hades252_amd import _derive as D  # noqa: E402

P = D.P
LB, NL = D.LIMB_BITS, D.NLIMB
MASK = (1 << LB) - 1
P29 = D.to_limbs29(P)
I63 = 1 << 63
I31 = 1 << 31
LAZY = 3 << 28          # |limb| < 1.5 * 2^29 after the balanced-limb ARK


def val(limbs):
    return sum(l << (LB * k) for k, l in enumerate(limbs))


def check_acc(acc):
    assert -I63 <= acc < I63, "signed 64-bit accumulator overflow"


def mont_fips(a, b, sqr=False):
    """mont_fips of hades_fast.hpp (signed digits): normalised limbs of a value
    == a*b/2^261 (mod p) in (a*b/Rp - p, a*b/Rp]."""
    assert all(-LAZY < x < LAZY for x in a + b), "|input limb| must be < 1.5 * 2^29"
    assert abs(val(a)) < (1 << 257) and abs(val(b)) < (1 << 257)
    m = [0] * NL
    r = [0] * NL
    acc = 0
    d = [2 * x for x in a]
    assert all(-I31 <= x < I31 for x in d)
    for k in range(2 * NL - 1):
        lo, hi = (0, k) if k < NL else (k - NL + 1, NL - 1)
        for i in range(lo, hi + 1):
            j = k - i
            if sqr:
                if i < j:
                    acc += a[i] * d[j]
                elif i == j:
                    acc += a[i] * a[i]
            else:
                acc += a[i] * b[j]
            check_acc(acc)
        for i in range(lo, hi + 1):
            if k - i >= 1:
                acc -= m[i] * P29[k - i]
                check_acc(acc)
        low = acc & MASK                     # two's complement low bits, in [0, 2^29)
        if k < NL:
            m[k] = low
            assert (acc - low) % (1 << LB) == 0
        else:
            r[k - NL] = low
        acc >>= LB                           # arithmetic shift (python ints floor)
    assert -I31 <= acc < I31
    r[NL - 1] = acc
    v = val(r)
    ab = val(a) * val(b)
    assert ab - P * (1 << (LB * NL)) < v * (1 << (LB * NL)) <= ab, "result outside (ab/Rp - p, ab/Rp]"
    return r


def mont_lin(a, factor, steps=2):
    """mont_lin (steps = 2) / mont_lin1 (steps = 1) of hades_fast.hpp: a * factor / Rp for a CONSTANT factor as a linear
    map over the limbs of a (table D.lin_table) + `steps` signed-digit steps.  Same congruence class as mont_fips(a, factor)."""
    assert all(0 <= x < (1 << LB) for x in a[:-1]) and abs(a[-1]) < (1 << 25), "input must be normalised"
    e = D.lin_table(factor, steps)
    assert len(e) == 81 and all(0 <= x < (1 << LB) for x in e)
    m = [0, 0]
    r = [0] * NL
    acc = 0
    for j in range(NL):
        for k in range(NL):
            acc += a[k] * e[NL * j + k]
            check_acc(acc)
        if j >= 1:
            acc -= m[0] * P29[j]
        if j >= 2 and steps == 2:
            acc -= m[1] * P29[j - 1]
        assert abs(acc) < (1 << 62)
        low = acc & MASK
        if j < steps:
            m[j] = low
            assert (acc - low) % (1 << LB) == 0
        else:
            r[j - steps] = low
        acc >>= LB
    if steps == 2:
        acc -= m[1] * P29[NL - 1]
        r[NL - 2] = acc & MASK
        acc >>= LB
    assert -(1 << 27) < acc < (1 << 27)
    r[NL - 1] = acc
    v = val(r)
    w = sum(a[k] * (factor * pow(2, LB * (k + steps - NL), P) % P) for k in range(NL))
    assert w - ((1 << (LB * steps)) - 1) * P <= v << (LB * steps) <= w, "result outside (W/2^(29 t) - p, W/2^(29 t)]"
    assert (v - val(a) * factor * pow(1 << (LB * NL), -1, P)) % P == 0, "not congruent to a * factor / Rp"
    if steps == 2:
        assert -P - (1 << 227) < v < (1 << 230)
    else:
        assert -P - (1 << 251) < v < 9 * P and abs(r[NL - 1]) < (1 << 27)
    return r


def small_mds(st):
    acc = [0] * 5
    m = [0] * 5
    out = [[0] * NL for _ in range(5)]
    for i in range(5):
        for j in range(5):
            acc[i] += st[j][0] * D.MDS_SMALL[i][j]
        check_acc(acc[i])
        m[i] = acc[i] & MASK
        acc[i] >>= LB
    for k in range(1, NL):
        for i in range(5):
            for j in range(5):
                acc[i] += st[j][k] * D.MDS_SMALL[i][j]
            acc[i] -= m[i] * P29[k]
            assert abs(acc[i]) < (1 << 60)
            out[i][k - 1] = acc[i] & MASK
            acc[i] >>= LB
    for i in range(5):
        assert -I31 <= acc[i] < I31
        out[i][NL - 1] = acc[i]
    return out


def sbox(x):
    x2 = mont_fips(x, x, True)
    x4 = mont_fips(x2, x2, True)
    return mont_fips(x4, x)


def normalised(x):
    return all(0 <= l < (1 << LB) for l in x[:-1]) and abs(val(x)) < (1 << 256)


def finalize_model(x, factor):
    """finalize(mont_mul_const(x, factor)) of hades_fast.hpp -> fully reduced integer."""
    v = val(mont_fips(x, D.to_limbs29(factor)))
    assert -2 * P < v < P
    v += 2 * P                       # + 2p, then two conditional subtractions
    assert 0 < v < 3 * P and v < (1 << 256)
    for _ in range(2):
        if v >= P:
            v -= P
    assert 0 <= v < P
    return v


def fast_perm_model(mont_vals, held=None):
    """mont_vals: 5 integers = in-memory BlsScalar values (value * 2^256 mod p).  Returns the same.
    held (a list): receives, per round, the integer value of each word the kernel holds after the round's linear layer."""
    sch = D.fast_schedule()
    st = [D.to_limbs29(v) for v in mont_vals]
    for r in range(D.ROUNDS):
        full = D.is_full_round(r)
        a, k = (sch["full"][r], None) if full else sch["part"][r]
        for w in (range(5) if full else (4,)):                               # partial: word 4 only
            st[w] = [x + y for x, y in zip(st[w], D.to_balanced29(a[w]))]     # lazy ARK
            assert all(-LAZY < l < LAZY for l in st[w])
        if not full:
            assert a[:4] == [0, 0, 0, 0]
        if full:
            st = [sbox(x) for x in st]
        else:
            st[4] = mont_lin(sbox(st[4]), k, steps=1)    # mont_lin1: one digit step, the linear layer divides again
        st = small_mds(st)
        for x in st:
            assert normalised(x)
        if held is not None:
            held.append([val(x) for x in st])
    out = []
    for x in st:
        v = val(mont_lin(x, sch["final_f"]))
        assert -2 * P < v < P
        v += 2 * P                       # finalize(): + 2p, then two conditional subtractions
        assert 0 < v < 3 * P and v < (1 << 256)
        for _ in range(2):
            if v >= P:
                v -= P
        assert 0 <= v < P
        out.append(v)
    return out


EDGE = [0, 1, P - 1, S.R, P - S.R, (1 << 255) % P, (1 << 254) - 1]


def finalize1_model(v):
    """finalize1 of hades_fast.hpp: v in (-p, p) -> + p, ONE conditional subtraction."""
    assert -P < v < P, "finalize1 needs its input in (-p, p)"
    v += P
    return v - P if v >= P else v


def test_wire_format_models():
    """k_wire (kernels_perm.hpp): from_bytes = finalize1(mont_lin(a, Rp * R)) -- a linear map over non-negative limbs, so
    W >= 0 and the two digit steps leave the result in (-p, 2^-25 p) -- and to_bytes = finalize1(mont_mul_small(x, 32)),
    whose result lies in (-p, 0] for a reduced x.  One conditional subtraction each; checked on edge and random values,
    and on the largest encodable non-canonical input (rejected by the kernel, but its arithmetic must stay in range)."""
    rng = random.Random(41)
    vals = EDGE + [P - 2, (1 << 255) - 19 if (1 << 255) - 19 < P else 5] + [rng.randrange(P) for _ in range(300)]
    f_from = D.RP * D.R % P
    for a in vals:
        w = val(mont_lin(D.to_limbs29(a), f_from))
        assert -P < w < (P >> 24)
        assert finalize1_model(w) == a * D.R % P                       # from_bytes: canonical -> Montgomery limbs
        x = a * D.R % P
        t = val(mont_fips(D.to_limbs29(x), D.to_limbs29(1 << (LB * NL - 256))))
        assert -P < t <= 0
        assert finalize1_model(t) == a                                 # to_bytes: Montgomery limbs -> canonical
    big = (1 << 256) - 1                                               # not canonical: zeroed by the kernel afterwards
    assert -P < val(mont_lin(D.to_limbs29(big), f_from)) < P


def test_mont_lin_bounds_adversarial():
    """mont_lin on the largest normalised operands (every limb 2^29 - 1, top limb +-(2^25 - 1)): the model asserts the
    64-bit column bound, the digit-step exactness and the result range; reachable inputs are smaller."""
    for top in ((1 << 25) - 1, -((1 << 25) - 1), 0):
        a = [MASK] * (NL - 1) + [top]
        for factor in (P - 1, D.RP * D.R % P, D.fast_schedule()["final_f"], D.fast_schedule()["part"][30][1], 1):
            for steps in (2, 1):
                r = mont_lin(a, factor, steps)
                assert all(0 <= x < (1 << LB) for x in r[:-1])
                if steps == 1:                                  # ... and the linear layer takes it as word 4 beside maximal words
                    big = [MASK] * (NL - 1) + [(1 << 24) - 1]
                    for row in small_mds([big, big, big, big, r]):
                        assert normalised(row)


def test_model_matches_spec_oracle():
    rng = random.Random(29)
    cases = [[1] * 5, [0] * 5, [P - 1] * 5, [17] * 5]
    cases += [[rng.choice(EDGE) for _ in range(5)] for _ in range(4)]
    cases += [[rng.randrange(P) for _ in range(5)] for _ in range(6)]
    for vals in cases:
        got = fast_perm_model([S.to_mont(v) for v in vals])
        assert got == [S.to_mont(v) for v in S.perm(vals)]


def finalize_window_model(limbs):
    """finalize_window of kernels_perm.hpp on normalised limbs: two's-complement packing, + p, the two rare sides."""
    assert normalised(limbs) and abs(limbs[-1]) < (1 << 26)
    x = val(limbs)
    assert -P - (1 << 250) < x <= (1 << 250), "outside the window"
    m256 = (1 << 256) - 1
    t = 0
    for w in range(8):                                  # from_f29: word w = bits [32 w, 32 w + 32) of sum limb_k 2^(29 k)
        k, sh = (32 * w) // LB, 32 * w - LB * ((32 * w) // LB)
        acc = (limbs[k] & 0xFFFFFFFF) >> sh
        have = LB - sh
        if k + 1 < NL:
            acc |= (limbs[k + 1] & 0xFFFFFFFF) << have
        have += LB
        if have < 32 and k + 2 < NL:
            acc |= (limbs[k + 2] & 0xFFFFFFFF) << have
        t |= (acc & 0xFFFFFFFF) << (32 * w)
    assert t == x & m256, "packing is not x mod 2^256"
    u = (t + P) & m256
    pos, low = (t >> 255) == 0, (u >> 255) == 1
    assert not (pos and low) and pos == (x >= 0) and low == (x + P < 0)
    r = t if pos else ((u + P) & m256 if low else u)
    assert r == x % P and r < P
    return r, pos or low


def test_finalize_window_edges_and_rarity():
    edge_vals = [0, 1, -1, -P, -P + 1, -P - 1, (1 << 250), -P - (1 << 250) + 1, -(P >> 1), (1 << 249) + 12345,
                 -P - (1 << 249) - 7]
    rng = random.Random(8)
    edge_vals += [-rng.randrange(P) for _ in range(200)]
    for x in edge_vals:
        limbs = [(x >> (LB * k)) & MASK for k in range(NL - 1)] + [x >> (LB * (NL - 1))]       # normalised, signed top limb
        assert val(limbs) == x
        r, _ = finalize_window_model(limbs)
        assert r == x % P
    # on real states the fix-up side is rare (what lets a wave skip it): a few per thousand words
    fixups = words = 0
    for _ in range(6):
        held = []
        fast_perm_model([rng.randrange(P) for _ in range(5)], held)
        for row in held:
            for v in row:
                limbs = [(v >> (LB * k)) & MASK for k in range(NL - 1)] + [v >> (LB * (NL - 1))]
                fixups += finalize_window_model(limbs)[1]
                words += 1
    assert words == 6 * 67 * 5 and fixups / words < 0.02, (fixups, words)


def test_scaled_trace_model_matches_spec_oracle():
    """k_perm_trace_scaled (kernels_perm.hpp): the state the throughput kernel HOLDS after each round leaves through
    `finalize` alone -- its window (-p - 2^250, 2^250] must hold for every word of every round -- and the host's table
    (D.trace_scaled_tables: one multiplier per round, one addend per round and word) turns it into the true state:
    true = scaled (*) mul[r] (+) add[r][w] as BlsScalar operations on in-memory values, against the oracle's trace."""
    rng = random.Random(61)
    mul, add = D.trace_scaled_tables()
    assert len(mul) == 67 and len(add) == 67 and all(len(a) == 5 for a in add)
    # addends are zero wherever no constant is deferred: the full rounds and the last partial round's hand-over
    assert all(add[r] == [0] * 5 for r in range(67) if D.is_full_round(r)) and any(add[30])
    r_inv = pow(S.R, -1, P)
    cases = [[1] * 5, [0] * 5, [P - 1] * 5] + [[rng.choice(EDGE) for _ in range(5)] for _ in range(3)]
    cases += [[rng.randrange(P) for _ in range(5)] for _ in range(6)]
    for vals in cases:
        held, tr = [], []
        out = fast_perm_model([S.to_mont(v) for v in vals], held)
        assert S.perm(vals, tr) == [S.from_mont(v) for v in out] and len(held) == 67
        for r in range(67):
            for w in range(5):
                v = held[r][w]
                assert -P - (1 << 250) < v <= (1 << 250), "outside finalize's window"
                scaled = v % P                                           # what finalize stores: fully reduced
                assert (v + 2 * P) - scaled in (0, P, 2 * P)             # ... by two conditional subtractions after + 2p
                true_mem = (scaled * mul[r] % P * r_inv + add[r][w]) % P  # BlsScalar mul (a b / R), then add
                assert true_mem == S.to_mont(tr[r][w]), (r, w)


def small_mds_row(i, st):
    """small_mds_row of hades_coop.hpp: one output row, same arithmetic as row i of small_mds."""
    acc = sum(st[j][0] * D.MDS_SMALL[i][j] for j in range(5))
    check_acc(acc)
    m = acc & MASK
    acc >>= LB
    out = [0] * NL
    for k in range(1, NL):
        acc += sum(st[j][k] * D.MDS_SMALL[i][j] for j in range(5)) - m * P29[k]
        assert abs(acc) < (1 << 60)
        out[k - 1] = acc & MASK
        acc >>= LB
    assert -I31 <= acc < I31
    out[NL - 1] = acc
    return out


def coop_perm_model(mont_vals):
    """Limb-exact replay of k_perm_coop (hades_coop.hpp): every word on its own wave; partial rounds scale
    words 0..3 up (G_r) instead of scaling word 4 down."""
    co = D.coop_schedule()
    st = [D.to_limbs29(v) for v in mont_vals]
    for r in range(D.ROUNDS):
        full = D.is_full_round(r)
        nxt = []
        for w in range(5):
            x = st[w]
            if full or w == 4:
                x = [a + b for a, b in zip(x, D.to_balanced29(co["a"][r][w]))]
                assert all(-LAZY < l < LAZY for l in x)
                x = sbox(x)
            else:
                assert co["a"][r][w] == 0
                x = mont_fips(x, D.to_limbs29(co["g"][r]))
            assert normalised(x)
            nxt.append(x)
        rows = [small_mds_row(i, nxt) for i in range(5)]
        assert rows == small_mds(nxt)                 # the row form IS the shipped linear layer
        st = rows
    return [finalize_model(x, co["final_f"]) for x in st]


def test_coop_model_matches_spec_oracle():
    rng = random.Random(41)
    cases = [[1] * 5, [0] * 5, [P - 1] * 5, [15, 1, 2, 3, 4]]
    cases += [[rng.choice(EDGE) for _ in range(5)] for _ in range(3)]
    cases += [[rng.randrange(P) for _ in range(5)] for _ in range(4)]
    for vals in cases:
        got = coop_perm_model([S.to_mont(v) for v in vals])
        assert got == [S.to_mont(v) for v in S.perm(vals)]


def finalize32_model(x):
    """finalize32 of kernels_perm.hpp: x / 32 mod p, fully reduced, for an Rp-form value x in (-30 p, p / 8) with lazy
    limbs: m = (-x mod 32) + 32 makes x + m p a multiple of 32 in (2 p, 64 p); one conditional subtraction.  (Any x in
    (-32 p, p) would do; the schedule stays inside (-2 p - 2^245, 2^253): rows reach -p - 2^245 before a round constant in
    (-p, 0] is appended -- the bound the row path is driven to in test_finalize32_window_adversarial.)"""
    v = val(x)
    assert -30 * P < v < (P >> 3), "finalize32: input outside its window"
    assert all(abs(l) < I31 - (1 << LB) for l in x)
    m = ((-x[0]) & 31) + 32
    out, carry = [], 0
    for k in range(NL):
        t = m * P29[k] + x[k] + carry                # one multiply-add on a 32-bit carry chain
        assert abs(x[k] + carry) < I31 and abs(t) < (1 << 40)
        out.append(t & MASK if k < NL - 1 else t)
        carry = t >> LB
    assert 0 <= out[NL - 1] < (1 << LB), "the sum is below 64 p < 2^261"
    t = val(out)
    assert t == v + m * P and t % 32 == 0
    t >>= 5                                          # the packing reads bits [5, 261)
    assert 0 < t < 2 * P and t < (1 << 256)
    return t - P if t >= P else t


def mds_row_cols(u, j, ncol):
    """mds_row_cols<NCOL> of kernels_perm.hpp: (sum_{k < ncol} C[j][k] U_k - m p) / 2^29, normalised; lazy limbs allowed."""
    acc = sum(u[c][0] * D.MDS_SMALL[j][c] for c in range(ncol))
    check_acc(acc)
    m = acc & MASK
    acc >>= LB
    out = [0] * NL
    for kk in range(1, NL):
        acc += sum(u[c][kk] * D.MDS_SMALL[j][c] for c in range(ncol)) - m * P29[kk]
        assert abs(acc) < (1 << 60)
        out[kk - 1] = acc & MASK
        acc >>= LB
    assert -I31 <= acc < I31
    out[NL - 1] = acc
    assert normalised(out)
    return out


def add_lazy(x, c):
    r = [a + b for a, b in zip(x, c)]
    assert all(-LAZY < l < LAZY for l in r)
    return r


def witness_model(mont_vals, trace=None):
    """Limb-exact replay of k_perm_witness and k_perm_trace_fast (kernels_perm.hpp): the TRUE-FORM schedule of
    hades252_amd/_derive.py::witness_schedule -- every held value is x * Rp, Montgomery products are closed in that form,
    the linear layer is one constant linear map per word (U = Y lam 2^29) + the small-integer rows + the one-limb step,
    and every gate output of the reference's GadgetStrategy leaves through finalize32 (the exact division by 32)."""
    ws = D.witness_schedule()
    c = [[D.to_balanced29_signed(v) for v in row] for row in ws["c"]]
    ck = [[D.to_balanced29(v) for v in row] for row in ws["ck"]]
    y = [mont_lin(D.to_limbs29(v), ws["f_in"]) for v in mont_vals]          # x Rp, normalised, no constants yet
    wires = []
    for r in range(D.ROUNDS):
        full = D.is_full_round(r)
        if r == 0:
            wires += [finalize32_model(add_lazy(y[w], c[0][w])) for w in range(5)]
        u = [None] * 5
        for w in range(5):
            if full or w == 4:
                z = add_lazy(y[w], c[r][w])
                v2 = mont_fips(z, z, True)
                v4 = mont_fips(v2, v2, True)
                v5 = mont_fips(v4, z)
                wires += [finalize32_model(v2), finalize32_model(v4), finalize32_model(v5)]
                u[w] = mont_lin(v5, ws["f_k"])
            else:
                u[w] = add_lazy(mont_lin(y[w], ws["f_k"]), ck[r][w])        # the round constant, seen through the map
        for j in range(5):
            wires.append(finalize32_model(mds_row_cols(u, j, 3)))
            wires.append(None)                                              # r2[j]: after the rows
        y = small_mds(u)
        assert y == [mds_row_cols(u, j, 5) for j in range(5)]
        if trace is not None:                                                   # k_perm_trace_fast: the same rounds, the
            trace.append([finalize32_model(x) for x in y])                      # state after the round as the only output
        for j in range(5):
            wires[len(wires) - 10 + 2 * j + 1] = finalize32_model(add_lazy(y[j], c[r + 1][j]))
    # gate order: in a full round the S-box gates come word 0 first
    return wires


def test_witness_model_matches_gadget_schedule():
    """Every one of the 972 gate outputs of the reference's GadgetStrategy (oracle: hades_spec.perm_gadget,
    src/strategies/gadget.rs:41-133) from the true-form rounds; all machine-word bounds and finalize32's window are asserted
    on the way."""
    rng = random.Random(53)
    assert D.WITNESS_WIRES == 972
    cases = [[5000] * 5, [P - 1, 0, 1, P - 2, 2], [0] * 5, [P - 1] * 5]
    cases += [[rng.choice(EDGE) for _ in range(5)] for _ in range(3)]
    cases += [[rng.randrange(P) for _ in range(5)] for _ in range(3)]
    for vals in cases:
        spec = []
        S.perm_gadget(vals, spec)
        got = witness_model([S.to_mont(v) for v in vals])
        assert len(got) == len(spec) == 972
        bad = [i for i in range(972) if got[i] != S.to_mont(spec[i])]
        assert not bad, bad[:10]


def test_trace_model_matches_spec_oracle():
    """k_perm_trace_fast: the true-form rounds with the five words after every round as outputs (finalize32 each)."""
    rng = random.Random(31)
    for vals in ([1] * 5, [P - 1, 0, 1, P - 2, 2], [0] * 5, [rng.randrange(P) for _ in range(5)]):
        tr, spec_tr = [], []
        witness_model([S.to_mont(v) for v in vals], tr)
        out = S.perm(vals, spec_tr)
        assert len(tr) == 67
        for r in range(67):
            assert tr[r] == [S.to_mont(v) for v in spec_tr[r]], r
        assert tr[66] == [S.to_mont(v) for v in out]


def test_finalize32_window_adversarial():
    """finalize32 on the extremes of what the witness kernel hands it: products (mont_fips of lazy operands: (-p - 2^253,
    2^253)), rows with and without an appended constant ((-2p - eps, eps)), and the window's own edges."""
    rng = random.Random(59)
    top = (P >> 3) - 1
    for v in [0, 1, -1, 31, -31, 32, top, top - 1, -2 * P - (1 << 232) + 1, -2 * P, -P, -P + 1, -P - 1, P >> 4,
              -2 * P - (1 << 246)] + \
             [rng.randrange(-2 * P, P >> 3) for _ in range(300)]:
        got = finalize32_model(D.to_balanced29_signed(v))
        assert got == v * pow(32, -1, P) % P
        if 0 <= v < (P >> 3):
            assert finalize32_model(D.to_limbs29(v)) == got                 # plain limbs, same value
    # the window's far edge (top limb about -2^28: beyond the schedule's own values, inside the kernel's 32-bit limbs)
    for v in (-30 * P + 1, -29 * P - 17, -17 * P + 5):
        x = D.to_limbs29(v % (1 << (LB * (NL - 1)))) [:NL - 1] + [v >> (LB * (NL - 1))]
        assert val(x) == v and finalize32_model(x) == v * pow(32, -1, P) % P
    # lazy limbs at the bound: a normalised value + a balanced addend
    for _ in range(50):
        a, b = rng.randrange(P >> 4), rng.randrange(P)
        x = [p + q for p, q in zip(D.to_limbs29(a), D.to_balanced29_signed(b - P))]
        assert finalize32_model(x) == (a + b - P) * pow(32, -1, P) % P


def test_finalize32_after_rows_at_the_schedule_extremes():
    """The row path at its worst: all five words U at the bottom of mont_lin's output range (about -p - 2^227, with and
    without a lazy addend of about -p: the round constant seen through the map), the one-limb digit m steered to 2^29 - 1
    (so that m p / 2^29 ~ p is subtracted), then the appended round constant at -p + 1 -- finalize32 receives about
    -2 p - 2^245, below the window the comment used to state and inside the one that is asserted."""
    rng = random.Random(67)
    lo = -P - (1 << 227) + 1
    worst = 0
    for trial in range(60):
        base = [lo + rng.randrange(1 << 20) for _ in range(5)]
        if trial % 2:                                                    # words 0..3 of a partial round carry an addend
            base = [v - P + 1 + rng.randrange(1 << 10) for v in base]
        for j in range(5):
            vals = list(base)
            if j < 4:                                                    # C[j][3 - j] = 360360 / 8 is odd: steer the digit
                k = 3 - j
                assert D.MDS_SMALL[j][k] % 2 == 1
                y0 = sum(D.MDS_SMALL[j][c] * vals[c] for c in range(5))
                vals[k] += ((MASK - rng.randrange(4) - y0) * pow(D.MDS_SMALL[j][k], -1, 1 << LB)) & MASK
            u = [D.to_balanced29_signed(v) for v in vals]
            row = mds_row_cols(u, j, 5)
            y = sum(D.MDS_SMALL[j][c] * vals[c] for c in range(5))
            assert j == 4 or (y & MASK) >= MASK - 3
            assert val(row) == (y - (y & MASK) * P) >> LB and ((y - (y & MASK) * P) & MASK) == 0
            for c in (-P + 1, -(P >> 1), 0):
                x = add_lazy(row, D.to_balanced29_signed(c)) if c else row
                worst = min(worst, val(x))
                assert finalize32_model(x) == (val(row) + c) * pow(32, -1, P) % P
    assert worst < -2 * P - (1 << 232), "the test did not reach below the old, too tight bound"
    assert worst > -2 * P - (1 << 247)


def test_per_op_models_match_spec_oracle():
    """k_states_fast / k_sbox: the trait's per-operation methods on the radix-2^29 path."""
    rng = random.Random(61)
    op = D.per_op_constants()
    mds = S.mds_matrix()
    for vals in ([1] * 5, [P - 1, 0, 1, P - 2, 2], [rng.randrange(P) for _ in range(5)]):
        st = [D.to_limbs29(S.to_mont(v)) for v in vals]
        # quintic_s_box
        assert finalize_model(sbox(st[0]), op["k"]) == S.to_mont(S.quintic_s_box(vals[0]))
        # mul_matrix
        exp = [sum(mds[i][j] * vals[j] for j in range(5)) % P for i in range(5)]
        assert [finalize_model(x, op["w"]) for x in small_mds(st)] == [S.to_mont(v) for v in exp]
        # full round body after the round key: S-box everywhere, matrix
        sb = [S.quintic_s_box(v) for v in vals]
        exp = [sum(mds[i][j] * sb[j] for j in range(5)) % P for i in range(5)]
        assert [finalize_model(x, op["w_full"]) for x in small_mds([sbox(x) for x in st])] == [S.to_mont(v) for v in exp]
        # partial round body: S-box on the last word, re-scaled with K, matrix
        pb = vals[:4] + [S.quintic_s_box(vals[4])]
        exp = [sum(mds[i][j] * pb[j] for j in range(5)) % P for i in range(5)]
        pst = st[:4] + [mont_fips(sbox(st[4]), D.to_limbs29(op["k"]))]
        assert [finalize_model(x, op["w"]) for x in small_mds(pst)] == [S.to_mont(v) for v in exp]


def test_product_bounds_adversarial():
    """Operand limbs at the lazy extremes (positive and mixed-sign): no signed 64-bit overflow."""
    hi_limb = LAZY - 1
    for pattern in ([hi_limb] * (NL - 1) + [(1 << 24) - 1],
                    [hi_limb, -(1 << 28)] * 4 + [-(1 << 24)],
                    [-(1 << 28)] * (NL - 1) + [(1 << 24)]):
        assert abs(val(pattern)) < (1 << 257)
        for sq in (False, True):
            r = mont_fips(pattern, pattern, sq)
            assert normalised(r)
            assert val(r) % P == val(pattern) * val(pattern) * pow(1 << (LB * NL), -1, P) % P


def test_raw_32bit_quotient_digit_is_not_free():
    """VERDICT r5 next #6, counted (docs/history.md section 14): taking the raw low 32 bits of the accumulator as a SIGNED
    quotient digit (p == 1 mod 2^32 as well as mod 2^29) would drop the AND of a digit column -- but the column is then
    divided by 2^29 EXACTLY only after the digit itself has been subtracted: (acc - m) / 2^29, which is NOT the arithmetic
    shift the kernel does (they differ by the signed bits 29 .. 31 of acc).  The subtraction is one more 64-bit operation
    per digit column (4 issue cycles) in place of one 32-bit AND (2): the variant is slower by count, on every product.
    This test pins the three facts the count rests on."""
    rng = random.Random(6)
    assert P % (1 << 32) == 1 and P29[0] == 1
    differ = 0
    for _ in range(2000):
        acc = rng.randrange(-(1 << 62), 1 << 62)
        low29 = acc & MASK
        m = ((acc + I31) & 0xFFFFFFFF) - I31                       # the raw low word, as the signed 32-bit operand of a mad
        assert (m - low29) % (1 << LB) == 0                        # a legal digit: same residue mod 2^29 ...
        assert (acc - m) % (1 << 32) == 0                          # ... whose multiple of p clears the whole low word
        exact = (acc - m) >> LB                                    # what the next column must start from
        assert exact == (acc >> LB) - (m >> LB)                    # = the kernel's shift MINUS the digit's top three bits
        differ += exact != acc >> LB
    assert differ > 1500                                            # 7 of 8 accumulators: never "mostly free"
    # and with the correction the reduction terms grow 4x: a lazy first operand no longer fits the signed column
    lazy_sq = NL * (LAZY - 1) ** 2
    assert lazy_sq + 8 * (1 << 31) * (1 << 28) > I63 > lazy_sq + 8 * MASK * max(abs(x) for x in P29)


def test_linear_layer_bounds_adversarial():
    big = [LAZY - 1] * (NL - 1) + [(1 << 24) - 1]
    neg = [-(1 << 28)] * (NL - 1) + [-(1 << 24)]
    for st in ([big] * 5, [neg] * 5, [big, neg, big, neg, big]):
        out = small_mds(st)
        for i in range(5):
            y = sum(D.MDS_SMALL[i][j] * val(st[j]) for j in range(5))
            assert normalised(out[i])
            assert val(out[i]) % P == y * pow(1 << LB, -1, P) % P


def test_balanced_constants():
    sch = D.fast_schedule()
    for r in range(D.ROUNDS):
        a = sch["full"][r] if r in sch["full"] else sch["part"][r][0]
        for v in a:
            limbs = D.to_balanced29(v)
            assert val(limbs) == v and all(-(1 << 28) <= l < (1 << 28) for l in limbs[:-1])


def test_schedule_tables_shape():
    sch = D.fast_schedule()
    assert len(sch["full"]) == 8 and len(sch["part"]) == 59
    assert sorted(sch["full"]) == [0, 1, 2, 3, 63, 64, 65, 66]
    assert D.MDS_L == 360360 and max(max(r) for r in D.MDS_SMALL) == 72072
    # M = lam * C: the small matrix times lam reproduces the reference matrix values
    lam = S.R * pow(D.MDS_L, -1, P) % P
    m = S.mds_matrix()
    for i in range(5):
        for j in range(5):
            assert lam * D.MDS_SMALL[i][j] % P == m[i][j]


def test_other_loader_reading_is_one_flag_away(monkeypatch):
    """If the real crate ever shows the 'howto' reading, the product tables regenerate with
    HADES252_LOADER=howto; the scale-tracked schedule stays valid (lam = 1/L) -- checked here
    value-level against the spec oracle in that mode."""
    monkeypatch.setattr(D, "LOADER", "howto")
    S.set_loader("howto")
    try:
        D.check_blobs()
        vals = [3, 1, 4, 1, 5]
        got = fast_perm_model([S.to_mont(v) for v in vals])
        assert got == [S.to_mont(v) for v in S.perm(vals)]
    finally:
        S.set_loader("from_raw")


# ---------------------------------------------------------------------------------------------
# lane-split schedule (csrc/hades_lanes.hpp): one field element on the 16 lanes of a DPP row, all-unsigned arithmetic.
# The model keeps a row as a list of 16 lane values and replays every statement of lane_mont_mul_n / lane_mds_row /
# lanes_perm with the machine-word bounds asserted (u64 columns, u32 limbs).
# ---------------------------------------------------------------------------------------------
U64, U32 = 1 << 64, 1 << 32
PINV29 = D.to_limbs29((-pow(P, -1, D.RP)) % D.RP)
LANE_IN_MAX = (1 << 30) + 66           # operand limb bound of the lane product (a product's own lazy output)


def row_shr(v, n):
    return [v[k - n] if k - n >= 0 else 0 for k in range(16)]


def row_shl(v, n):
    return [v[k + n] if k + n <= 15 else 0 for k in range(16)]


def row_bcast(v, n):
    return [v[n]] * 16


def u64(x):
    assert 0 <= x < U64, "unsigned 64-bit column overflow"
    return x


def u32(x):
    assert 0 <= x < U32, "unsigned 32-bit overflow"
    return x


def carry_split(acc):
    """carry_split of hades_lanes.hpp -> (limbs, c16, c17)."""
    lo = [a & MASK for a in acc]
    mid = [(a >> LB) & MASK for a in acc]
    tp = [a >> 58 for a in acc]
    u = [u32(x + y) for x, y in zip(mid, row_shr(tp, 1))]
    t = [u32(x + y) for x, y in zip(lo, row_shr(u, 1))]
    assert all(x < (1 << 30) + 64 for x in t)
    return t, u, tp


def carry_light(t):
    h = [x >> LB for x in t]
    return [u32((x & MASK) + y) for x, y in zip(t, row_shr(h, 1))], h


def lane_val(v):
    assert all(x == 0 for x in v[NL:]), "lanes 9..15 must be zero"
    return val(v[:NL])


def lane_mont_mul(a, b):
    """lane_mont_mul_n<1> of hades_lanes.hpp: a, b = 16-lane rows (limb k in lane k); returns the row of a value
    == a*b/2^261 (mod p) in [0, a*b/2^261 + 2.01 p)."""
    assert all(0 <= x <= LANE_IN_MAX for x in a + b) and lane_val(a) < (1 << 258) and lane_val(b) < (1 << 258)
    acc = [0] * 16
    for i in range(NL):
        ai, bi = row_bcast(a, i), row_shr(b, i)
        acc = [u64(c + x * y) for c, x, y in zip(acc, ai, bi)]
    top = [u64(x * y) for x, y in zip(row_bcast(a, NL - 1), row_shl(b, 8))]
    assert all(x == 0 for x in top[1:])
    t, c16a, c17a = carry_split(acc)
    assert val(t) + ((top[0] + c16a[15]) << (LB * 16)) + (c17a[15] << (LB * 17)) == lane_val(a) * lane_val(b)
    acc = [0] * 16
    for i in range(NL):
        acc = [u64(c + x * PINV29[i]) for c, x in zip(acc, row_shr(t, i))]
    m, _, _ = carry_split(acc)
    m = [x if k < NL else 0 for k, x in enumerate(m)]
    assert (val(m[:NL]) - val(t[:NL]) * val(PINV29)) % D.RP == 0          # M == T p' (mod 2^261)
    acc = list(t)
    for i in range(NL):
        acc = [u64(c + x * P29[i]) for c, x in zip(acc, row_shr(m, i))]
    top = [u64(x + y * P29[NL - 1]) for x, y in zip(top, row_shl(m, 8))]
    w, c16b, c17b = carry_split(acc)
    assert c17a[15] == 0 and c17b[15] == 0, "column 15 never reaches bit 58 (hades_lanes.hpp drops c17)"
    low = val(w[:NL])
    assert low in (0, D.RP, 2 * D.RP), "the low nine limbs cancel to 0, 2^261 or 2 * 2^261"
    z = [(w[k] + 3) >> LB if k == NL - 1 else 0 for k in range(16)]
    assert z[NL - 1] * D.RP == low, "limb 8 alone tells the residual carry"
    w = [u32(x + y) for x, y in zip(w, row_shr(z, 1))]
    top = [u64(x + y) for x, y in zip(top, row_shl([u32(p + q) for p, q in zip(c16a, c16b)], 15))]
    assert all(x == 0 for x in top[1:])
    r7 = [x & MASK for x in top]
    r8 = [u32(x >> LB) for x in top]
    out = [u32(x + y + zz) for x, y, zz in zip(row_shl(w, 9), row_shr(r7, 7), row_shr(r8, 8))]
    ab = lane_val(a) * lane_val(b)
    r = lane_val(out)
    assert r * D.RP == ab + val(m[:NL]) * P, "R = (ab + M p) / 2^261 exactly"
    assert r < ab // D.RP + 2 * P + (P >> 20) and all(x <= LANE_IN_MAX for x in out) and out[NL - 1] < (1 << 26)
    return out


def lane_lin(a, factor):
    """lane_lin of hades_lanes.hpp: a * factor / 2^261 (mod p) for a CONSTANT factor as a linear map by the row:
    W = sum_k a_k E_k with E_k = factor 2^(29 (k - 7)) mod p as per-lane constants (lane j: limb j), then the two-limb form
    of the row's Montgomery step: M = W p' mod 2^58 (two multiply-adds), R = (W + M p) / 2^58 (two more)."""
    assert all(0 <= x <= LANE_IN_MAX for x in a) and lane_val(a) < (1 << 258)
    e = [D.to_limbs29(factor * pow(2, LB * (k + D.LIN_STEPS - NL), P) % P) + [0] * 7 for k in range(NL)]
    acc = [0] * 16
    for k in range(NL):
        acc = [u64(c + x * y) for c, x, y in zip(acc, row_bcast(a, k), e[k])]
    t, c16, c17 = carry_split(acc)
    assert c16[15] == 0 and c17[15] == 0
    wv = sum(x << (LB * k) for k, x in enumerate(t))
    assert wv == sum(a[k] * val(e[k][:NL]) for k in range(NL))
    am = [u64(x * PINV29[0] + y * PINV29[1]) for x, y in zip(t, row_shr(t, 1))]
    m, _, _ = carry_split(am)
    mv = m[0] + (m[1] << LB)
    assert (mv - wv * val(PINV29)) % (1 << (2 * LB)) == 0 and mv < (1 << 59) + (1 << 30)      # M == W p' (mod 2^58)
    pk = P29 + [0] * 7
    acc = [u64(x + p0 * q0 + p1 * q1) for x, p0, q0, p1, q1 in zip(t, row_bcast(m, 0), pk, row_bcast(m, 1), row_shr(pk, 1))]
    w, c16, c17 = carry_split(acc)
    assert c16[15] == 0 and c17[15] == 0
    assert w[0] == 0 and w[1] & MASK == 0, "the low two limbs cancel to a multiple of 2^58"
    z = [w[k] >> LB if k == 1 else 0 for k in range(16)]
    w = [u32(x + y) for x, y in zip(w, row_shr(z, 1))]
    out = row_shl(w, 2)
    r = lane_val(out)
    assert r << (2 * LB) == wv + mv * P, "R = (W + M p) / 2^58 exactly"
    assert (r - lane_val(a) * factor * pow(D.RP, -1, P)) % P == 0
    assert r < (wv >> (2 * LB)) + 2 * P + (P >> 20) and all(x <= LANE_IN_MAX for x in out) and out[NL - 1] < (1 << 26)
    return out


def lane_sbox(v):
    v2 = lane_mont_mul(v, v)
    v4 = lane_mont_mul(v2, v2)
    return lane_mont_mul(v, v4)


def lane_mds_row(c, xs):
    """lane_mds_row of hades_lanes.hpp: (sum_j c_j x_j + m p) / 2^29, m = -Y_0 mod 2^29."""
    y = [0] * 16
    for j in range(5):
        y = [u64(a + x * c[j]) for a, x in zip(y, xs[j])]
    m = ((0 - y[0]) % U32) & MASK
    pk = P29 + [0] * 7
    y = [u64(a + m * q) for a, q in zip(y, pk)]
    assert all(a < (1 << 59) for a in y) and y[0] & MASK == 0 and all(x <= LANE_IN_MAX for v in xs for x in v)
    t = [u32((a & MASK) + b) for a, b in zip(y, row_shr([(a >> LB) & (U32 - 1) for a in y], 1))]
    w, _ = carry_light(t)
    assert w[0] == 0
    out = row_shl(w, 1)
    assert lane_val(out) << LB == sum(c[j] * lane_val(xs[j]) for j in range(5)) + m * P
    assert all(x <= (1 << LB) + 2 for x in out)
    return out


def lanes_perm_model(mont_vals):
    """Limb-exact replay of lanes_perm (hades_lanes.hpp): the coop schedule on rows, plain-limb round constants."""
    co = D.coop_schedule()
    row_of = lambda v: D.to_limbs29(v) + [0] * 7
    st = [row_of(v) for v in mont_vals]
    for r in range(D.ROUNDS):
        full = D.is_full_round(r)
        x = [[u32(a + b) for a, b in zip(st[w], row_of(co["a"][r][w]))] for w in range(5)]
        if full:
            nxt = [lane_sbox(x[w]) for w in range(5)]
        else:
            assert all(co["a"][r][w] == 0 for w in range(4))
            g = row_of(co["g"][r])
            nxt = [lane_mont_mul(x[w], g) for w in range(4)] + [lane_sbox(x[4])]
        st = [lane_mds_row(D.MDS_SMALL[i], nxt) for i in range(5)]
    return [finalize_model(x[:NL], co["final_f"]) for x in st]


def test_lanes_model_matches_spec_oracle():
    rng = random.Random(43)
    cases = [[1] * 5, [0] * 5, [P - 1] * 5, [15, 1, 2, 3, 4]]
    cases += [[rng.choice(EDGE) for _ in range(5)] for _ in range(3)]
    cases += [[rng.randrange(P) for _ in range(5)] for _ in range(3)]
    for vals in cases:
        got = lanes_perm_model([S.to_mont(v) for v in vals])
        assert got == [S.to_mont(v) for v in S.perm(vals)]


def test_lanes_product_bounds_adversarial():
    """Operand limbs at the lazy maximum 2^30 + 1 (both operands: the square of an S-box input after its round key),
    all-ones, sparse and tiny operands: no u64 / u32 overflow, exact division, zero-or-2^261 low part."""
    rng = random.Random(47)
    top = (1 << 25) - 1                                   # value < 2^258 (the kernel's values stay below 2^257)
    pats = [[LANE_IN_MAX] * (NL - 1) + [top], [MASK] * (NL - 1) + [top], [0] * NL, [1] + [0] * (NL - 1),
            [0] * (NL - 1) + [top], [LANE_IN_MAX, 0] * 4 + [top], [(1 << LB) + 2] * (NL - 1) + [top],
            [(1 << 30) + 1] * (NL - 1) + [top]]
    pats += [[rng.randrange(LANE_IN_MAX + 1) for _ in range(NL - 1)] + [rng.randrange(top)] for _ in range(40)]
    rows = [p + [0] * 7 for p in pats]
    for a in rows:
        for b in rows[:9]:
            r = lane_mont_mul(a, b)
            assert lane_val(r) % P == lane_val(a) * lane_val(b) * pow(D.RP, -1, P) % P
    for a in rows:                                        # the linear map by the row on the same operands
        for factor in (P - 1, 1, D.fast_schedule()["part"][17][1], D.fast_schedule()["part"][62][1]):
            lane_lin(a, factor)
    # linear layer at its maxima
    big = [LANE_IN_MAX] * (NL - 1) + [(1 << 26) - 1] + [0] * 7             # the result bound of a product
    for i in range(5):
        lane_mds_row(D.MDS_SMALL[i], [big] * 5)
        lane_mds_row(D.MDS_SMALL[i], [rows[0]] * 5)          # even straight after a round key


def test_lanes_tables_are_the_coop_schedule_with_plain_limbs():
    text = open(os.path.join(ROOT, "hades252_amd", "csrc", "hades_constants.inc")).read()
    assert "#define HADES_LANES_ROUND_INIT" in text and "#define HADES_NEG_PINV29" in text
    co = D.coop_schedule()
    first = [x for v in co["a"][0] for x in D.to_limbs29(v)] + D.to_limbs29(co["g"][0]) + [0] * 10
    assert "{" + ", ".join("%d" % x for x in first) + "}" in text
    assert "{" + ", ".join("%d" % x for x in PINV29) + "}" in text
    assert (val(PINV29) * P + 1) % D.RP == 0


# ------------------------------------------------------------------------------------------------------------------
# k_perm_rows (hades_lanes.hpp::rows_perm): one state per 16-lane ROW, four states per wave -- the lane arithmetic above
# under the THROUGHPUT kernel's schedule (fast_schedule: word 4 comes back to the common scale with K_r, words 0..3 meet
# only the linear layer in the partial rounds), round constants with plain limbs
# ------------------------------------------------------------------------------------------------------------------
def rows_perm_model(mont_vals):
    sch = D.fast_schedule()
    row_of = lambda v: D.to_limbs29(v) + [0] * 7
    st = [row_of(v) for v in mont_vals]
    for r in range(D.ROUNDS):
        if D.is_full_round(r):
            nxt = [lane_sbox([u32(a + b) for a, b in zip(st[w], row_of(sch["full"][r][w]))]) for w in range(5)]
        else:
            a, k = sch["part"][r]
            assert all(a[w] == 0 for w in range(4))
            v5 = lane_sbox([u32(x + y) for x, y in zip(st[4], row_of(a[4]))])
            nxt = st[:4] + [lane_lin(v5, k)]                      # K_r as a linear map by the row (round 4)
        st = [lane_mds_row(D.MDS_SMALL[i], nxt) for i in range(5)]
    return [finalize_model(x[:NL], sch["final_f"]) for x in st]


def test_rows_model_matches_spec_oracle():
    rng = random.Random(53)
    cases = [[1] * 5, [0] * 5, [P - 1] * 5, [15, 1, 2, 3, 4]]
    cases += [[rng.choice(EDGE) for _ in range(5)] for _ in range(3)]
    cases += [[rng.randrange(P) for _ in range(5)] for _ in range(3)]
    for vals in cases:
        assert rows_perm_model([S.to_mont(v) for v in vals]) == [S.to_mont(v) for v in S.perm(vals)]


def test_rows_tables_are_the_fast_schedule_with_plain_limbs():
    text = open(os.path.join(ROOT, "hades252_amd", "csrc", "hades_constants.inc")).read()
    assert "#define HADES_ROWS_ROUND_INIT" in text
    sch = D.fast_schedule()
    for r in (0, 4, 62, 66):
        a, k = (sch["full"][r], 0) if r in sch["full"] else sch["part"][r]
        rec = [x for v in a for x in D.to_limbs29(v)] + D.to_limbs29(k) + [0] * 10
        assert "{" + ", ".join("%d" % x for x in rec) + "}" in text, r
