"""CPU tier: a limb-exact Python replay of the scale-tracked kernel (hades_fast.cuh) with the
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
U64 = 1 << 64
U32 = 1 << 32


def val(limbs):
    return sum(l << (LB * k) for k, l in enumerate(limbs))


def mont_fips(a, b, sqr=False):
    """mont_fips<SQR> of hades_fast.cuh: returns limbs of a*b/2^261 (mod p), value < 2^256."""
    assert all(0 <= x < (1 << 30) for x in a + b), "input limbs must be < 2^30"
    assert val(a) < (1 << 258) and val(b) < (1 << 258)
    m = [0] * NL
    r = [0] * NL
    acc = 0
    d = [(x << 1) for x in a]
    assert all(x < U32 for x in d)
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
            assert acc < U64
        for i in range(lo, hi + 1):
            if k - i >= 1:
                acc += m[i] * P29[k - i]
                assert acc < U64
        if k < NL:
            m[k] = (-acc) & MASK
            assert (acc + m[k]) & MASK == 0
            assert acc + MASK < U64
            assert (acc + MASK) >> LB == (acc + m[k]) >> LB
            acc = (acc + MASK) >> LB
        else:
            r[k - NL] = acc & MASK
            acc >>= LB
    assert acc < U32
    r[NL - 1] = acc
    return r


def small_mds_row(st, i):
    acc = 0
    for j in range(5):
        acc += st[j][0] * D.MDS_SMALL[i][j]
    assert acc < U64
    m = (-acc) & MASK
    acc = (acc + MASK) >> LB
    r = [0] * NL
    for k in range(1, NL):
        for j in range(5):
            acc += st[j][k] * D.MDS_SMALL[i][j]
        acc += m * P29[k]
        assert acc < (1 << 60)
        r[k - 1] = acc & MASK
        acc >>= LB
    assert acc < U32
    r[NL - 1] = acc
    return r


def sbox(x):
    x2 = mont_fips(x, x, True)
    x4 = mont_fips(x2, x2, True)
    return mont_fips(x4, x)


def fast_perm_model(mont_vals):
    """mont_vals: 5 integers = in-memory BlsScalar values (value * 2^256 mod p).  Returns the same."""
    sch = D.fast_schedule()
    st = [D.to_limbs29(v) for v in mont_vals]
    for r in range(D.ROUNDS):
        full = D.is_full_round(r)
        a, k = (sch["full"][r], None) if full else sch["part"][r]
        for w in range(5):
            st[w] = [x + y for x, y in zip(st[w], D.to_limbs29(a[w]))]     # lazy ARK, limbs < 2^30
        if full:
            st = [sbox(x) for x in st]
        else:
            st[4] = mont_fips(sbox(st[4]), D.to_limbs29(k))
        st = [small_mds_row(st, i) for i in range(5)]
        for x in st:
            assert all(l < (1 << LB) for l in x) and val(x) < (1 << 256)
    f = D.to_limbs29(sch["final_f"])
    out = []
    for x in st:
        v = val(mont_fips(x, f))
        assert v < 2 * P            # one conditional subtraction suffices
        out.append(v - P if v >= P else v)
    return out


EDGE = [0, 1, P - 1, S.R, P - S.R, (1 << 255) % P, (1 << 254) - 1]


def test_model_matches_spec_oracle():
    rng = random.Random(29)
    cases = [[1] * 5, [0] * 5, [P - 1] * 5, [17] * 5]
    cases += [[rng.choice(EDGE) for _ in range(5)] for _ in range(4)]
    cases += [[rng.randrange(P) for _ in range(5)] for _ in range(6)]
    for vals in cases:
        got = fast_perm_model([S.to_mont(v) for v in vals])
        assert got == [S.to_mont(v) for v in S.perm(vals)]


def test_product_bounds_adversarial():
    """All limbs at the lazy maximum 2^30 - 1 except the top ones (value < 2^258): no column overflow."""
    big = [(1 << 30) - 1] * (NL - 1) + [(1 << 25) - 1]
    assert val(big) < (1 << 258)
    for sq in (False, True):
        r = mont_fips(big, big, sq)
        assert val(r) < (1 << 256)
        assert val(r) % P == val(big) * val(big) * pow(1 << (LB * NL), -1, P) % P


def test_linear_layer_bounds_adversarial():
    big = [(1 << 30) - 1] * (NL - 1) + [(1 << 24) - 1]      # value < 2^256
    st = [big] * 5
    for i in range(5):
        r = small_mds_row(st, i)
        y = sum(D.MDS_SMALL[i][j] for j in range(5)) * val(big)
        assert val(r) < (1 << 256)
        assert val(r) % P == y * pow(1 << LB, -1, P) % P


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
