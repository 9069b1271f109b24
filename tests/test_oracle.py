"""CPU tier: the oracles against the golden vectors and against each other (no GPU)."""
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

EDGE = [0, 1, 2, P - 1, P - 2, R, (R * R) % P, (1 << 255) % P, (1 << 254), 0xFFFFFFFF, 0xFFFFFFFF00000001 % P,
        (P - 1) // 2, (P + 1) // 2, 0xFFFFFFFFFFFFFFFF, (1 << 128) - 1, P - (1 << 32)]


def test_blobs_match_reference_hashes():
    # the regenerated blobs ARE the reference's assets (sha256 from SURVEY.md section 4)
    assert hashlib.sha256(S.ark_blob()).hexdigest() == "78c427449282315729eaa2e39e1937e0aa0b010c4c38bcbb1d57016011880485"
    assert hashlib.sha256(S.mds_blob()).hexdigest() == "131915cbeae1bde75422cce7fcf7feb9223a4dec370a937a2133c1f998ded0e7"
    assert len(S.ark_blob()) == 30720 and len(S.mds_blob()) == 800


def test_round_constants_like_reference_test():
    # reference src/round_constants.rs:55-65: non-zero, to_bytes/from_bytes round trip (=> all < p)
    for c in S.round_constants():
        assert 0 < c < P


def test_mds_structure():
    # loader semantics: value = R/(i+j+5); Hankel with 9 distinct entries
    m = S.mds_matrix()
    for i in range(5):
        for j in range(5):
            assert m[i][j] == R * pow(i + j + 5, -1, P) % P
    assert len({v for row in m for v in row}) == 9


def test_spec_survey_anchors():
    # SURVEY.md section 8(a) anchors, recorded before this repo existed
    tr = []
    out = S.perm([1] * 5, tr)
    assert out[0] == 0x71a5b8040ed5c21f5900c854f34748e89dfb577514b9bd816e62e1b3e3f039c3
    assert out[4] == 0x4390d7dec01afe00e2f7e5148b8070d99021df24b53d4bffec7d42433e4b8ca2
    assert S.to_mont(out[0]) == 0x23338e018f505a2a832c33cbf2dd481f2409c7dd1a61ab1c935feb66a5e6cf3c
    assert tr[0][0] == 0x3511a8142a89e25d8fa1832cd2280ede04d17a7591b277bb1f1e782e42f37fe2
    assert tr[3][4] == 0x18c40abeb49493bfb4ec60cde628c1b7ce6658dbf98791efa5c408b640c12c5c
    assert tr[4][0] == 0x31264961a5a7f2d097f7125c6d2c23bf2fb7a24c77e223803aa97834bc038f2c
    assert tr[62][4] == 0x59748fcf48a245d29a7cd4ab7aa2dac637bbc066a5155c51941a804cf2b9675a
    assert tr[63][0] == 0x332763bdeced1e5fc2d01933f9feca59357777bfa31b16d5648d827ee12e7c1a
    assert S.perm([17] * 5) == [
        0x4a335a5be470b8c178e7e78dfd8abcedee607c75afbff0491c074bae3415b320,
        0x04f108127cc563090c4724a4c394334fd38b6b59654e38fae442351793024684,
        0x4c5a86584cb6661cce9074cc64d18d56aaf1dc1a0c6c0dae0319a5afcd6c1033,
        0x432c2c79d317cc36030483f9b06879dce6f0b7c5a421555ee32de0dbb8fb5444,
        0x5e0f4e5bf6fa474cf727ce87dd64e6a4753f60758bb8273e04715a469ab14f91]
    assert S.perm([0] * 5)[0] == 0x4448679e00a28dd381089245efaab4249e99c5825ceec146d8aac63a3c3bbc95
    assert S.perm([5000] * 5)[0] == 0x246568a8dca8b3c5e44d952f8816bb6a40d6fb81c9df08af255afbc1cd4fe26e
    assert S.perm([15, 1, 2, 3, 4])[1] == 0x161a1c0e8772e21b8165b88e9f852de875b1d14774a1b186e98319f169a5a57f


def test_hades_det_like_reference(oracle):
    # reference src/strategies/scalar.rs:62-74
    x = np.array(sum([limbs_of(S.to_mont(17))] * 5, []), dtype=np.uint64)
    z = np.array(sum([limbs_of(S.to_mont(19))] * 5, []), dtype=np.uint64)
    a, b, c = oracle.perm_batch(x, 1), oracle.perm_batch(x, 1), oracle.perm_batch(z, 1)
    assert (a == b).all() and not (a == c).all()


def test_scalar_equals_gadget_like_reference():
    """The reference's `preimage` / `preimage_constant` tests (src/strategies/gadget.rs:207-244) assert, inside a
    proof, that GadgetStrategy's witness values -- a DIFFERENT schedule of the same permutation: round keys after
    the first are appended to the previous linear layer, gadget.rs:41-57, :71-133 -- equal ScalarStrategy::perm.
    Same check between the two restatements: it is the one cross-check of the round / constant schedule the
    reference itself holds."""
    rng = random.Random(2024)
    cases = [[5000] * 5, [17] * 5, [0] * 5, [P - 1] * 5]          # [5000;5]: gadget.rs:230
    cases += [[rng.randrange(P) for _ in range(5)] for _ in range(8)]
    for vals in cases:
        wires = []
        assert S.perm_gadget(vals, wires) == S.perm(vals)
        # (CHANGELOG.md:134-135 counts 973 gates per permutation; this schedule has 972 gate outputs)
        assert len(wires) == 5 + 3 * (8 * 5 + 59) + 10 * 67 == 972


def test_wrong_width_rejected():
    # the reference panics for len != WIDTH (scalar.rs:48)
    with pytest.raises(ValueError):
        S.perm([1, 2, 3, 4])


def test_c_oracle_single_kats(oracle, kat):
    for s in kat["single"]:
        st = np.array(sum([limbs_of(int(x, 16)) for x in s["in_mont"]], []), dtype=np.uint64)
        exp = np.array(sum([limbs_of(int(x, 16)) for x in s["out_mont"]], []), dtype=np.uint64)
        out, tr = oracle.perm_trace(st)
        assert (out == exp).all()
        assert (oracle.perm_batch(st, 1) == exp).all()
        for rnd, (w0, w4) in s["trace_w0_w4"].items():
            r = int(rnd) - 1
            assert S.from_mont(int_of(tr[r][0])) == int(w0, 16)
            assert S.from_mont(int_of(tr[r][4])) == int(w4, 16)


def test_spec_single_kats(kat):
    for s in kat["single"]:
        assert [hex(v) for v in S.perm([int(x, 16) for x in s["in"]])] == s["out"]


@pytest.mark.parametrize("name", ["gen_a", "gen_b"])
def test_c_oracle_batch_digests(oracle, kat, name):
    n = kat[name]["n"]
    buf = oracle.gen_a(0, 5 * n) if name == "gen_a" else oracle.gen_b(0, 5 * n)
    assert hashlib.sha256(buf.tobytes()).hexdigest() == kat[name]["sha256_in"]
    assert hex(int_of(buf[:4])) == kat[name]["perm0_word0_in_mont"]
    out = oracle.perm_batch(buf)
    assert hashlib.sha256(out.tobytes()).hexdigest() == kat[name]["sha256_out"]
    assert hex(int_of(out[:4])) == kat[name]["perm0_word0_out_mont"]
    # thread count must not matter
    assert (oracle.perm_batch(buf, 1)[: 20 * 64] == out[: 20 * 64]).all()


def test_generators_match_spec(oracle):
    b = oracle.gen_b(7, 9)
    for e in range(9):
        assert list(map(int, b[4 * e:4 * e + 4])) == S.gen_b_element(7 + e)
        assert int_of(b[4 * e:4 * e + 4]) < P
    a = oracle.gen_a(3, 5)
    for e in range(5):
        assert list(map(int, a[4 * e:4 * e + 4])) == S.gen_a_element(3 + e)


def test_c_oracle_field_ops_vs_bigint(oracle):
    rng = random.Random(252)
    vals = EDGE + [rng.randrange(P) for _ in range(200)]
    for a in vals:
        assert oracle.fr1("square", a) == a * a * S.R_INV % P
        assert oracle.fr1("from_raw", a) == a * R % P
        assert oracle.fr1("to_canonical", a) == a * S.R_INV % P
    for _ in range(2000):
        a, b = rng.choice(vals), rng.choice(vals)
        assert oracle.fr2("add", a, b) == (a + b) % P
        assert oracle.fr2("mul", a, b) == a * b * S.R_INV % P


def test_c_oracle_field_ops_hypothesis(oracle):
    from hypothesis import given, settings, strategies as st
    elem = st.one_of(st.sampled_from(EDGE), st.integers(min_value=0, max_value=P - 1))

    @settings(max_examples=300, deadline=None)
    @given(elem, elem)
    def prop(a, b):
        assert oracle.fr2("add", a, b) == (a + b) % P
        assert oracle.fr2("mul", a, b) == a * b * S.R_INV % P
        assert oracle.fr1("square", a) == oracle.fr2("mul", a, a)
    prop()


def test_c_oracle_tables_follow_loader(oracle):
    ark, mds = S.round_constants(), S.mds_matrix()
    for i in (0, 1, 4, 5, 334, 335, 959):
        assert oracle.round_constant(i) == ark[i] * R % P
    for i in range(5):
        for j in range(5):
            assert oracle.mds(i, j) == mds[i][j] * R % P


def test_c_oracle_per_op_vs_spec(oracle):
    rng = random.Random(5)
    vals = [rng.randrange(P) for _ in range(10)]
    st = np.array(sum([limbs_of(S.to_mont(v)) for v in vals], []), dtype=np.uint64)
    ark, mds = S.round_constants(), S.mds_matrix()
    out = oracle.add_round_key(st, 7).reshape(-1, 4)
    for k, v in enumerate(vals):
        assert S.from_mont(int_of(out[k])) == (v + ark[35 + k % 5]) % P
    out = oracle.quintic_s_box(st).reshape(-1, 4)
    for k, v in enumerate(vals):
        assert S.from_mont(int_of(out[k])) == pow(v, 5, P)
    out = oracle.mul_matrix(st).reshape(-1, 5, 4)
    for s in range(2):
        w = vals[5 * s:5 * s + 5]
        S.mul_matrix(w, mds)
        assert [S.from_mont(int_of(out[s][k])) for k in range(5)] == w
    # composed rounds == spec trace
    tr = []
    S.perm(vals[:5], tr)
    s0 = st[:20]
    assert [S.from_mont(int_of(x)) for x in oracle.full_round(s0, 0).reshape(5, 4)] == tr[0]
    s4 = np.array(sum([limbs_of(S.to_mont(v)) for v in tr[3]], []), dtype=np.uint64)
    assert [S.from_mont(int_of(x)) for x in oracle.partial_round(s4, 4).reshape(5, 4)] == tr[4]


def test_bytes_format(oracle):
    rng = random.Random(9)
    for v in EDGE + [rng.randrange(P) for _ in range(50)]:
        rc, limbs = oracle.from_bytes(v.to_bytes(32, "little"))
        assert rc == 0 and int_of(limbs) == v * R % P
        assert oracle.to_bytes(limbs) == v.to_bytes(32, "little")
    for bad in (P, P + 1, (1 << 256) - 1):
        rc, _ = oracle.from_bytes(bad.to_bytes(32, "little"))
        assert rc == -1


def test_merkle_golden(oracle, kat):
    g = kat["merkle4_root_mont"]
    tag = S.to_mont(g["tag"])
    for n_str, root_hex in g["leaves_gen_b"].items():
        n = int(n_str)
        leaves = oracle.gen_b(0, n)
        root = oracle.merkle4_root(leaves, tag, g["out_idx"])
        assert hex(int_of(root)) == root_hex


def test_loader_semantics_disambiguator():
    """The reference's loaders call from_raw on blobs that hold Montgomery limbs.
    Both readings are computable; one run of the real crate on perm([1;5]) tells them apart."""
    try:
        assert S.perm([1] * 5)[0] == 0x71a5b8040ed5c21f5900c854f34748e89dfb577514b9bd816e62e1b3e3f039c3
        S.set_loader("howto")
        assert S.perm([1] * 5)[0] == 0x5221c7bb3c002df76daf1d97d2eef86392182eee91e0554079095df74aca0c56
        assert S.mds_matrix()[0][0] == pow(5, -1, P)
    finally:
        S.set_loader("from_raw")
    assert S.perm([1] * 5)[0] == 0x71a5b8040ed5c21f5900c854f34748e89dfb577514b9bd816e62e1b3e3f039c3


def test_sponge_c_oracle_vs_spec(oracle):
    rng = random.Random(4)
    cap = 1 << 64
    for length in (1, 3, 4, 5, 8, 9):
        for pad in (0, 1):
            msgs = [[rng.randrange(P) for _ in range(length)] for _ in range(3)]
            flat = np.array([l for m in msgs for v in m for l in limbs_of(S.to_mont(v))], dtype=np.uint64)
            got = oracle.sponge(flat, length, S.to_mont(cap), pad).reshape(-1, 4)
            for i, m in enumerate(msgs):
                assert S.from_mont(int_of(got[i])) == S.sponge_hash(m, cap, pad)
    # padding 1 makes a message and its zero-extended version hash differently
    assert S.sponge_hash([7], cap, 1) != S.sponge_hash([7, 0], cap, 1)
    assert S.sponge_hash([7], cap, 0) == S.sponge_hash([7, 0], cap, 0)


def test_config5_golden_digest_is_the_oracles(oracle, kat):
    """BASELINE configs[4] at FULL size against the oracle (tests/golden/kat.json config5_2p30): the digest the C oracle
    computed over all 2^30 outputs (tools/oracle_config5_digest.py, ~1 h on 16 cores; profiles/r5/oracle_config5_digest.log)
    equals the digest of the device's outputs recorded since round 2 -- whole range and rank 7's shard -- and is the wrapping
    sum of the eight shard digests bench.py combines at world size 8.  A 4 096-state probe of that run (shard 5's first
    states, global indices) is recomputed here, so the committed value is tied to this tree's oracle and digest code."""
    import oracle_lib
    rec = kat["config5_2p30"]
    assert rec["n"] == 1 << 30 and rec["oracle_digest"] == rec["digest"]
    assert rec["oracle_shard_digests"][7] == rec["rank7_shard_digest"] and len(rec["oracle_shard_digests"]) == 8
    acc = [0, 0, 0, 0]
    for sd in rec["oracle_shard_digests"]:
        acc = [(a + int(x, 16)) & ((1 << 64) - 1) for a, x in zip(acc, sd)]
    assert ["%016x" % a for a in acc] == rec["oracle_digest"]
    pr = rec["oracle_probe"]
    out = oracle.perm_batch(oracle.gen_b(5 * pr["first_state"], 5 * pr["n"]))
    assert ["%016x" % x for x in oracle_lib.digest_ref(out, 20 * pr["first_state"])] == pr["digest"]


def test_headline_block_digests_are_consistent_with_the_config5_run(oracle, kat):
    """tests/golden/kat.json headline_2p26_blocks: the oracle's digest of ALL outputs of every 2^26-state block 0 .. 7 (what
    rank g of an N-GPU bench run holds after its first launch; tools/oracle_block_digests.py).  Two consecutive blocks add
    up to one 2^27 shard of the configs[4] run -- computed by a different run with a different driver loop -- and a
    probe of block 3 is recomputed here."""
    import oracle_lib
    rec, c5 = kat["headline_2p26_blocks"], kat["config5_2p30"]
    assert rec["block_states"] == 1 << 26 and sorted(rec["blocks"], key=int) == [str(g) for g in range(8)]
    m64 = (1 << 64) - 1
    for g in range(4):
        pair = [(int(x, 16) + int(y, 16)) & m64 for x, y in zip(rec["blocks"][str(2 * g)], rec["blocks"][str(2 * g + 1)])]
        assert ["%016x" % v for v in pair] == c5["oracle_shard_digests"][g], g
    pr = rec["probe"]
    out = oracle.perm_batch(oracle.gen_b(5 * pr["first_state"], 5 * pr["n"]))
    assert ["%016x" % x for x in oracle_lib.digest_ref(out, 20 * pr["first_state"])] == pr["digest"]


def test_sponge_golden_vectors_through_the_c_oracle(oracle, kat):
    """tests/golden/kat.json `sponge` (written from the big-integer spec; embedded in rust/tests/kat_scalar.rs for the real
    crate's `perm` to confirm): the C oracle's variable-length sponge on the same generator-B messages."""
    vecs = kat["sponge"]["vectors"]
    assert len(vecs) == 32
    for v in vecs:
        msg = oracle.gen_b(v["first_elem"], v["len"])
        cap = S.to_mont(int(v["capacity"], 16))
        got = oracle.sponge_var(msg if v["len"] else np.zeros(4, dtype=np.uint64), [0], [v["len"]], cap, v["pad_mode"])
        assert int_of(got) == int(v["digest_mont"], 16), v
