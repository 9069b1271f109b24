"""CPU tier: rust/tests/kat_scalar.rs -- the GPU-free known-answer test of the reference's own `ScalarStrategy::perm`
(src/strategies/scalar.rs:52-74 is the only test of that path and pins no value) -- is current, carries exactly the
vectors of tests/golden/kat.json, and those vectors are what the oracle computes.

No Rust toolchain exists in this image, so the file has never met a compiler; what can be checked here is:
* the committed file equals what tools/gen_rust_kat.py renders from kat.json (staleness);
* every embedded vector, parsed back out of the Rust text, is re-derived from oracle/hades_spec.py;
* the inlined SHA-256 (rendered from the same constants and mirrored statement for statement by `py_sha256`) agrees
  with hashlib, and the two batch digests are those of the oracle's generator-A / -B batches;
* brackets balance outside strings and comments in every shipped .rs file, and the test only uses items the reference
  itself uses or exports (so that the first `cargo test` fails on values, not on names).
"""
import hashlib
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_rust_kat as G  # noqa: E402
import hades_spec as S  # noqa: E402

RS = os.path.join(ROOT, "rust", "tests", "kat_scalar.rs")
with open(RS) as _f:
    TEXT = _f.read()
with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as _f:
    KAT = json.load(_f)

LIMBS = r"\[(0x[0-9a-f]{16}), (0x[0-9a-f]{16}), (0x[0-9a-f]{16}), (0x[0-9a-f]{16})\]"


def val(m):
    return sum(int(m[k], 16) << (64 * k) for k in range(4))


def rows(field, text):
    out = []
    for blk in re.findall(r"\b%s: \[\n((?:\s+%s,\n){5})\s+\]" % (field, LIMBS.replace("(", "(?:")), text):
        out.append([val(m) for m in re.findall(LIMBS, blk)])
    return out


def test_committed_file_is_what_the_generator_renders():
    assert TEXT == G.render(), "rust/tests/kat_scalar.rs is stale: python tools/gen_rust_kat.py"


def test_single_vectors_are_the_golden_file_and_the_oracle():
    ins, ins_m, outs, outs_m = (rows(f, TEXT) for f in ("input", "input_mont", "output", "output_mont"))
    assert len(ins) == len(ins_m) == len(outs) == len(outs_m) == len(KAT["single"]) == 11
    traces = re.findall(r"trace: \[\n((?:\s+\(\d+, %s, %s\),\n)+)\s+\]," % ((LIMBS.replace("(", "(?:"),) * 2), TEXT)
    assert len(traces) == 11
    for k, s in enumerate(KAT["single"]):
        assert ins[k] == [int(v, 16) for v in s["in"]] and outs[k] == [int(v, 16) for v in s["out"]]
        tr = []
        assert S.perm(ins[k], tr) == outs[k]                       # oracle, not just the json
        assert ins_m[k] == [S.to_mont(v) for v in ins[k]] and outs_m[k] == [S.to_mont(v) for v in outs[k]]
        entries = re.findall(r"\((\d+), %s, %s\)" % (LIMBS, LIMBS), traces[k])
        assert [int(e[0]) for e in entries] == [1, 4, 5, 63, 64, 67]
        for e in entries:
            r = int(e[0])
            assert val(e[1:5]) == tr[r - 1][0] and val(e[5:9]) == tr[r - 1][4]
    # the reference's own test inputs are among them: hades_det (scalar.rs:64-66), preimage_constant (gadget.rs:230)
    assert [17] * 5 in ins and [19] * 5 in ins and [5000] * 5 in ins


def const(name):
    m = re.search(r"const %s: \[u64; 4\] = %s;" % (name, LIMBS), TEXT)
    return val(m.groups())


def const_bytes(name):
    m = re.search(r"const %s: \[u8; 32\] = \[(.*?)\];" % name, TEXT, flags=re.S)
    return bytes(int(x, 16) for x in re.findall(r"0x([0-9a-f]{2})", m.group(1)))


def test_constants_and_disambiguator():
    assert const("R_INV") * S.R % S.P == 1 and const("R_MONT") == S.R
    assert const("ONES_FROM_RAW") == S.perm([1] * 5)[0] == int(KAT["single"][0]["out"][0], 16)
    # the other reading of the blobs (file limbs taken as already-Montgomery), computed, not remembered
    S.set_loader("howto")
    try:
        assert const("ONES_HOWTO") == S.perm([1] * 5)[0]
    finally:
        S.set_loader("from_raw")
    assert const("ONES_HOWTO") != const("ONES_FROM_RAW")


def test_inlined_sha256_and_batch_digests():
    k, h = G.sha256_k()
    assert [int(x, 16) for x in re.search(r"SHA256_H: \[u32; 8\] = \[(.*?)\];", TEXT).group(1).split(", ")] == h
    ks = re.search(r"SHA256_K: \[u32; 64\] = \[(.*?)\];", TEXT, flags=re.S).group(1)
    assert [int(x, 16) for x in re.findall(r"0x[0-9a-f]{8}", ks)] == k
    for d in (b"", b"abc", b"a" * 55, b"a" * 56, b"a" * 63, b"a" * 64, b"a" * 119, bytes(range(256)) * 9):
        assert G.py_sha256(d) == hashlib.sha256(d).digest()
    assert const_bytes("SHA256_ABC") == hashlib.sha256(b"abc").digest()
    assert const_bytes("SHA256_LONG") == hashlib.sha256(b"Hades252 " * 1000).digest()
    # the statements of the Rust block function are the statements of the Python twin
    for stmt in ("w[t - 15].rotate_right(7) ^ w[t - 15].rotate_right(18) ^ (w[t - 15] >> 3)",
                 "w[t - 2].rotate_right(17) ^ w[t - 2].rotate_right(19) ^ (w[t - 2] >> 10)",
                 "v[4].rotate_right(6) ^ v[4].rotate_right(11) ^ v[4].rotate_right(25)",
                 "(v[4] & v[5]) ^ (!v[4] & v[6])",
                 "v[0].rotate_right(2) ^ v[0].rotate_right(13) ^ v[0].rotate_right(22)",
                 "(v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2])",
                 "[t1.wrapping_add(t2), v[0], v[1], v[2], v[3].wrapping_add(t1), v[4], v[5], v[6]]"):
        assert stmt in TEXT
    # (the full 1024-state batches behind the two digests are recomputed by tests/test_oracle.py)
    for name, gen in (("GEN_A", S.gen_a_element), ("GEN_B", S.gen_b_element)):
        g = KAT[name.lower()]
        assert const_bytes(name + "_SHA256_IN").hex() == g["sha256_in"]
        assert const_bytes(name + "_SHA256_OUT").hex() == g["sha256_out"]
        first = gen(0)
        assert const(name + "_W0_IN_MONT") == sum(first[k] << (64 * k) for k in range(4))
        limbs = [x for w in range(5) for x in gen(w)]
        res = S.perm_mont_limbs(limbs)
        assert const(name + "_W0_OUT_MONT") == sum(res[k] << (64 * k) for k in range(4))


def test_generator_b_text_is_the_spec():
    """The Rust generator is the splitmix64 finaliser of oracle/hades_spec.py::splitmix_limb, constant for constant."""
    for c in ("0x4861646573323532", "0x9E3779B97F4A7C15", "0xBF58476D1CE4E5B9", "0x94D049BB133111EB", "0x3FFF_FFFF_FFFF_FFFF"):
        assert c in TEXT
    assert "4u64.wrapping_mul(e).wrapping_add(k).wrapping_add(1)" in TEXT
    assert "(z ^ (z >> 30))" in TEXT and "(z ^ (z >> 27))" in TEXT and "z ^ (z >> 31)" in TEXT
    assert S.GEN_SEED == 0x4861646573323532


def test_caller_shape_vectors_are_the_oracles():
    """Merkle roots and sponge digests embedded for `kat_merkle4_roots_over_perm` / `kat_sponge_over_perm`: parsed back out of
    the Rust text and re-derived from oracle/hades_spec.py; the Rust restatements of node / sponge carry the spec's steps."""
    def leaf(e):
        l = S.gen_b_element(e)
        return S.from_mont(sum(l[k] << (64 * k) for k in range(4)))
    roots = re.findall(r"^    \((\d+), %s\),$" % LIMBS, TEXT, flags=re.M)
    assert [int(r[0]) for r in roots] == [4, 16, 64, 256]
    for r in roots:
        n = int(r[0])
        assert val(r[1:5]) == S.to_mont(S.merkle4_root([leaf(e) for e in range(n)], 15, 1)) == int(
            KAT["merkle4_root_mont"]["leaves_gen_b"][str(n)], 16)
    vecs = re.findall(r"^    \((\d+), (\d+), %s, (true|false), %s\),$" % (LIMBS, LIMBS), TEXT, flags=re.M)
    assert len(vecs) == len(KAT["sponge"]["vectors"]) == 32
    seen = set()
    for v, k in zip(vecs, KAT["sponge"]["vectors"]):
        first, length, cap, pad = int(v[0]), int(v[1]), val(v[2:6]), v[6] == "true"
        assert (first, length, cap, 1 if pad else 0) == (k["first_elem"], k["len"], int(k["capacity"], 16), k["pad_mode"])
        msg = [leaf(first + e) for e in range(length)]
        assert val(v[7:11]) == S.to_mont(S.sponge_hash(msg, cap, 1 if pad else 0)) == int(k["digest_mont"], 16)
        seen.add((length, cap, pad))
    assert len(seen) == 32 and {l for l, _, _ in seen} == {0, 1, 3, 4, 5, 8, 9, 17}
    for stmt in ("let mut state = [tag, c[0], c[1], c[2], c[3]];", "state[out_idx]", "state[1 + k] += block[k];",
                 "padded.push(BlsScalar::from(1u64));", "while padded.len() % 4 != 0", "state[1]\n}",
                 "BlsScalar::from(15u64), 1)"):
        assert stmt in TEXT, stmt


def strip_strings_and_comments(text):
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r'b?"(?:\\.|[^"\\])*"', '""', text, flags=re.S)
    return re.sub(r"'(?:\\.|[^'\\])'", "' '", text)


def rust_sources():
    for d, _, files in os.walk(os.path.join(ROOT, "rust")):
        for f in files:
            if f.endswith(".rs"):
                yield os.path.join(d, f)


def test_brackets_balance_in_every_rust_file():
    pairs = {")": "(", "]": "[", "}": "{"}
    seen = 0
    for path in rust_sources():
        stack = []
        for ch in strip_strings_and_comments(open(path).read()):
            if ch in "([{":
                stack.append(ch)
            elif ch in pairs:
                assert stack and stack.pop() == pairs[ch], path
        assert not stack, path
        seen += 1
    assert seen >= 4


def test_rust_sources_lex_cleanly():
    """No Rust parser exists in this image, but pygments ships a Rust LEXER: every shipped .rs file must tokenise without a
    single error token (unterminated strings / comments, malformed literals, stray characters)."""
    pygments = pytest.importorskip("pygments")
    from pygments.lexers import RustLexer
    from pygments.token import Error
    for path in rust_sources():
        toks = list(RustLexer().get_tokens(open(path).read()))
        bad = [v for t, v in toks if t in Error]
        assert not bad and len(toks) > 500, (path, bad[:5])


def test_only_names_the_reference_uses_or_exports():
    code = strip_strings_and_comments(TEXT)
    # dusk_hades exports (src/lib.rs:20-31); BlsScalar calls the reference itself makes (round_constants.rs:31,41,61-62;
    # scalar.rs:64; assets/HOWTO.md:45 for internal_repr)
    m = re.search(r"use dusk_hades::\{(.*?)\};", code)
    assert set(x.strip() for x in m.group(1).split(",")) <= {"ScalarStrategy", "Strategy", "WIDTH", "TOTAL_FULL_ROUNDS",
                                                             "PARTIAL_ROUNDS"}
    assert set(re.findall(r"BlsScalar::(\w+)", code)) <= {"from_raw", "zero", "from"}
    scalar_methods = set(re.findall(r"\b(?:s|one|x|r_inv|state\[\w+\])\.(\w+)\(", code))
    assert {"internal_repr", "to_bytes"} <= scalar_methods | set(re.findall(r"\.(\w+)\(\)", code))
    assert "#![allow(deprecated)]" in TEXT                          # src/lib.rs:10 marks the whole crate deprecated
    assert 'include_bytes!("../assets/ark.bin")' in TEXT            # tests/ -> crate root, like src/round_constants.rs:30
    for t in ("layout_and_montgomery_form", "loader_reading_of_the_constant_blobs", "kat_single_states",
              "kat_round_intermediates", "sha256_self_test", "kat_batch_generator_a", "kat_batch_generator_b",
              "kat_merkle4_roots_over_perm", "kat_sponge_over_perm"):
        assert re.search(r"#\[test\]\s*fn %s\(\)" % t, TEXT), t
