"""Pin #0: the C oracle's field arithmetic reproduces bytes the REAL field crate wrote.

The only reference-held artefacts that carry field-arithmetic results are the two blobs
`assets/ark.bin` / `assets/mds.bin`: they were written by dusk-bls12_381 itself
(`assets/HOWTO.md:21-48` and `:71-108`: `internal_repr()` = Montgomery limbs of values
computed with that crate's `+`, `from_bytes_wide`, `From<u64>` and `invert`).  Here every one of
those 985 x 32 bytes is regenerated THROUGH `oracle/hades_oracle.c`'s exported field operations
(`fr_add`, `fr_mul`, `fr_square`, `fr_from_raw`) -- not through Python big integers -- and must
equal the blob: sha256 always (hashes recorded in SURVEY.md section 4), byte for byte when
/root/reference is present.  A wrong carry, reduction or conditional subtraction in the oracle's
Montgomery arithmetic cannot survive ~19 000 chained operations that end on these bytes.

  ark.bin  c_0 = from_bytes_wide(SHA512("poseidon-for-plonk")) + 1, c_i = from_bytes_wide(SHA512(prev digest)) + c_{i-1}
           from_bytes_wide(lo || hi) = lo * R^2 + hi * R^3  (Montgomery products; the published
           algorithm of the crate's `from_u512`)                                   HOWTO.md:21-39
  mds.bin  M[i][j] = (from(i) + from(j + 5)).invert(): Fermat x^(p-2) by square-and-multiply
           with fr_square / fr_mul (the inverse is unique, whatever addition chain the crate uses)
                                                                                   HOWTO.md:71-97
The loader (`from_raw` of each chunk, src/round_constants.rs:29-48, src/mds_matrix.rs:18-40) is then
applied through the oracle as well and must give the oracle's own constant tables.
"""
import hashlib
import os

from oracle_lib import P, R, limbs_of

ARK_SHA256 = "78c427449282315729eaa2e39e1937e0aa0b010c4c38bcbb1d57016011880485"
MDS_SHA256 = "131915cbeae1bde75422cce7fcf7feb9223a4dec370a937a2133c1f998ded0e7"
REF_ASSETS = "/root/reference/assets"


def le32(v):
    return b"".join(int(l).to_bytes(8, "little") for l in limbs_of(v))


def oracle_ark_blob(orc):
    r2 = orc.fr1("from_raw", R)                 # from_raw(R) = R * R^2 / R = R^2 mod p, via the oracle
    assert r2 == R * R % P
    r3 = orc.fr2("mul", r2, r2)                 # R^4 / R
    one = orc.fr1("from_raw", 1)                # BlsScalar::one()
    prev, data, out = one, b"poseidon-for-plonk", []
    for _ in range(960):
        data = hashlib.sha512(data).digest()
        lo, hi = int.from_bytes(data[:32], "little"), int.from_bytes(data[32:], "little")
        wide = orc.fr2("add", orc.fr2("mul", lo, r2), orc.fr2("mul", hi, r3))
        prev = orc.fr2("add", wide, prev)
        out.append(prev)
    return out


def oracle_invert(orc, x):
    """x^(p-2) in Montgomery form, left-to-right square-and-multiply through the oracle."""
    e = P - 2
    acc = x
    for bit in bin(e)[3:]:
        acc = orc.fr1("square", acc)
        if bit == "1":
            acc = orc.fr2("mul", acc, x)
    return acc


def oracle_mds_blob(orc):
    out = []
    for i in range(5):
        for j in range(5):
            s = orc.fr2("add", orc.fr1("from_raw", i), orc.fr1("from_raw", j + 5))
            out.append(oracle_invert(orc, s))
    return out


def test_ark_blob_through_c_oracle_field_ops(oracle):
    limbs = oracle_ark_blob(oracle)
    blob = b"".join(le32(v) for v in limbs)
    assert len(blob) == 30720
    assert hashlib.sha256(blob).hexdigest() == ARK_SHA256
    path = os.path.join(REF_ASSETS, "ark.bin")
    if os.path.exists(path):
        assert blob == open(path, "rb").read()
    # the loader, through the oracle: chunk -> from_raw -> ROUND_CONSTANTS[i]
    for i in (0, 1, 4, 5, 334, 335, 959):
        assert oracle.fr1("from_raw", limbs[i]) == oracle.round_constant(i)


def test_mds_blob_through_c_oracle_field_ops(oracle):
    limbs = oracle_mds_blob(oracle)
    blob = b"".join(le32(v) for v in limbs)
    assert len(blob) == 800
    assert hashlib.sha256(blob).hexdigest() == MDS_SHA256
    path = os.path.join(REF_ASSETS, "mds.bin")
    if os.path.exists(path):
        assert blob == open(path, "rb").read()
    for i in range(5):
        for j in range(5):
            assert oracle.fr1("from_raw", limbs[5 * i + j]) == oracle.mds(i, j)
            # and it really is the inverse: x * x^-1 = one, through the oracle
            s = oracle.fr2("add", oracle.fr1("from_raw", i), oracle.fr1("from_raw", j + 5))
            assert oracle.fr2("mul", s, limbs[5 * i + j]) == R


# ---------------------------------------------------------------------------------------------
# The reference's own constant test (src/round_constants.rs:55-65: every constant is non-zero and
# survives to_bytes -> from_bytes), made absolute: under the loader's from_raw reading the canonical
# bytes of ROUND_CONSTANTS[i] ARE chunk i of assets/ark.bin (and MDS_MATRIX[i][j] chunk 5i+j of
# assets/mds.bin).  CPU twin of tests/test_gpu_f3_wire.py::test_wire_format_pinned_to_reference_blobs.
# ---------------------------------------------------------------------------------------------
def blob_bytes(name):
    """The blob, regenerated with plain integers from the documented algorithm (HOWTO.md:21-39, :71-97),
    sha256-pinned; the reference's file itself when /root/reference is present (this container)."""
    import hashlib as h
    if name == "ark":
        run, data, out = 1, b"poseidon-for-plonk", []
        for _ in range(960):
            data = h.sha512(data).digest()
            run = (int.from_bytes(data, "little") + run) % P
            out.append(run * R % P)
        want = ARK_SHA256
    else:
        out = [pow(i + j + 5, -1, P) * R % P for i in range(5) for j in range(5)]
        want = MDS_SHA256
    blob = b"".join(v.to_bytes(32, "little") for v in out)
    assert h.sha256(blob).hexdigest() == want
    path = os.path.join(REF_ASSETS, name + ".bin")
    if os.path.exists(path):
        assert blob == open(path, "rb").read()
    return blob


def test_round_constants_to_bytes_from_bytes_equal_blob(oracle):
    import numpy as np
    ark = blob_bytes("ark")
    for i in range(960):
        c = np.array(limbs_of(oracle.round_constant(i)), dtype=np.uint64)
        assert c.any()                                              # round_constants.rs:58
        chunk = ark[32 * i:32 * i + 32]
        assert oracle.to_bytes(c) == chunk                          # to_bytes(ROUND_CONSTANTS[i]) == file chunk i
        rc, back = oracle.from_bytes(chunk)                         # from_bytes(...) == ROUND_CONSTANTS[i]  (:61-62)
        assert rc == 0 and (back == c).all()
    mds = blob_bytes("mds")
    for i in range(5):
        for j in range(5):
            c = np.array(limbs_of(oracle.mds(i, j)), dtype=np.uint64)
            chunk = mds[32 * (5 * i + j):32 * (5 * i + j) + 32]
            assert oracle.to_bytes(c) == chunk
            rc, back = oracle.from_bytes(chunk)
            assert rc == 0 and (back == c).all()
