"""Specification-level oracle for the Hades252 permutation (TEST INFRASTRUCTURE ONLY).

This file is a pure-Python big-integer restatement of the reference's CPU path.  It is
only ever imported by ``tests/``, ``tools/`` (fixture/constant generation),
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg -- never by the product
path under ``hades252_amd/``.

Parity status: **unpinned by the reference** -- the reference holds no known-answer
vectors (its tests check determinism and Scalar<->Gadget agreement only,
``src/strategies/scalar.rs:62-74``, ``src/strategies/gadget.rs:207-271``) and its field
arithmetic lives in the un-vendored crate ``dusk-bls12_381 = "0.13"`` (``Cargo.toml:12``).
What pins this oracle instead:
  * the two constant blobs regenerated from the algorithm the reference documents
    (``assets/HOWTO.md:21-39`` ARK, ``:71-97`` MDS) match the reference's blobs by sha256
    (``ark.bin`` 78c42744..., ``mds.bin`` 131915cb...), see ``tools/gen_constants.py``;
  * three independent implementations (this one, ``oracle/hades_oracle.c`` with 4x64-bit
    Montgomery limbs, and the HIP kernels) agree bit for bit;
  * the anchor values recorded in SURVEY.md section 8(a) (``tests/golden/kat.json``).

What it follows, line by line:
  * round schedule           ``src/strategies.rs:140-157``  (4 full, 59 partial, 4 full)
  * full round               ``src/strategies.rs:107-119``  (ARK all, S-box all, MDS)
  * partial round            ``src/strategies.rs:79-93``    (ARK all, S-box LAST word, MDS)
  * constant cursor          ``src/strategies.rs:33-41,141`` (index = 5*round + word)
  * add_round_key            ``src/strategies/scalar.rs:23-30``
  * quintic_s_box            ``src/strategies/scalar.rs:32-34``
  * mul_matrix               ``src/strategies/scalar.rs:36-49`` (result[k] += MDS[k][j]*v[j])
  * constant loaders         ``src/round_constants.rs:29-48``, ``src/mds_matrix.rs:18-40``
                             (file chunk -> ``BlsScalar::from_raw`` = canonical integer)
  * parameters               ``src/lib.rs:20-27``
"""
from __future__ import annotations

import hashlib

# BLS12-381 scalar field modulus (src/strategies.rs:14, README.md:35)
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % P          # Montgomery radix of the upstream field crate (4 x u64 limbs)
R_INV = pow(R, -1, P)

WIDTH = 5                   # src/lib.rs:27
TOTAL_FULL_ROUNDS = 8       # src/lib.rs:21
PARTIAL_ROUNDS = 59         # src/lib.rs:25
N_CONSTANTS = 960           # src/round_constants.rs:18

ARK_SHA256 = "78c427449282315729eaa2e39e1937e0aa0b010c4c38bcbb1d57016011880485"
MDS_SHA256 = "131915cbeae1bde75422cce7fcf7feb9223a4dec370a937a2133c1f998ded0e7"


# --------------------------------------------------------------------------------------
# Constant blobs, regenerated from the documented algorithm (assets/HOWTO.md)
# --------------------------------------------------------------------------------------
def howto_ark_values() -> list[int]:
    """The 960 field elements of assets/HOWTO.md:21-39 (SHA-512 chain with running sum)."""
    out = []
    p_run = 1                               # BlsScalar::one()
    data = b"poseidon-for-plonk"
    for _ in range(N_CONSTANTS):
        data = hashlib.sha512(data).digest()
        c = (int.from_bytes(data, "little") + p_run) % P    # from_bytes_wide(&v) + p
        p_run = c
        out.append(c)
    return out


def howto_mds_values() -> list[list[int]]:
    """Cauchy matrix 1/(x_i + y_j), x_i = i, y_j = j + WIDTH (assets/HOWTO.md:71-97)."""
    return [[pow(i + j + WIDTH, -1, P) for j in range(WIDTH)] for i in range(WIDTH)]


def ark_blob() -> bytes:
    """Byte image of assets/ark.bin: `internal_repr()` (Montgomery limbs) of each element,
    little-endian (assets/HOWTO.md:41-48)."""
    return b"".join(((v * R) % P).to_bytes(32, "little") for v in howto_ark_values())


def mds_blob() -> bytes:
    """Byte image of assets/mds.bin (assets/HOWTO.md:100-108), row-major."""
    return b"".join(((v * R) % P).to_bytes(32, "little")
                    for row in howto_mds_values() for v in row)


def load_round_constants(blob: bytes) -> list[int]:
    """src/round_constants.rs:29-48: each 32-byte chunk, 4 LE u64 -> BlsScalar::from_raw,
    i.e. the chunk read as a canonical little-endian integer (all chunks are < P)."""
    assert len(blob) == 32 * N_CONSTANTS
    vals = [int.from_bytes(blob[i:i + 32], "little") for i in range(0, len(blob), 32)]
    assert all(0 < v < P for v in vals)     # src/round_constants.rs:55-65
    return vals


def load_mds(blob: bytes) -> list[list[int]]:
    """src/mds_matrix.rs:18-40: row-major mds[i][j] from consecutive 32-byte chunks."""
    assert len(blob) == 32 * WIDTH * WIDTH
    flat = [int.from_bytes(blob[k:k + 32], "little") for k in range(0, len(blob), 32)]
    assert all(v < P for v in flat)
    return [flat[i * WIDTH:(i + 1) * WIDTH] for i in range(WIDTH)]


_CACHE: dict[str, object] = {}

# How the blob bytes become constants.
#   "from_raw"  (default, what the reference's code does): src/round_constants.rs:41 and
#               src/mds_matrix.rs:33 hand each chunk to BlsScalar::from_raw, i.e. the chunk is a
#               CANONICAL integer; since the generator wrote Montgomery limbs (HOWTO.md:45,104) the
#               effective constants are howto * R.
#   "howto"     the other reading (chunk = in-memory Montgomery limbs, constants = the HOWTO values).
#               NOT what the code does; kept so that one run of the real crate can settle the
#               question: perm([1;5])[0] is 0x71a5b804... under "from_raw" and 0x5221c7bb... under
#               "howto" (SURVEY.md section 8(a) "Disambiguator").
LOADER = "from_raw"


def set_loader(mode: str) -> None:
    global LOADER
    assert mode in ("from_raw", "howto")
    LOADER = mode
    _CACHE.clear()


def round_constants() -> list[int]:
    if "ark" not in _CACHE:
        blob = ark_blob()
        assert hashlib.sha256(blob).hexdigest() == ARK_SHA256, "regenerated ark.bin differs"
        vals = load_round_constants(blob)
        if LOADER == "howto":
            vals = [v * R_INV % P for v in vals]
            assert vals == howto_ark_values()
        _CACHE["ark"] = vals
    return _CACHE["ark"]            # type: ignore[return-value]


def mds_matrix() -> list[list[int]]:
    if "mds" not in _CACHE:
        blob = mds_blob()
        assert hashlib.sha256(blob).hexdigest() == MDS_SHA256, "regenerated mds.bin differs"
        m = load_mds(blob)
        if LOADER == "howto":
            m = [[v * R_INV % P for v in row] for row in m]
            assert m == howto_mds_values()
        _CACHE["mds"] = m
    return _CACHE["mds"]            # type: ignore[return-value]


# --------------------------------------------------------------------------------------
# The permutation on canonical integers
# --------------------------------------------------------------------------------------
def add_round_key(words: list[int], consts, cursor: int) -> int:
    """src/strategies/scalar.rs:23-30."""
    for w in range(len(words)):
        words[w] = (words[w] + consts[cursor]) % P
        cursor += 1
    return cursor


def quintic_s_box(v: int) -> int:
    """src/strategies/scalar.rs:32-34: value.square().square() * value."""
    v2 = v * v % P
    v4 = v2 * v2 % P
    return v4 * v % P


def mul_matrix(words: list[int], mds) -> None:
    """src/strategies/scalar.rs:36-49."""
    result = [0] * WIDTH
    for j in range(WIDTH):
        for k in range(WIDTH):
            result[k] = (result[k] + mds[k][j] * words[j]) % P
    words[:] = result


def perm(words: list[int], trace: list | None = None) -> list[int]:
    """src/strategies.rs:140-157 on canonical integers.  `words` must hold exactly WIDTH
    values in [0, P) (a different length panics in the reference, scalar.rs:48)."""
    if len(words) != WIDTH:
        raise ValueError("Hades252 state must have exactly WIDTH words")
    ark, mds = round_constants(), mds_matrix()
    st = list(words)
    cur = 0
    for r in range(TOTAL_FULL_ROUNDS + PARTIAL_ROUNDS):
        full = r < TOTAL_FULL_ROUNDS // 2 or r >= TOTAL_FULL_ROUNDS // 2 + PARTIAL_ROUNDS
        cur = add_round_key(st, ark, cur)
        if full:
            st = [quintic_s_box(v) for v in st]
        else:
            st[WIDTH - 1] = quintic_s_box(st[WIDTH - 1])
        mul_matrix(st, mds)
        if trace is not None:
            trace.append(list(st))
    return st


def perm_gadget(words: list[int], wires: list | None = None) -> list[int]:
    """The witness values ``GadgetStrategy`` assigns when the trait's provided ``perm``
    (src/strategies.rs:140-157) drives ITS overrides -- a different schedule of the same permutation:
      * add_round_key (src/strategies/gadget.rs:41-57) adds constants in the FIRST round only
        (``self.count == 0``); every later round key is appended to the previous linear layer;
      * quintic_s_box (:59-69): three multiplication gates v2 = v*v, v4 = v2*v2, v5 = v4*v;
      * mul_matrix (:71-133): ``count += 1``; per output row two addition gates,
        r1 = M[j][0] v0 + M[j][1] v1 + M[j][2] v2 and r2 = M[j][3] v3 + M[j][4] v4 + r1 + c with
        c = next round constant while ``count < rounds()``, else 0.
    The reference's own tests assert that this equals ``ScalarStrategy::perm`` on the same input
    (``preimage`` / ``preimage_constant``, gadget.rs:166-175, :207-244); tests/test_oracle.py repeats that
    check between this function and ``perm`` -- the one cross-check of the round/constant schedule the
    reference itself holds.  ``wires`` (a list) receives every gate output in gate order (972 values)."""
    if len(words) != WIDTH:
        raise ValueError("Hades252 state must have exactly WIDTH words")
    ark, mds = round_constants(), mds_matrix()
    rounds = TOTAL_FULL_ROUNDS + PARTIAL_ROUNDS
    st = list(words)
    cur = 0
    count = 0

    def emit(v):
        if wires is not None:
            wires.append(v)
        return v

    def sbox(v):
        v2 = emit(v * v % P)
        v4 = emit(v2 * v2 % P)
        return emit(v4 * v % P)

    for r in range(rounds):
        full = r < TOTAL_FULL_ROUNDS // 2 or r >= TOTAL_FULL_ROUNDS // 2 + PARTIAL_ROUNDS
        if count == 0:
            for w in range(WIDTH):
                st[w] = emit((st[w] + ark[cur]) % P)
                cur += 1
        if full:
            st = [sbox(v) for v in st]
        else:
            st[WIDTH - 1] = sbox(st[WIDTH - 1])
        count += 1
        result = [0] * WIDTH
        for j in range(WIDTH):
            c = 0
            if count < rounds:
                c = ark[cur]
                cur += 1
            r1 = emit((mds[j][0] * st[0] + mds[j][1] * st[1] + mds[j][2] * st[2]) % P)
            result[j] = emit((mds[j][3] * st[3] + mds[j][4] * st[4] + r1 + c) % P)
        st = result
    assert cur == WIDTH * rounds
    return st


# --------------------------------------------------------------------------------------
# Memory-format helpers (BlsScalar = 4 x u64 LE limbs of value*R mod P)
# --------------------------------------------------------------------------------------
def to_mont(v: int) -> int:
    return v * R % P


def from_mont(m: int) -> int:
    return m * R_INV % P


def perm_mont_limbs(limbs: list[int]) -> list[int]:
    """Permute one state given as 20 u64 Montgomery limbs; returns 20 u64 limbs."""
    assert len(limbs) == 4 * WIDTH
    vals = []
    for w in range(WIDTH):
        m = sum(limbs[4 * w + k] << (64 * k) for k in range(4))
        assert m < P, "non-canonical Montgomery limbs"
        vals.append(from_mont(m))
    out = perm(vals)
    res = []
    for v in out:
        m = to_mont(v)
        res.extend((m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4))
    return res


# --------------------------------------------------------------------------------------
# Synthetic input generators; values are Montgomery-limb integers
# --------------------------------------------------------------------------------------
GEN_SEED = 0x4861646573323532
_M64 = (1 << 64) - 1


def splitmix_limb(seed: int, idx: int) -> int:
    z = (seed + ((idx + 1) * 0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def gen_b_element(e: int, seed: int = GEN_SEED) -> list[int]:
    """Generator B: element e -> 4 u64 Montgomery limbs, top limb masked to 62 bits."""
    limbs = [splitmix_limb(seed, 4 * e + k) for k in range(4)]
    limbs[3] &= 0x3FFFFFFFFFFFFFFF
    return limbs


def gen_a_element(e: int) -> list[int]:
    """Generator A: element e has VALUE e, stored as Montgomery limbs."""
    m = to_mont(e % P)
    return [(m >> (64 * k)) & _M64 for k in range(4)]


# --------------------------------------------------------------------------------------
# Merkle (arity 4) on top of perm: node = perm([tag, c0, c1, c2, c3])[out_idx]
# --------------------------------------------------------------------------------------
def merkle4_node(children: list[int], tag: int = 15, out_idx: int = 1) -> int:
    return perm([tag % P] + list(children))[out_idx]


def merkle4_root(leaves: list[int], tag: int = 15, out_idx: int = 1) -> int:
    level = list(leaves)
    assert len(level) >= 4
    while len(level) > 1:
        assert len(level) % 4 == 0
        level = [merkle4_node(level[i:i + 4], tag, out_idx) for i in range(0, len(level), 4)]
    return level[0]


# --------------------------------------------------------------------------------------
# Fixed-length sponge (convention parameters: see include/hades252.h); canonical integers
# --------------------------------------------------------------------------------------
def sponge_hash(msg: list[int], capacity: int, pad_mode: int = 1) -> int:
    padded = list(msg) + ([1] if pad_mode == 1 else [])
    if not padded:
        padded = [0]
    while len(padded) % 4:
        padded.append(0)
    st = [capacity % P, 0, 0, 0, 0]
    for t in range(0, len(padded), 4):
        for k in range(4):
            st[1 + k] = (st[1 + k] + padded[t + k]) % P
        st = perm(st)
    return st[1]
