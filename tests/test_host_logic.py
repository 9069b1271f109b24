"""CPU tier: host-side mirror logic and the derived device tables."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hades_spec as S  # noqa: E402
from hades252_amd import _derive as D  # noqa: E402
from hades252_amd import strategy as H  # noqa: E402


def test_parameters_match_reference():
    # src/lib.rs:20-27, src/strategies.rs:160-162
    import hades252_amd
    assert (hades252_amd.WIDTH, hades252_amd.TOTAL_FULL_ROUNDS, hades252_amd.PARTIAL_ROUNDS) == (5, 8, 59)
    assert H.Strategy.rounds() == 67


def test_cursor_semantics():
    it = H.RoundConstantsIter()
    assert [H.Strategy.next_c(it) for _ in range(7)] == list(range(7))
    it = H.RoundConstantsIter(959)
    assert H.Strategy.next_c(it) == 959
    with pytest.raises(RuntimeError, match="Hades252 out of ARK constants"):   # src/strategies.rs:40
        H.Strategy.next_c(it)


def test_derived_literal_tables(oracle):
    D.check_blobs()
    ark, mds = D.literal_tables()
    assert len(ark) == 960 and len(mds) == 25          # all of src/round_constants.rs:18
    for i in (0, 1, 5, 170, 334, 335, 700, 959):
        assert ark[i] == oracle.round_constant(i)
    for i in range(5):
        for j in range(5):
            assert mds[5 * i + j] == oracle.mds(i, j)
    # independent derivation agrees with the spec oracle's values
    assert D.ark_values() == S.round_constants()
    assert D.mds_values() == S.mds_matrix()


def test_committed_inc_is_current(tmp_path):
    p = tmp_path / "x.inc"
    D.emit_inc(str(p))
    assert p.read_text() == open(os.path.join(ROOT, "hades252_amd", "csrc", "hades_constants.inc")).read()


def test_host_buffer_validation():
    s = H.ScalarStrategy.__new__(H.ScalarStrategy)   # no library needed for argument checks
    s.kernel = 0
    with pytest.raises(TypeError):
        s.perm(np.zeros(20, dtype=np.int32))
    with pytest.raises(ValueError):
        s.perm(np.zeros(19, dtype=np.uint64))      # not a whole state: reference panics (scalar.rs:48)


def test_merkle_subtree_split():
    from hades252_amd import merkle
    # 2^24 leaves over 8 GPUs = two complete 10-level sub-trees (2^20 leaves) per GPU
    assert merkle.subtree_split(1 << 24, 8) == (1 << 20, 2)
    assert merkle.subtree_split(1 << 24, 1) == (1 << 24, 1)
    assert merkle.subtree_split(1 << 24, 4) == (1 << 22, 1)
    assert merkle.subtree_split(1 << 24, 2) == (1 << 22, 2)
    assert merkle.subtree_split(16, 4) == (4, 1)
    for bad in ((1 << 23, 2), (1 << 24, 3), (4, 4)):
        with pytest.raises(ValueError):
            merkle.subtree_split(*bad)


def test_merkle_shape_helpers():
    """n_l = ceil(n_{l-1} / arity): the level sizes the builder, the openings and the oracle wrapper all use."""
    assert H.merkle_level_sizes(16, 4) == [4, 1] and H.merkle_level_sizes(8, 2) == [4, 2, 1]
    assert H.merkle_level_sizes(12, 4) == [3, 1] and H.merkle_level_sizes(16, 3) == [6, 2, 1]
    assert H.merkle_level_sizes(2, 4) == [1] and H.merkle_level_sizes(5, 2) == [3, 2, 1]
    assert sum(H.merkle_level_sizes(4 ** 7, 4)) == (4 ** 7 - 1) // 3
    assert set(H.SPONGE_PRESETS) == {"sponge/pad10", "merkle/arity4"}
