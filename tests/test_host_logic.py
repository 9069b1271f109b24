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


_STAGE_CHILD = r"""
import ctypes, os, sys
cpus = sorted(os.sched_getaffinity(0))[:int(sys.argv[1])]
os.sched_setaffinity(0, cpus)
lib = ctypes.CDLL(sys.argv[2])
lib.hades252_stage_threads.restype = ctypes.c_int
lib.hades252_stage_threads.argtypes = [ctypes.c_int]
print(len(cpus), *[lib.hades252_stage_threads(w) for w in (1, 2, 4, 8, 64, 0, -3)])
"""


def test_staging_threads_never_outnumber_the_usable_cpus():
    """an 8-GPU hades252_perm_batch_multi call used 3 + 3 helper threads per worker whatever the
    process may run on.  hades252_stage_threads(workers) = min(HADES252_STAGE_THREADS, cpus / (2 workers)), >= 1 -- pure
    host arithmetic (no HIP call), checked under restricted affinity masks in child processes."""
    import subprocess
    from hades252_amd import build
    lib = build.build(verbose=False)
    have = len(os.sched_getaffinity(0))
    for n_cpus in sorted({1, 2, 4, 6, 8, have}):
        if n_cpus > have:
            continue
        for env_threads in (None, "6", "1"):
            env = dict(os.environ)
            env.pop("HADES252_STAGE_THREADS", None)
            if env_threads:
                env["HADES252_STAGE_THREADS"] = env_threads
            r = subprocess.run([sys.executable, "-c", _STAGE_CHILD, str(n_cpus), lib], capture_output=True, text=True,
                               env=env, timeout=120)
            assert r.returncode == 0, r.stderr[-800:]
            got = [int(x) for x in r.stdout.split()]
            cpus, per_workers = got[0], got[1:]
            configured = int(env_threads or 3)
            for workers, t in zip((1, 2, 4, 8, 64, 1, 1), per_workers):         # (0 and negative mean one caller)
                assert t == max(1, min(configured, cpus // (2 * workers))), (cpus, env_threads, workers, t)
                assert t == 1 or workers * 2 * t <= cpus


def test_trace_scale_table_of_the_library_is_the_derived_one():
    """hades252_perm_trace_scale_table copies host tables (no HIP call): they are D.trace_scaled_tables(), i.e.
    mul[r] = R^2 / s_after(r), add[r][w] = (deferred constant) * R -- the limb model checks the algebra against the oracle
    (tests/test_fast_model.py::test_scaled_trace_model_matches_spec_oracle)."""
    from oracle_lib import int_of
    mul, add = H.trace_scale_table()
    want_mul, want_add = D.trace_scaled_tables()
    assert mul.shape == (67, 4) and add.shape == (67, 5, 4)
    assert [int_of(m) for m in mul] == want_mul
    assert [[int_of(a) for a in row] for row in add] == want_add
    assert all(v < D.P for v in want_mul) and len(set(want_mul)) == 67
