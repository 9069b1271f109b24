"""CPU tier: the C-ABI library builds, loads and exports every symbol include/hades252.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hades252.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hades252_\w+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("hades252_perm_batch", "hades252_perm_batch_dev", "hades252_perm_batch_bytes",
                 "hades252_merkle4_level_dev", "hades252_device_count", "hades252_strerror"):
        assert must in syms


def test_library_exports_every_declared_symbol(hades_lib):
    from hades252_amd import _lib
    syms = declared_symbols()
    assert set(syms) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), "missing export " + s


def test_meta_calls_without_gpu(hades_lib):
    assert hades_lib.hades252_rounds() == 67
    assert hades_lib.hades252_strerror(0) == b"ok"
    assert b"canonical" in hades_lib.hades252_strerror(-3)
    assert hades_lib.hades252_version().startswith(b"hades252-amd")
    # empty batches are a no-op success, like perm on zero states would be
    assert hades_lib.hades252_perm_batch(None, 0) == 0
    assert hades_lib.hades252_perm_batch_dev(None, 0, None) == 0
    # NULL with n > 0 is an argument error, not a crash
    assert hades_lib.hades252_perm_batch_dev(None, 4, None) == -1
    assert hades_lib.hades252_merkle4_scratch_bytes(16) == 4 * 32 + 32
    assert hades_lib.hades252_merkle4_scratch_bytes(4) == 0          # one level: no scratch needed (a valid shape:
    assert hades_lib.hades252_merkle_depth(4, 4) == 1                # depth tells valid from invalid)
    assert hades_lib.hades252_merkle_scratch_bytes(8, 2) == 4 * 32 + 2 * 32
    assert hades_lib.hades252_merkle_tree_bytes(16, 4) == 5 * 32 and hades_lib.hades252_merkle_tree_bytes(8, 2) == 7 * 32
    # any leaf count >= 2, arity 2..4: n_l = ceil(n_{l-1} / arity)
    assert hades_lib.hades252_merkle_depth(12, 4) == 2 and hades_lib.hades252_merkle_tree_bytes(12, 4) == (3 + 1) * 32
    assert hades_lib.hades252_merkle_depth(16, 3) == 3 and hades_lib.hades252_merkle_tree_bytes(16, 3) == (6 + 2 + 1) * 32
    assert hades_lib.hades252_merkle4_scratch_bytes(8) == (2 + 1) * 32
    assert hades_lib.hades252_merkle_depth(3 ** 9, 3) == 9 and hades_lib.hades252_merkle_depth(4 ** 7 * 3, 4) == 8
    for n, a in ((1, 4), (0, 2), (16, 5), (16, 1), (16, 0)):
        assert hades_lib.hades252_merkle_depth(n, a) == -1 and hades_lib.hades252_merkle_tree_bytes(n, a) == 0
    assert hades_lib.hades252_merkle_forest_scratch_bytes(10, 64, 4) == (160 + 40) * 32
    assert hades_lib.hades252_merkle_forest_scratch_bytes(10, 48, 4) == 0        # trees of a forest: powers of the arity


def test_no_cpu_fallback_in_product():
    """The product package must not reference the oracle."""
    pkg = os.path.join(ROOT, "hades252_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".hpp", ".rs")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "hades_oracle" not in text and "hades_spec" not in text, f
                assert "oracle_lib" not in text, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from hades252_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
