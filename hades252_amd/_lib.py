"""ctypes binding of libhades252.so -- the same C ABI a Rust `extern "C"` block would bind
(include/hades252.h).  There is no fallback: if the library is missing this module raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_size_t, c_uint64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# HADES252_LIB overrides the path (development A/B builds only)
LIB_PATH = os.environ.get("HADES252_LIB") or os.path.join(_HERE, "csrc", "libhades252.so")

OK = 0
ERR_INVALID_ARG = -1
ERR_HIP = -2
ERR_NOT_CANONICAL = -3
ERR_NO_DEVICE = -4
ERR_SCRATCH = -5
ERR_OUT_OF_CONSTANTS = -6

KERNEL_DEFAULT = 0
KERNEL_LITERAL = 1
KERNEL_FAST = 2
KERNEL_COOP = 3
KERNEL_LANES = 4
KERNEL_ROWS = 5
MULTI_VIRTUAL = 1

# every symbol include/hades252.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "hades252_rounds": (c_int, []),
    "hades252_device_count": (c_int, []),
    "hades252_strerror": (c_char_p, [c_int]),
    "hades252_last_hip_error": (c_int, []),
    "hades252_version": (c_char_p, []),
    "hades252_kernel_for": (c_int, [c_size_t]),
    "hades252_chain_form_for": (c_int, [c_size_t]),
    "hades252_kernel_name": (c_char_p, [c_int, c_size_t]),
    "hades252_trim": (c_int, []),
    "hades252_pool_bytes": (c_size_t, []),
    "hades252_stage_threads": (c_int, [c_int]),
    "hades252_fault_inject": (c_int, [c_char_p]),
    "hades252_perm_batch": (c_int, [c_void_p, c_size_t]),
    "hades252_perm_batch_bytes": (c_int, [c_void_p, c_size_t]),
    "hades252_perm_batch_dev": (c_int, [c_void_p, c_size_t, c_void_p]),
    "hades252_perm_batch_dev_ex": (c_int, [c_void_p, c_size_t, c_void_p, c_int]),
    "hades252_perm_batch_multi": (c_int, [c_void_p, c_size_t, c_int]),
    "hades252_perm_batch_multi_ex": (c_int, [c_void_p, c_size_t, c_int, ctypes.c_uint]),
    "hades252_host_alloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "hades252_host_free": (c_int, [c_void_p]),
    "hades252_host_register": (c_int, [c_void_p, c_size_t]),
    "hades252_host_unregister": (c_int, [c_void_p]),
    "hades252_host_is_pinned": (c_int, [c_void_p, c_size_t]),
    "hades252_dev_alloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "hades252_dev_free": (c_int, [c_void_p]),
    "hades252_dev_upload": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "hades252_dev_download": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "hades252_stream_create": (c_int, [POINTER(c_void_p)]),
    "hades252_stream_destroy": (c_int, [c_void_p]),
    "hades252_stream_sync": (c_int, [c_void_p]),
    "hades252_merkle_root": (c_int, [c_void_p, c_size_t, c_int, POINTER(c_uint64), c_int, c_void_p, c_void_p]),
    "hades252_sponge_hash": (c_int, [c_void_p, c_size_t, c_size_t, POINTER(c_uint64), c_int, c_void_p]),
    "hades252_merkle_root_multi": (c_int, [c_void_p, c_size_t, c_int, POINTER(c_uint64), c_int, c_int, ctypes.c_uint, c_void_p]),
    "hades252_sponge_hash_var": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, POINTER(c_uint64), c_int, c_void_p,
                                         POINTER(c_size_t)]),
    "hades252_perm_trace_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "hades252_perm_trace_dev_ex": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int]),
    "hades252_perm_trace_scaled_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "hades252_perm_trace_scale_table": (c_int, [c_void_p, c_void_p]),
    "hades252_witness_wires": (c_int, []),
    "hades252_warm_up": (c_int, [c_size_t]),
    "hades252_perm_witness_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "hades252_add_round_key_dev": (c_int, [c_void_p, c_size_t, c_int, c_void_p]),
    "hades252_add_round_key_at_dev": (c_int, [c_void_p, c_size_t, c_int, c_void_p]),
    "hades252_apply_full_round_at_dev": (c_int, [c_void_p, c_size_t, c_int, c_void_p]),
    "hades252_apply_partial_round_at_dev": (c_int, [c_void_p, c_size_t, c_int, c_void_p]),
    "hades252_fr_op_dev": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hades252_quintic_s_box_dev": (c_int, [c_void_p, c_size_t, c_void_p]),
    "hades252_mul_matrix_dev": (c_int, [c_void_p, c_size_t, c_void_p]),
    "hades252_apply_full_round_dev": (c_int, [c_void_p, c_size_t, c_int, c_void_p]),
    "hades252_apply_partial_round_dev": (c_int, [c_void_p, c_size_t, c_int, c_void_p]),
    "hades252_from_bytes_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "hades252_to_bytes_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "hades252_merkle_level_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_int, POINTER(c_uint64), c_int, c_void_p]),
    "hades252_merkle_depth": (c_int, [c_size_t, c_int]),
    "hades252_merkle_level_pad_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_int, POINTER(c_uint64), c_int, c_void_p,
                                              c_void_p]),
    "hades252_merkle_root_pad_dev": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_size_t, POINTER(c_uint64), c_int,
                                             c_void_p, c_void_p, c_void_p]),
    "hades252_merkle_build_pad_dev": (c_int, [c_void_p, c_size_t, c_int, POINTER(c_uint64), c_int, c_void_p, c_void_p,
                                              c_void_p]),
    "hades252_merkle_empty_digests_dev": (c_int, [c_int, c_int, POINTER(c_uint64), POINTER(c_uint64), c_int, c_void_p,
                                                  c_void_p]),
    "hades252_merkle_open_pad_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_void_p,
                                             c_void_p]),
    "hades252_merkle_verify_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, POINTER(c_uint64), c_int,
                                           c_void_p, c_void_p]),
    "hades252_merkle_update_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_int, POINTER(c_uint64), c_int, c_void_p,
                                           c_void_p, c_size_t, c_void_p]),
    "hades252_merkle_forest_scratch_bytes": (c_size_t, [c_size_t, c_size_t, c_int]),
    "hades252_merkle_forest_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_int, c_void_p, c_size_t, POINTER(c_uint64),
                                           c_int, c_void_p, c_void_p]),
    "hades252_merkle_scratch_bytes": (c_size_t, [c_size_t, c_int]),
    "hades252_merkle_root_dev": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_size_t, POINTER(c_uint64), c_int,
                                         c_void_p, c_void_p]),
    "hades252_merkle_tree_bytes": (c_size_t, [c_size_t, c_int]),
    "hades252_merkle_build_dev": (c_int, [c_void_p, c_size_t, c_int, POINTER(c_uint64), c_int, c_void_p, c_void_p]),
    "hades252_merkle_open_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "hades252_merkle4_level_dev": (c_int, [c_void_p, c_void_p, c_size_t, POINTER(c_uint64), c_int, c_void_p]),
    "hades252_merkle4_scratch_bytes": (c_size_t, [c_size_t]),
    "hades252_merkle4_root_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, POINTER(c_uint64), c_int,
                                          c_void_p, c_void_p]),
    "hades252_sponge_hash_dev": (c_int, [c_void_p, c_size_t, c_size_t, POINTER(c_uint64), c_int, c_void_p, c_void_p]),
    "hades252_sponge_hash_var_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, POINTER(c_uint64), c_int,
                                             c_void_p, c_void_p, c_void_p]),
    "hades252_sponge_sort_scratch_bytes": (c_size_t, [c_size_t]),
    "hades252_sponge_hash_var_ex_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, POINTER(c_uint64), c_int,
                                                c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hades252_sponge_init_dev": (c_int, [c_void_p, c_size_t, POINTER(c_uint64), c_void_p]),
    "hades252_sponge_absorb_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "hades252_sponge_squeeze_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "hades252_gen_b_dev": (c_int, [c_void_p, c_uint64, c_size_t, c_uint64, c_void_p]),
    "hades252_gen_a_dev": (c_int, [c_void_p, c_uint64, c_size_t, c_void_p]),
    "hades252_digest_dev": (c_int, [c_void_p, c_uint64, c_size_t, c_void_p, c_void_p]),
}

_lib = None


class HadesError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        msg = lib().hades252_strerror(code).decode()
        if code == ERR_HIP:
            msg += " [hipError_t %d]" % lib().hades252_last_hip_error()
        super().__init__("%s: %s (%d)" % (where, msg, code))


def lib() -> ctypes.CDLL:
    """Load libhades252.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libhades252.so is missing (%s). Build it with `python -m hades252_amd.build`; "
                "hades252_amd has no CPU fallback." % LIB_PATH)
        # torch bundles its own HIP runtime; it must be the one this process uses (device pointers
        # and streams are torch's), so make sure it is loaded before libhades252's dependency on
        # libamdhip64 is resolved.  A C/Rust host links the system runtime directly.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)           # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code: int, where: str) -> None:
    if code != OK:
        raise HadesError(code, where)
