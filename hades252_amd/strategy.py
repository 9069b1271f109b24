"""Host-side mirror of the reference's operator interface for the hot path.

The reference exposes ``pub trait Strategy<T>`` (src/strategies.rs:31-163) and the implementor
``ScalarStrategy`` (src/strategies/scalar.rs:11-50).  This module keeps the same names, argument
meaning and error behaviour, batched: wherever the reference takes ``&mut [BlsScalar]`` of
exactly ``WIDTH`` words, these methods take a buffer holding any whole number of states
(AoS, 160 bytes each, in-memory ``BlsScalar`` = 4 x u64 LE Montgomery limbs) and apply the
operation to every state, in place, on the GPU through the C ABI of ``include/hades252.h``.

Buffers
  * ``torch.Tensor`` on a ``cuda`` device (any integer dtype, contiguous): device path,
    asynchronous on the current torch stream -- zero-copy.
  * ``numpy.ndarray`` of dtype uint64 (C-contiguous): host path, synchronous, the library moves
    the data (``hades252_perm_batch``); only ``perm`` supports it.
    RATE WARNING for this path: in a process that imported ``torch`` first, the library runs on the
    HIP runtime PyTorch bundles (ROCm 7.0.2), which serialises the copy-in and copy-out streams of
    the chunk pipeline -- 2^22 states take 29.6 ms (22.7 GB/s each way: the rate the system
    runtime gives with ``HSA_ENABLE_SDMA=0``) against 15.1 ms on the system runtime (ROCm 7.2) a
    native / Rust caller links.  No environment setting repairs it (``HSA_ENABLE_SDMA``,
    ``GPU_MAX_HW_QUEUES``, ``HSA_ENABLE_INTERRUPT``, ``HIP_FORCE_DEV_KERNARG``: no change;
    ``AMD_DIRECT_DISPATCH=0``: 24 ms; chunk sizes 2^14 .. 2^18 states: 27-33 ms), and the library cannot: the cause is inside that runtime
    (profiles/r5/host_path_torch_probe.txt).  A host-only Python caller does not need torch -- use
    this module WITHOUT importing torch (the library then binds the system runtime: 15.1 ms).
    Binding the system runtime first and importing torch afterwards is also full speed, but puts
    two HIP runtimes in one process: torch tensors and streams must then never be passed to the
    device path.

There is no CPU implementation behind these methods: if ``libhades252.so`` is not built or no
GPU is usable they raise.
"""
from __future__ import annotations

import ctypes
from typing import Iterator

import numpy as np

from . import _lib
from ._lib import check

WIDTH = 5
TOTAL_FULL_ROUNDS = 8
PARTIAL_ROUNDS = 59
STATE_BYTES = WIDTH * 32
N_ROUND_CONSTANTS = 960          # src/round_constants.rs:18 (335 are ever consumed)


class RoundConstantsIter:
    """``ROUND_CONSTANTS.iter()`` (src/strategies.rs:141): a cursor into the constant table.
    The table itself lives in device memory; the cursor is an index."""

    def __init__(self, pos: int = 0):
        self.pos = pos

    def __iter__(self) -> Iterator[int]:
        return self

    def __next__(self) -> int:
        if self.pos >= N_ROUND_CONSTANTS:
            raise StopIteration
        self.pos += 1
        return self.pos - 1


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _dev_buffer(x, unit_bytes: int, what: str):
    """-> (data_ptr, n_units, torch device) of a contiguous CUDA tensor."""
    import torch
    if not isinstance(x, torch.Tensor) or x.device.type != "cuda":
        raise TypeError("%s: expected a CUDA torch.Tensor" % what)
    if not x.is_contiguous():
        raise ValueError("%s: tensor must be contiguous" % what)
    nbytes = x.numel() * x.element_size()
    # same failure the reference has for a slice whose length is not WIDTH
    # (copy_from_slice length panic, src/strategies/scalar.rs:48)
    if nbytes % unit_bytes != 0:
        raise ValueError("%s: buffer of %d bytes is not a whole number of %d-byte units"
                         % (what, nbytes, unit_bytes))
    if x.data_ptr() % 16 != 0:
        raise ValueError("%s: buffer must be 16-byte aligned" % what)
    return x.data_ptr(), nbytes // unit_bytes, x.device


def _stream_ptr(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream


class Strategy:
    """Mirror of ``pub trait Strategy<T: Clone + Copy>`` (src/strategies.rs:31)."""

    @staticmethod
    def next_c(constants: RoundConstantsIter) -> int:
        """src/strategies.rs:33-41 -- returns the index of the constant consumed."""
        try:
            return next(constants)
        except StopIteration:
            raise RuntimeError("Hades252 out of ARK constants") from None

    # required methods of the trait (src/strategies.rs:50-65)
    def add_round_key(self, constants: RoundConstantsIter, words) -> None:
        raise NotImplementedError

    def quintic_s_box(self, value) -> None:
        raise NotImplementedError

    def mul_matrix(self, constants: RoundConstantsIter, values) -> None:
        raise NotImplementedError

    # provided methods (src/strategies.rs:79-157)
    def apply_partial_round(self, constants: RoundConstantsIter, words) -> None:
        raise NotImplementedError

    def apply_full_round(self, constants: RoundConstantsIter, words) -> None:
        raise NotImplementedError

    def perm(self, data) -> None:
        raise NotImplementedError

    @staticmethod
    def rounds() -> int:
        """src/strategies.rs:160-162."""
        return TOTAL_FULL_ROUNDS + PARTIAL_ROUNDS


class ScalarStrategy(Strategy):
    """Batched GPU ``ScalarStrategy`` (src/strategies/scalar.rs:11-50).  Stateless, like the
    reference's zero-sized struct; ``kernel`` selects one of the two bit-identical kernels."""

    def __init__(self, kernel: int = _lib.KERNEL_DEFAULT):
        self.kernel = kernel
        _lib.lib()                       # fail loudly at construction if the library is missing

    @classmethod
    def new(cls) -> "ScalarStrategy":
        """src/strategies/scalar.rs:17-19."""
        return cls()

    @staticmethod
    def _cursor_of(constants: RoundConstantsIter) -> int:
        # any cursor is legal (the trait methods take the iterator wherever it stands,
        # src/strategies.rs:33-41); running past the 960 constants is the reference's panic
        if constants.pos + WIDTH > N_ROUND_CONSTANTS:
            raise RuntimeError("Hades252 out of ARK constants")
        return constants.pos

    def add_round_key(self, constants: RoundConstantsIter, words) -> None:
        """src/strategies/scalar.rs:23-30: word w of every state += next_c()."""
        import torch
        ptr, n, dev = _dev_buffer(words, STATE_BYTES, "add_round_key")
        cur = self._cursor_of(constants)
        with torch.cuda.device(dev):
            check(_lib.lib().hades252_add_round_key_at_dev(ptr, n, cur, _stream_ptr(dev)), "add_round_key")
        for _ in range(WIDTH):
            self.next_c(constants)

    def quintic_s_box(self, value) -> None:
        """src/strategies/scalar.rs:32-34 on every 32-byte scalar of ``value``."""
        import torch
        ptr, n, dev = _dev_buffer(value, 32, "quintic_s_box")
        with torch.cuda.device(dev):
            check(_lib.lib().hades252_quintic_s_box_dev(ptr, n, _stream_ptr(dev)), "quintic_s_box")

    def mul_matrix(self, constants: RoundConstantsIter, values) -> None:
        """src/strategies/scalar.rs:36-49 (``_constants`` is unused there too)."""
        import torch
        ptr, n, dev = _dev_buffer(values, STATE_BYTES, "mul_matrix")
        with torch.cuda.device(dev):
            check(_lib.lib().hades252_mul_matrix_dev(ptr, n, _stream_ptr(dev)), "mul_matrix")

    def apply_partial_round(self, constants: RoundConstantsIter, words) -> None:
        """src/strategies.rs:79-93, fused in one launch."""
        import torch
        ptr, n, dev = _dev_buffer(words, STATE_BYTES, "apply_partial_round")
        cur = self._cursor_of(constants)
        with torch.cuda.device(dev):
            check(_lib.lib().hades252_apply_partial_round_at_dev(ptr, n, cur, _stream_ptr(dev)), "apply_partial_round")
        for _ in range(WIDTH):
            self.next_c(constants)

    def apply_full_round(self, constants: RoundConstantsIter, words) -> None:
        """src/strategies.rs:107-119, fused in one launch."""
        import torch
        ptr, n, dev = _dev_buffer(words, STATE_BYTES, "apply_full_round")
        cur = self._cursor_of(constants)
        with torch.cuda.device(dev):
            check(_lib.lib().hades252_apply_full_round_at_dev(ptr, n, cur, _stream_ptr(dev)), "apply_full_round")
        for _ in range(WIDTH):
            self.next_c(constants)

    def perm(self, data) -> None:
        """src/strategies.rs:140-157 on every state of ``data``, in place."""
        if isinstance(data, np.ndarray):
            if data.dtype != np.uint64 or not data.flags["C_CONTIGUOUS"]:
                raise TypeError("perm: host buffers must be C-contiguous numpy uint64")
            if data.size % (WIDTH * 4) != 0:
                raise ValueError("perm: %d limbs is not a whole number of %d-word states" % (data.size, WIDTH))
            check(_lib.lib().hades252_perm_batch(data.ctypes.data_as(ctypes.c_void_p), data.size // (WIDTH * 4)),
                  "perm")
            return
        import torch
        ptr, n, dev = _dev_buffer(data, STATE_BYTES, "perm")
        with torch.cuda.device(dev):
            check(_lib.lib().hades252_perm_batch_dev_ex(ptr, n, _stream_ptr(dev), self.kernel), "perm")

    def perm_stepwise(self, data) -> None:
        """The provided ``perm`` body of the trait written out over the per-round entry points
        (src/strategies.rs:140-157): 4 full, 59 partial, 4 full, one shared cursor."""
        constants = RoundConstantsIter()
        for _ in range(TOTAL_FULL_ROUNDS // 2):
            self.apply_full_round(constants, data)
        for _ in range(PARTIAL_ROUNDS):
            self.apply_partial_round(constants, data)
        for _ in range(TOTAL_FULL_ROUNDS // 2):
            self.apply_full_round(constants, data)


def kernel_for(n_perms: int) -> int:
    """The selector ``hades252_perm_batch_dev`` runs for a batch of n_perms states (the library's one size rule)."""
    return _lib.lib().hades252_kernel_for(n_perms)


def chain_form_for(n_chains: int) -> int:
    """The form the chain entry points (sponge, absorb, verify, update) and Merkle levels run for n_chains chains."""
    return _lib.lib().hades252_chain_form_for(n_chains)


def kernel_name(kernel: int = _lib.KERNEL_DEFAULT, n_perms: int = 0) -> str:
    """Name of the __global__ function a profiler shows for ``hades252_perm_batch_dev_ex(.., n_perms, .., kernel)``."""
    name = _lib.lib().hades252_kernel_name(kernel, n_perms)
    if name is None:
        raise ValueError("unknown kernel selector %r" % (kernel,))
    return name.decode()


def warm_up(n_perms_hint: int = 0) -> None:
    """``hades252_warm_up``: load the code object and pre-create the pipe a host batch of ``n_perms_hint`` states would take."""
    check(_lib.lib().hades252_warm_up(int(n_perms_hint)), "warm_up")


def trim() -> None:
    """``hades252_trim``: give back everything the pool of pipes caches (device memory, streams, staging buffers)."""
    check(_lib.lib().hades252_trim(), "trim")


def pool_bytes() -> int:
    """``hades252_pool_bytes``: device memory the pool holds right now."""
    return int(_lib.lib().hades252_pool_bytes())


def fault_inject(spec: str | None) -> None:
    """Test hook ``hades252_fault_inject``: "<site>:<nth>" arms, None disarms."""
    check(_lib.lib().hades252_fault_inject(None if spec is None else spec.encode()), "fault_inject")


class HostBuffer:
    """Page-locked host memory for the host-pointer path (``hades252_host_alloc`` / ``_free``): what a Rust caller
    keeps its long-lived ``Vec<BlsScalar>`` in, so that ``perm`` goes straight to DMA instead of page-locking the
    slice on every call.  ``.array`` is a numpy uint64 view (20 limbs per state); close() frees the memory (the
    view must not be used afterwards)."""

    def __init__(self, n_states: int):
        self.ptr = ctypes.c_void_p()
        self.nbytes = max(1, n_states) * STATE_BYTES
        check(_lib.lib().hades252_host_alloc(ctypes.byref(self.ptr), self.nbytes), "host_alloc")
        raw = (ctypes.c_uint64 * (n_states * WIDTH * 4)).from_address(self.ptr.value)
        self.array = np.frombuffer(raw, dtype=np.uint64)

    def close(self) -> None:
        if self.ptr:
            self.array = None
            check(_lib.lib().hades252_host_free(self.ptr), "host_free")
            self.ptr = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def host_register(arr: np.ndarray) -> None:
    """Page-lock an existing host array in place, once (``hades252_host_register``)."""
    check(_lib.lib().hades252_host_register(arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes), "host_register")


def host_unregister(arr: np.ndarray) -> None:
    check(_lib.lib().hades252_host_unregister(arr.ctypes.data_as(ctypes.c_void_p)), "host_unregister")


def host_is_pinned(arr: np.ndarray) -> bool:
    return bool(_lib.lib().hades252_host_is_pinned(arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))


def perm_multi(data: np.ndarray, n_workers: int = 0, virtual: bool = False) -> None:
    """``hades252_perm_batch_multi_ex``: a host batch sharded over n_workers host threads / devices (0 = every
    visible device), contiguous range per worker, no collective.  ``virtual``: worker g runs on device
    g % device_count, so n_workers may exceed the number of GPUs."""
    if data.dtype != np.uint64 or not data.flags["C_CONTIGUOUS"]:
        raise TypeError("perm_multi: host buffers must be C-contiguous numpy uint64")
    if data.size % (WIDTH * 4) != 0:
        raise ValueError("perm_multi: %d limbs is not a whole number of %d-word states" % (data.size, WIDTH))
    check(_lib.lib().hades252_perm_batch_multi_ex(data.ctypes.data_as(ctypes.c_void_p), data.size // (WIDTH * 4),
                                                  n_workers, _lib.MULTI_VIRTUAL if virtual else 0), "perm_multi")


def _host_u64(a, what: str) -> np.ndarray:
    if not isinstance(a, np.ndarray) or a.dtype != np.uint64 or not a.flags["C_CONTIGUOUS"]:
        raise TypeError("%s: host buffers must be C-contiguous numpy uint64" % what)
    return a


def merkle_root_host(leaves: np.ndarray, arity: int, tag_mont: int, out_idx: int = 1, pad: np.ndarray | None = None) -> np.ndarray:
    """``hades252_merkle_root``: root (4 limbs) of the tree over leaves held in HOST memory (n x 4 uint64); the leaves travel
    to the device in chunks while the first tree level is hashed behind them.  ``pad``: host table [depth, 4] or None."""
    leaves = _host_u64(leaves, "merkle_root_host")
    if leaves.size % 4:
        raise ValueError("merkle_root_host: %d limbs is not a whole number of scalars" % leaves.size)
    n = leaves.size // 4
    depth = merkle_depth(n, arity, "merkle_root_host")
    pptr = None
    if pad is not None:
        pad = _host_u64(pad, "merkle_root_host")
        if pad.size != depth * 4:
            raise ValueError("merkle_root_host: the padding table needs %d digests" % depth)
        pptr = pad.ctypes.data_as(ctypes.c_void_p)
    root = np.zeros(4, dtype=np.uint64)
    check(_lib.lib().hades252_merkle_root(leaves.ctypes.data_as(ctypes.c_void_p), n, arity, _tag_arr(tag_mont), out_idx,
                                          pptr, root.ctypes.data_as(ctypes.c_void_p)), "merkle_root_host")
    return root


def merkle_root_multi(leaves: np.ndarray, arity: int, tag_mont: int, out_idx: int = 1, n_workers: int = 0, virtual: bool = False):
    """``hades252_merkle_root_multi``: the root of a FULL tree (arity^k leaves in host memory), sub-trees sharded over
    n_workers devices (0 = all visible); ``virtual``: worker g on device g % device_count."""
    leaves = _host_u64(leaves, "merkle_root_multi")
    root = np.zeros(4, dtype=np.uint64)
    check(_lib.lib().hades252_merkle_root_multi(leaves.ctypes.data_as(ctypes.c_void_p), leaves.size // 4, arity,
                                                _tag_arr(tag_mont), out_idx, n_workers,
                                                _lib.MULTI_VIRTUAL if virtual else 0, root.ctypes.data_as(ctypes.c_void_p)),
          "merkle_root_multi")
    return root


def sponge_hash_host(msgs: np.ndarray, n_msgs: int, msg_len: int, capacity_mont: int, pad_mode: int = 1) -> np.ndarray:
    """``hades252_sponge_hash``: digests [n_msgs, 4] of n_msgs fixed-length messages held in HOST memory."""
    msgs = _host_u64(msgs, "sponge_hash_host")
    if msgs.size != n_msgs * msg_len * 4:
        raise ValueError("sponge_hash_host: buffer is not %d messages of %d scalars" % (n_msgs, msg_len))
    out = np.zeros((n_msgs, 4), dtype=np.uint64)
    check(_lib.lib().hades252_sponge_hash(msgs.ctypes.data_as(ctypes.c_void_p) if msgs.size else None, n_msgs, msg_len,
                                          _tag_arr(capacity_mont), pad_mode, out.ctypes.data_as(ctypes.c_void_p)),
          "sponge_hash_host")
    return out


def sponge_hash_var_host(scalars: np.ndarray, offsets: np.ndarray, lengths: np.ndarray, capacity_mont: int, pad_mode: int = 1):
    """``hades252_sponge_hash_var``: (digests [n, 4], number of out-of-pool messages) for ragged messages in HOST memory."""
    scalars, offsets, lengths = (_host_u64(a, "sponge_hash_var_host") for a in (scalars, offsets, lengths))
    if scalars.size % 4 or offsets.size != lengths.size:
        raise ValueError("sponge_hash_var_host: malformed pool / index arrays")
    n = offsets.size
    out = np.zeros((n, 4), dtype=np.uint64)
    bad = ctypes.c_size_t(0)
    check(_lib.lib().hades252_sponge_hash_var(scalars.ctypes.data_as(ctypes.c_void_p) if scalars.size else None, scalars.size // 4,
                                              offsets.ctypes.data_as(ctypes.c_void_p), lengths.ctypes.data_as(ctypes.c_void_p),
                                              n, _tag_arr(capacity_mont), pad_mode, out.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.byref(bad)), "sponge_hash_var_host")
    return out, int(bad.value)


def perm_trace(states_t, kernel: int = _lib.KERNEL_DEFAULT, out=None):
    """State after every round (round-major: result[r] is the batch after round r); the input is
    left untouched.  Witness pre-computation for the reference's GadgetStrategy
    (src/strategies/gadget.rs:41-133)."""
    import torch
    ptr, n, dev = _dev_buffer(states_t, STATE_BYTES, "perm_trace")
    trace = torch.empty((Strategy.rounds(), n, WIDTH, 4), dtype=torch.int64, device=dev) if out is None else out
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_perm_trace_dev_ex(ptr, trace.data_ptr(), n, _stream_ptr(dev), kernel), "perm_trace")
    return trace


def perm_trace_scaled(states_t, out=None):
    """The per-round trace in SCALED form (include/hades252.h: hades252_perm_trace_scaled_dev): same shape as
    ``perm_trace``; ``true[r][w] = scaled[r][w] * mul[r] + add[r][w]`` with the tables of ``trace_scale_table()``."""
    import torch
    ptr, n, dev = _dev_buffer(states_t, STATE_BYTES, "perm_trace_scaled")
    trace = torch.empty((Strategy.rounds(), n, WIDTH, 4), dtype=torch.int64, device=dev) if out is None else out
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_perm_trace_scaled_dev(ptr, trace.data_ptr(), n, _stream_ptr(dev)), "perm_trace_scaled")
    return trace


def trace_scale_table():
    """(mul [67, 4], add [67, 5, 4]) uint64 numpy arrays: in-memory BlsScalar limbs (host tables, no device call)."""
    mul = np.zeros((Strategy.rounds(), 4), dtype=np.uint64)
    add = np.zeros((Strategy.rounds(), WIDTH, 4), dtype=np.uint64)
    check(_lib.lib().hades252_perm_trace_scale_table(mul.ctypes.data_as(ctypes.c_void_p), add.ctypes.data_as(ctypes.c_void_p)),
          "trace_scale_table")
    return mul, add


def witness_wires() -> int:
    """Gate outputs per permutation (972): the first dimension of what ``perm_witness`` returns."""
    return int(_lib.lib().hades252_witness_wires())


def perm_witness(states_t, out=None):
    """Every gate output of the reference's GadgetStrategy (src/strategies/gadget.rs:41-133) for every state:
    [972, n, 4] int64, wire-major in gate order (include/hades252.h).  The input is left untouched."""
    import torch
    ptr, n, dev = _dev_buffer(states_t, STATE_BYTES, "perm_witness")
    n_wires = _lib.lib().hades252_witness_wires()
    wires = torch.empty((n_wires, n, 4), dtype=torch.int64, device=dev) if out is None else out
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_perm_witness_dev(ptr, wires.data_ptr(), n, _stream_ptr(dev)), "perm_witness")
    return wires


FR_ADD, FR_MUL, FR_SQUARE, FR_FROM_RAW, FR_REDUCE_SIGNED = 0, 1, 2, 3, 4
FR_IMPL_SATURATED32, FR_IMPL_RADIX29 = 0, 1


def fr_op(op: int, a_t, b_t=None, impl: int = FR_IMPL_RADIX29):
    """Batched ``BlsScalar`` add / mul / square / from_raw on device (include/hades252.h)."""
    import torch
    ptr, n, dev = _dev_buffer(a_t, 32, "fr_op")
    bptr = 0
    if b_t is not None:
        bptr, nb, _ = _dev_buffer(b_t, 32, "fr_op")
        if nb != n:
            raise ValueError("fr_op: operand sizes differ")
    out = torch.empty_like(a_t)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_fr_op_dev(op, impl, ptr, bptr, out.data_ptr(), n, _stream_ptr(dev)), "fr_op")
    return out


# ---- helpers around the strategy (wire format, Merkle, synthetic data) ----------------------
def from_bytes(bytes_t, out_t=None):
    """``BlsScalar::from_bytes`` on device: 32-byte canonical LE -> Montgomery limbs.
    Raises ValueError if any input is >= p (the reference returns an error option)."""
    import torch
    ptr, n, dev = _dev_buffer(bytes_t, 32, "from_bytes")
    out_t = torch.empty_like(bytes_t) if out_t is None else out_t
    optr, n2, _ = _dev_buffer(out_t, 32, "from_bytes")
    assert n2 == n
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_from_bytes_dev(ptr, optr, n, bad.data_ptr(), _stream_ptr(dev)), "from_bytes")
    if int(bad.item()) != 0:
        raise ValueError("from_bytes: %d scalar(s) not canonical (>= p)" % int(bad.item()))
    return out_t


def to_bytes(limbs_t, out_t=None):
    """``BlsScalar::to_bytes`` on device."""
    import torch
    ptr, n, dev = _dev_buffer(limbs_t, 32, "to_bytes")
    out_t = torch.empty_like(limbs_t) if out_t is None else out_t
    optr, n2, _ = _dev_buffer(out_t, 32, "to_bytes")
    assert n2 == n
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_to_bytes_dev(ptr, optr, n, _stream_ptr(dev)), "to_bytes")
    return out_t


def _tag_arr(tag_mont: int):
    return (ctypes.c_uint64 * 4)(*[(tag_mont >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)])


def merkle_depth(n: int, arity: int, what: str = "merkle") -> int:
    """Levels above the leaves (n_l = ceil(n_{l-1} / arity)); raises for an invalid shape."""
    d = _lib.lib().hades252_merkle_depth(n, arity)
    if d < 1:
        raise ValueError("%s: need arity 2..4 and at least 2 leaves (got arity %d, %d leaves)" % (what, arity, n))
    return d


def merkle_level_sizes(n: int, arity: int):
    """[n_1, n_2, ..., 1]: nodes per level above the leaves."""
    out = []
    while n > 1:
        n = -(-n // arity)
        out.append(n)
    return out


def _pad_ptr(pad_t, depth: int, what: str):
    if pad_t is None:
        return 0
    ptr, n, _ = _dev_buffer(pad_t, 32, what)
    if n < depth:
        raise ValueError("%s: the padding table needs one digest per level (%d)" % (what, depth))
    return ptr


def merkle_level(children_t, arity: int, tag_mont: int, out_idx: int = 1, pad=None):
    """One tree level: parent = perm([tag, c_0 .. c_{arity-1}, 0 ..])[out_idx], arity 1..4.  A ragged level (child
    count not a multiple of the arity) takes ``pad`` (one 32-byte digest on the device; None = zero) for the missing
    children."""
    import torch
    if arity not in (1, 2, 3, 4):
        raise ValueError("merkle_level: arity must be 1..4")
    ptr, n_children, dev = _dev_buffer(children_t, 32, "merkle_level")
    n = -(-n_children // arity)
    parents = torch.empty((n, 4), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_merkle_level_pad_dev(ptr, n_children, parents.data_ptr(), arity, _tag_arr(tag_mont),
                                                       out_idx, _pad_ptr(pad, 1, "merkle_level"), _stream_ptr(dev)),
              "merkle_level")
    return parents


def merkle_empty_digests(arity: int, depth: int, e0_mont: int, tag_mont: int, out_idx: int = 1, device="cuda"):
    """The padding table of empty subtrees: [depth, 4] int64 -- pad[0] = e0, pad[l+1] = parent of `arity` x pad[l]."""
    import torch
    pad = torch.empty((depth, 4), dtype=torch.int64, device=device)
    with torch.cuda.device(pad.device):
        check(_lib.lib().hades252_merkle_empty_digests_dev(arity, depth, _tag_arr(e0_mont), _tag_arr(tag_mont), out_idx,
                                                           pad.data_ptr(), _stream_ptr(pad.device)), "merkle_empty_digests")
    return pad


def merkle_root(leaves_t, arity: int, tag_mont: int, out_idx: int = 1, scratch=None, pad=None):
    """Root of the arity-`arity` tree over ``leaves_t`` (n_leaves x 32 B; any n_leaves >= 2, arity 2..4)."""
    import torch
    ptr, n, dev = _dev_buffer(leaves_t, 32, "merkle_root")
    depth = merkle_depth(n, arity, "merkle_root")
    need = _lib.lib().hades252_merkle_scratch_bytes(n, arity)
    if scratch is None:
        scratch = torch.empty(max(need // 8, 2), dtype=torch.int64, device=dev)
    sptr = scratch.data_ptr()
    sbytes = scratch.numel() * scratch.element_size()
    root = torch.empty(4, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_merkle_root_pad_dev(ptr, n, arity, sptr, sbytes, _tag_arr(tag_mont), out_idx,
                                                      _pad_ptr(pad, depth, "merkle_root"), root.data_ptr(),
                                                      _stream_ptr(dev)), "merkle_root")
    return root


def merkle_build(leaves_t, arity: int, tag_mont: int, out_idx: int = 1, pad=None):
    """Every level of the tree: [n_1 + n_2 + ... + 1, 4] int64 -- level 1 first, the root last."""
    import torch
    ptr, n, dev = _dev_buffer(leaves_t, 32, "merkle_build")
    depth = merkle_depth(n, arity, "merkle_build")
    tree = torch.empty((sum(merkle_level_sizes(n, arity)), 4), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_merkle_build_pad_dev(ptr, n, arity, _tag_arr(tag_mont), out_idx,
                                                       _pad_ptr(pad, depth, "merkle_build"), tree.data_ptr(),
                                                       _stream_ptr(dev)), "merkle_build")
    return tree


def merkle_open(leaves_t, tree_t, arity: int, indices_t, pad=None):
    """Authentication paths: [n_queries, depth, arity-1, 4] int64 (siblings in child order, the path node's
    own position (index // arity^l) % arity skipped; positions past the end of a level read pad[l])."""
    import torch
    ptr, n, dev = _dev_buffer(leaves_t, 32, "merkle_open")
    depth = merkle_depth(n, arity, "merkle_open")
    tptr, nt, _ = _dev_buffer(tree_t, 32, "merkle_open")
    if nt != sum(merkle_level_sizes(n, arity)):
        raise ValueError("merkle_open: tree buffer does not belong to %d leaves" % n)
    iptr, nq, _ = _dev_buffer(indices_t, 8, "merkle_open")
    if nq and int(indices_t.max().item()) >= n:
        raise IndexError("merkle_open: leaf index out of range")
    paths = torch.empty((nq, depth, arity - 1, 4), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_merkle_open_pad_dev(ptr, tptr, n, arity, iptr, nq, _pad_ptr(pad, depth, "merkle_open"),
                                                      paths.data_ptr(), _stream_ptr(dev)), "merkle_open")
    return paths


def merkle_verify(leaf_values_t, indices_t, paths_t, arity: int, tag_mont: int, out_idx: int = 1):
    """Roots recomputed from (leaf value, index, opening) per query: [n_queries, 4] int64."""
    import torch
    lptr, nq, dev = _dev_buffer(leaf_values_t, 32, "merkle_verify")
    iptr, nq2, _ = _dev_buffer(indices_t, 8, "merkle_verify")
    if nq2 != nq or paths_t.shape[0] != nq:
        raise ValueError("merkle_verify: leaves, indices and paths differ in count")
    depth = int(paths_t.shape[1])
    pptr = paths_t.data_ptr() if paths_t.numel() else 0
    roots = torch.empty((nq, 4), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_merkle_verify_dev(lptr, iptr, pptr, nq, depth, arity, _tag_arr(tag_mont), out_idx,
                                                    roots.data_ptr(), _stream_ptr(dev)), "merkle_verify")
    return roots


def merkle_update(leaves_t, tree_t, arity: int, indices_t, tag_mont: int, out_idx: int = 1, pad=None):
    """Re-hash, in place in ``tree_t``, the ancestors of the leaves named by ``indices_t`` (device int64/uint64) after the
    caller has overwritten those rows of ``leaves_t``: depth x n_updates permutations instead of the whole tree."""
    import torch
    ptr, n, dev = _dev_buffer(leaves_t, 32, "merkle_update")
    depth = merkle_depth(n, arity, "merkle_update")
    tptr, nt, _ = _dev_buffer(tree_t, 32, "merkle_update")
    if nt != sum(merkle_level_sizes(n, arity)):
        raise ValueError("merkle_update: tree buffer does not belong to %d leaves" % n)
    iptr, nq, _ = _dev_buffer(indices_t, 8, "merkle_update")
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_merkle_update_dev(ptr, tptr, n, arity, _tag_arr(tag_mont), out_idx,
                                                    _pad_ptr(pad, depth, "merkle_update"), iptr, nq, _stream_ptr(dev)),
              "merkle_update")
    return tree_t


def merkle_forest(leaves_t, n_trees: int, arity: int, tag_mont: int, out_idx: int = 1, scratch=None):
    """Roots of n_trees equal trees (leaves contiguous, tree after tree; leaves per tree a power of the arity)."""
    import torch
    ptr, n, dev = _dev_buffer(leaves_t, 32, "merkle_forest")
    if n_trees <= 0 or n % n_trees:
        raise ValueError("merkle_forest: %d leaves do not split into %d trees" % (n, n_trees))
    per = n // n_trees
    need = _lib.lib().hades252_merkle_forest_scratch_bytes(n_trees, per, arity)
    if scratch is None:
        scratch = torch.empty(max(need // 8, 2), dtype=torch.int64, device=dev)
    roots = torch.empty((n_trees, 4), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_merkle_forest_dev(ptr, n_trees, per, arity, scratch.data_ptr(),
                                                    scratch.numel() * scratch.element_size(), _tag_arr(tag_mont), out_idx,
                                                    roots.data_ptr(), _stream_ptr(dev)), "merkle_forest")
    return roots


def merkle4_level(children_t, tag_mont: int, out_idx: int = 1):
    return merkle_level(children_t, 4, tag_mont, out_idx)


def merkle4_root(leaves_t, tag_mont: int, out_idx: int = 1, scratch=None):
    """Root of the arity-4 tree over ``leaves_t`` (BASELINE config 4)."""
    return merkle_root(leaves_t, 4, tag_mont, out_idx, scratch)


def sponge_hash(msgs_t, msg_len: int, capacity_mont: int, pad_mode: int = 1):
    """Batched fixed-length sponge (include/hades252.h): msgs_t holds n messages of msg_len
    scalars; returns [n, 4] int64 digests (word 1 of the final state)."""
    import torch
    if msg_len > 0:
        ptr, n_scalars, dev = _dev_buffer(msgs_t, 32, "sponge_hash")
        if n_scalars % msg_len != 0:
            raise ValueError("sponge_hash: buffer is not a whole number of messages")
        n = n_scalars // msg_len
    else:
        raise ValueError("sponge_hash: msg_len must be positive for a tensor batch")
    out = torch.empty((n, 4), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_sponge_hash_dev(ptr, n, msg_len, _tag_arr(capacity_mont), pad_mode,
                                                  out.data_ptr(), _stream_ptr(dev)), "sponge_hash")
    return out


def sponge_hash_var(scalars_t, offsets_t, lengths_t, capacity_mont: int, pad_mode: int = 1, sort: bool = False):
    """Batched variable-length sponge: message i = scalars[offsets[i] : offsets[i] + lengths[i]]
    (offsets / lengths: int64 CUDA tensors, in scalars).  Returns [n, 4] int64 digests.  ``sort``: order the messages
    by block count on the device first (ragged batches: the 64 messages of a wave then need the same number of
    permutations); same digests."""
    import torch
    sptr, n_scalars, dev = _dev_buffer(scalars_t, 32, "sponge_hash_var")
    optr, n, _ = _dev_buffer(offsets_t, 8, "sponge_hash_var")
    lptr, n2, _ = _dev_buffer(lengths_t, 8, "sponge_hash_var")
    if n != n2:
        raise ValueError("sponge_hash_var: offsets and lengths differ in size")
    out = torch.empty((n, 4), dtype=torch.int64, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch, sbytes = None, 0
    if sort:
        sbytes = _lib.lib().hades252_sponge_sort_scratch_bytes(n)
        scratch = torch.empty((sbytes + 15) // 16 * 2, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_sponge_hash_var_ex_dev(sptr, n_scalars, optr, lptr, n, _tag_arr(capacity_mont), pad_mode,
                                                         out.data_ptr(), bad.data_ptr(),
                                                         scratch.data_ptr() if sort else 0, sbytes, _stream_ptr(dev)),
              "sponge_hash_var")
    if int(bad.item()) != 0:
        raise IndexError("sponge_hash_var: %d message(s) reach outside the scalar pool" % int(bad.item()))
    return out


# Named, NOT pinned parameter sets (dusk-poseidon is outside the reference tree, README.md:9): include/hades252.h
SPONGE_PRESETS = {"sponge/pad10": {"capacity": 1 << 64, "pad_mode": 1, "digest_word": 1},
                  "merkle/arity4": {"tag": 15, "arity": 4, "digest_word": 1}}


class SpongeStates:
    """Streaming sponge (``hades252_sponge_init_dev`` / ``_absorb_dev`` / ``_squeeze_dev``): n states resident on the
    device; absorb blocks of 4 scalars whenever they arrive, squeeze when done."""

    def __init__(self, n: int, capacity_mont: int, device="cuda"):
        import torch
        self.states = torch.empty((n, WIDTH, 4), dtype=torch.int64, device=device)
        with torch.cuda.device(self.states.device):
            check(_lib.lib().hades252_sponge_init_dev(self.states.data_ptr(), n, _tag_arr(capacity_mont),
                                                      _stream_ptr(self.states.device)), "sponge_init")

    def absorb(self, blocks_t) -> None:
        """blocks_t: [n, blocks_each, 4 scalars] (any shape with n * blocks_each * 128 bytes, state-major)."""
        import torch
        n = self.states.shape[0]
        ptr, n_blocks, dev = _dev_buffer(blocks_t, 128, "sponge_absorb")
        if n == 0 or n_blocks % n:
            raise ValueError("sponge_absorb: %d blocks do not split evenly over %d states" % (n_blocks, n))
        with torch.cuda.device(dev):
            check(_lib.lib().hades252_sponge_absorb_dev(self.states.data_ptr(), ptr, n, n_blocks // n, _stream_ptr(dev)),
                  "sponge_absorb")

    def squeeze(self, word: int = 1):
        import torch
        n, dev = self.states.shape[0], self.states.device
        out = torch.empty((n, 4), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            check(_lib.lib().hades252_sponge_squeeze_dev(self.states.data_ptr(), out.data_ptr(), n, word, _stream_ptr(dev)),
                  "sponge_squeeze")
        return out


GEN_SEED = 0x4861646573323532


def gen_b(n_elems: int, device, first_elem: int = 0, seed: int = GEN_SEED, out=None):
    """Generator B on device: n_elems scalars starting at global element index."""
    import torch
    out = torch.empty((n_elems, 4), dtype=torch.int64, device=device) if out is None else out
    with torch.cuda.device(out.device):
        check(_lib.lib().hades252_gen_b_dev(out.data_ptr(), first_elem, n_elems, seed, _stream_ptr(out.device)),
              "gen_b")
    return out


def gen_a(n_elems: int, device, first_elem: int = 0):
    import torch
    out = torch.empty((n_elems, 4), dtype=torch.int64, device=device)
    with torch.cuda.device(out.device):
        check(_lib.lib().hades252_gen_a_dev(out.data_ptr(), first_elem, n_elems, _stream_ptr(out.device)), "gen_a")
    return out


def digest(t, first_index: int = 0):
    """256-bit position-sensitive digest of a device buffer (4 python ints)."""
    import torch
    ptr, n, dev = _dev_buffer(t, 8, "digest")
    out = torch.empty(4, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(_lib.lib().hades252_digest_dev(ptr, first_index, n, out.data_ptr(), _stream_ptr(dev)), "digest")
    return [int(v) & 0xFFFFFFFFFFFFFFFF for v in out.cpu().tolist()]
