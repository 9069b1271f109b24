"""ctypes wrapper of oracle/_build/libhades_oracle.so for the tests (numpy uint64 in / out)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "_build", "libhades_oracle.so")
GEN_SEED = 0x4861646573323532
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % P
M64 = (1 << 64) - 1

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build():
    """Build the oracle library if it is missing or stale.  Staleness is decided by a content hash of the sources
    (a copied tree does not preserve mtimes) and builds are serialised with a file lock: the ranks of a multi-GPU
    bench all call this at once."""
    import fcntl
    import hashlib
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("hades_oracle.c", "hades_oracle_constants.h", "Makefile")]
    h = hashlib.sha256()
    for f in srcs:
        with open(f, "rb") as fh:
            h.update(fh.read())
    want = h.hexdigest()
    out_dir = os.path.join(ORACLE_DIR, "_build")
    os.makedirs(out_dir, exist_ok=True)
    stamp = os.path.join(out_dir, "libhades_oracle.so.stamp")

    def fresh():
        return os.path.exists(SO) and os.path.exists(stamp) and open(stamp).read().strip() == want

    if fresh():
        return SO
    with open(os.path.join(out_dir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not fresh():
                subprocess.run(["make", "-B", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)
                with open(stamp + ".tmp", "w") as f:
                    f.write(want + "\n")
                os.replace(stamp + ".tmp", stamp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return SO


PORTABLE_FLAGS = "gcc -O3 -march=x86-64-v3"
NATIVE_FLAGS = "gcc -O3 -march=native"


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build_native():
    """The oracle's source compiled `-O3 -march=native` ON THIS HOST, for the cpu_baseline TIMING of
    bench.py only -- the checker stays the portable build.  Rebuilt whenever the sources or the host's CPU model change
    (the tree travels between machines).  Returns the path, or None when the host cannot build or run it."""
    import fcntl
    import hashlib
    build()
    h = hashlib.sha256(host_cpu_model().encode())
    for f in ("hades_oracle.c", "hades_oracle_constants.h", "Makefile"):
        with open(os.path.join(ORACLE_DIR, f), "rb") as fh:
            h.update(fh.read())
    want = h.hexdigest()
    out_dir = os.path.join(ORACLE_DIR, "_build")
    so = os.path.join(out_dir, "libhades_oracle_native.so")
    stamp = so + ".stamp"

    def fresh():
        return os.path.exists(so) and os.path.exists(stamp) and open(stamp).read().strip() == want

    try:
        with open(os.path.join(out_dir, ".lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if not fresh():
                    subprocess.run(["make", "-B", "-C", ORACLE_DIR, "native"], check=True, stdout=subprocess.DEVNULL,
                                   stderr=subprocess.DEVNULL)
                    with open(stamp + ".tmp", "w") as f:
                        f.write(want + "\n")
                    os.replace(stamp + ".tmp", stamp)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
        return so
    except Exception:
        return None


def limbs_of(m):
    return [(m >> (64 * k)) & M64 for k in range(4)]


def int_of(limbs):
    return sum(int(limbs[k]) << (64 * k) for k in range(4))


def _p(a):
    return a.ctypes.data_as(_u64p)


class Oracle:
    def __init__(self, so):
        self.l = ctypes.CDLL(so)
        self.l.hades_oracle_init()
        self.l.hades_oracle_from_bytes.restype = ctypes.c_int
        self.ncpu = os.cpu_count() or 1

    def perm_batch(self, states, threads=None):
        """states: uint64 array, size multiple of 20; returns a permuted COPY."""
        out = np.ascontiguousarray(states, dtype=np.uint64).copy()
        assert out.size % 20 == 0
        self.l.hades_oracle_perm_batch(_p(out), ctypes.c_size_t(out.size // 20),
                                       ctypes.c_int(threads or self.ncpu))
        return out

    def perm_trace(self, state):
        st = np.ascontiguousarray(state, dtype=np.uint64).copy()
        tr = np.zeros(67 * 20, dtype=np.uint64)
        self.l.hades_oracle_perm_trace(_p(st), _p(tr))
        return st, tr.reshape(67, 5, 4)

    def add_round_key(self, states, rnd):
        out = np.ascontiguousarray(states, dtype=np.uint64).copy()
        self.l.hades_oracle_add_round_key(_p(out), ctypes.c_size_t(out.size // 20), ctypes.c_int(rnd))
        return out

    def add_round_key_at(self, states, cursor):
        out = np.ascontiguousarray(states, dtype=np.uint64).copy()
        self.l.hades_oracle_add_round_key_at(_p(out), ctypes.c_size_t(out.size // 20), ctypes.c_int(cursor))
        return out

    def quintic_s_box(self, scalars):
        out = np.ascontiguousarray(scalars, dtype=np.uint64).copy()
        self.l.hades_oracle_quintic_s_box(_p(out), ctypes.c_size_t(out.size // 4))
        return out

    def mul_matrix(self, states):
        out = np.ascontiguousarray(states, dtype=np.uint64).copy()
        self.l.hades_oracle_mul_matrix(_p(out), ctypes.c_size_t(out.size // 20))
        return out

    def full_round(self, states, rnd):
        st = self.add_round_key(states, rnd)
        st = self.quintic_s_box(st)
        return self.mul_matrix(st)

    def full_round_at(self, states, cursor):
        return self.mul_matrix(self.quintic_s_box(self.add_round_key_at(states, cursor)))

    def partial_round_at(self, states, cursor):
        st = self.add_round_key_at(states, cursor).reshape(-1, 5, 4)
        st[:, 4, :] = self.quintic_s_box(st[:, 4, :].copy()).reshape(-1, 4)
        return self.mul_matrix(st.reshape(-1))

    def partial_round(self, states, rnd):
        st = self.add_round_key(states, rnd).reshape(-1, 5, 4)
        st[:, 4, :] = self.quintic_s_box(st[:, 4, :].copy()).reshape(-1, 4)
        return self.mul_matrix(st.reshape(-1))

    def gen_b(self, first_elem, n_elems, seed=GEN_SEED):
        out = np.zeros(n_elems * 4, dtype=np.uint64)
        self.l.hades_oracle_gen_b(_p(out), ctypes.c_uint64(first_elem), ctypes.c_size_t(n_elems), ctypes.c_uint64(seed))
        return out

    def gen_a(self, first_elem, n_elems):
        out = np.zeros(n_elems * 4, dtype=np.uint64)
        self.l.hades_oracle_gen_a(_p(out), ctypes.c_uint64(first_elem), ctypes.c_size_t(n_elems))
        return out

    def merkle4_level(self, children, tag_mont, out_idx=1, threads=None):
        ch = np.ascontiguousarray(children, dtype=np.uint64)
        n = ch.size // 16
        out = np.zeros(n * 4, dtype=np.uint64)
        tag = np.array(limbs_of(tag_mont), dtype=np.uint64)
        self.l.hades_oracle_merkle4_level(_p(ch), _p(out), ctypes.c_size_t(n), _p(tag), ctypes.c_int(out_idx),
                                          ctypes.c_int(threads or self.ncpu))
        return out

    def merkle_level(self, children, arity, tag_mont, out_idx=1, threads=None):
        ch = np.ascontiguousarray(children, dtype=np.uint64)
        n = ch.size // (4 * arity)
        out = np.zeros(n * 4, dtype=np.uint64)
        tag = np.array(limbs_of(tag_mont), dtype=np.uint64)
        self.l.hades_oracle_merkle_level(_p(ch), _p(out), ctypes.c_size_t(n), ctypes.c_int(arity), _p(tag),
                                         ctypes.c_int(out_idx), ctypes.c_int(threads or self.ncpu))
        return out

    def merkle_level_pad(self, children, arity, tag_mont, out_idx=1, pad=None, threads=None):
        """A ragged level: children padded up to a multiple of the arity with the digest `pad` (4 limbs; None = zero)."""
        ch = np.ascontiguousarray(children, dtype=np.uint64).reshape(-1, 4)
        miss = (-ch.shape[0]) % arity
        if miss:
            fill = np.zeros(4, dtype=np.uint64) if pad is None else np.ascontiguousarray(pad, dtype=np.uint64).reshape(4)
            ch = np.concatenate([ch, np.tile(fill, (miss, 1))])
        return self.merkle_level(ch.reshape(-1), arity, tag_mont, out_idx, threads)

    def merkle_tree(self, leaves, arity, tag_mont, out_idx=1, pad=None):
        """All levels above the leaves, level 1 first (the layout of hades252_merkle_build_dev), as a list.  Any number
        of leaves: level l's missing children take pad[l] (rows of 4 limbs; None = zeros)."""
        levels, cur, l = [], np.ascontiguousarray(leaves, dtype=np.uint64), 0
        while cur.size > 4:
            cur = self.merkle_level_pad(cur, arity, tag_mont, out_idx, None if pad is None else pad[l])
            levels.append(cur)
            l += 1
        return levels

    def merkle_empty_digests(self, arity, depth, e0_mont, tag_mont, out_idx=1):
        pad = [np.array(limbs_of(e0_mont), dtype=np.uint64)]
        for _ in range(depth - 1):
            pad.append(self.merkle_level(np.tile(pad[-1], arity), arity, tag_mont, out_idx, threads=1))
        return np.stack(pad)

    def merkle_verify_path(self, leaf, index, path, arity, tag_mont, out_idx=1):
        """Recompute the root from a leaf (4 limbs), its index and its opening path[l][s] (siblings in child
        order, own position skipped).  Returns the root's 4 limbs."""
        node = np.ascontiguousarray(leaf, dtype=np.uint64).reshape(4)
        path = np.ascontiguousarray(path, dtype=np.uint64).reshape(-1, max(arity - 1, 0), 4)
        for l in range(path.shape[0]):
            pos = index % arity
            sib = list(path[l])
            children = sib[:pos] + [node] + sib[pos:]
            node = self.merkle_level(np.concatenate(children), arity, tag_mont, out_idx, threads=1)
            index //= arity
        return node

    def merkle4_root(self, leaves, tag_mont, out_idx=1):
        level = np.ascontiguousarray(leaves, dtype=np.uint64)
        while level.size > 4:
            level = self.merkle4_level(level, tag_mont, out_idx)
        return level

    def sponge(self, msgs, msg_len, cap_mont, pad_mode=1):
        m = np.ascontiguousarray(msgs, dtype=np.uint64)
        n = m.size // (4 * msg_len) if msg_len else 0
        out = np.zeros(4 * n, dtype=np.uint64)
        cap = np.array(limbs_of(cap_mont), dtype=np.uint64)
        self.l.hades_oracle_sponge(_p(m), ctypes.c_size_t(n), ctypes.c_size_t(msg_len), _p(cap),
                                   ctypes.c_int(pad_mode), _p(out))
        return out

    def sponge_var(self, scalars, offsets, lengths, cap_mont, pad_mode=1):
        m = np.ascontiguousarray(scalars, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        out = np.zeros(4 * off.size, dtype=np.uint64)
        cap = np.array(limbs_of(cap_mont), dtype=np.uint64)
        self.l.hades_oracle_sponge_var(_p(m), _p(off), _p(ln), ctypes.c_size_t(off.size), _p(cap),
                                       ctypes.c_int(pad_mode), _p(out))
        return out

    def from_bytes(self, b32):
        buf = (ctypes.c_uint8 * 32)(*b32)
        out = np.zeros(4, dtype=np.uint64)
        rc = self.l.hades_oracle_from_bytes(buf, _p(out))
        return rc, out

    def to_bytes(self, limbs):
        a = np.ascontiguousarray(limbs, dtype=np.uint64)
        buf = (ctypes.c_uint8 * 32)()
        self.l.hades_oracle_to_bytes(_p(a), buf)
        return bytes(buf)

    def fr2(self, name, a, b):
        x = np.array(limbs_of(a), dtype=np.uint64)
        y = np.array(limbs_of(b), dtype=np.uint64)
        o = np.zeros(4, dtype=np.uint64)
        getattr(self.l, "hades_oracle_fr_" + name)(_p(x), _p(y), _p(o))
        return int_of(o)

    def fr1(self, name, a):
        x = np.array(limbs_of(a), dtype=np.uint64)
        o = np.zeros(4, dtype=np.uint64)
        getattr(self.l, "hades_oracle_fr_" + name)(_p(x), _p(o))
        return int_of(o)

    def round_constant(self, i):
        o = np.zeros(4, dtype=np.uint64)
        self.l.hades_oracle_round_constant(ctypes.c_int(i), _p(o))
        return int_of(o)

    def mds(self, i, j):
        o = np.zeros(4, dtype=np.uint64)
        self.l.hades_oracle_mds(ctypes.c_int(i), ctypes.c_int(j), _p(o))
        return int_of(o)


def load():
    return Oracle(build())


def load_native():
    """(Oracle tuned for this host, flags) for TIMING, or (None, reason)."""
    so = build_native()
    if so is None:
        return None, "could not build -march=native here"
    try:
        return Oracle(so), NATIVE_FLAGS
    except OSError as e:
        return None, repr(e)


def digest_ref(words, first_index=0):
    """numpy restatement of hades252_digest_dev (include/hades252.h)."""
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    idx = np.arange(w.size, dtype=np.uint64) + np.uint64(first_index)
    with np.errstate(over="ignore"):
        z = w ^ (idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xD1B54A32D192ED03))
        z = (z ^ (z >> np.uint64(32))) * np.uint64(0xD6E8FEB86659FD93)
        z = (z ^ (z >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9)
        z = z ^ (z >> np.uint64(32))
        out = [0, 0, 0, 0]
        for k in range(4):
            sel = (idx & np.uint64(3)) == np.uint64(k)
            out[k] = int(z[sel].sum(dtype=np.uint64))
    return out
