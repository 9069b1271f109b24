"""Build libhades252.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m hades252_amd.build [--force]

The library is built into ``hades252_amd/csrc/libhades252.so`` so that it travels with the source
tree (it is git-ignored).  hipcc cross-compiles without a GPU.

Staleness is decided by a content hash of the sources (kept in ``libhades252.so.stamp``), not by
mtimes -- a copied tree does not preserve them.  Builds are serialised with a file lock and the
library is replaced atomically, so several ranks of one job may call ``build()`` at once.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libhades252.so")
STAMP = LIB + ".stamp"
LOCK = LIB + ".lock"
SOURCES = ["hades252.hip"]
# device code: arithmetic, kernels, generated tables
DEVICE_DEPS = ["fr32.hpp", "staging.hpp", "hades_literal.hpp", "hades_fast.hpp", "k_perm_fast.hpp", "hades_coop.hpp", "hades_lanes.hpp",
               "device_tables.hpp", "kernels_perm.hpp", "kernels_merkle.hpp", "kernels_sponge.hpp", "kernels_aux.hpp",
               "hades_constants.inc"]
# host code that decides WHAT is launched and with which geometry / LDS / level-fusion policy: a kernel's traffic per launch
# and the launches that make a tree depend on it
LAUNCH_POLICY_DEPS = ["launch.hpp", "abi_perm.hpp", "abi_merkle.hpp", "abi_sponge.hpp"]
# host plumbing (error / fault hook, page-locked memory, pipe pool, chunk pipeline, one-shot host callers, generators)
HOST_DEPS = ["hades252.hip", "host_fault.hpp", "abi_util.hpp", "host_pin.hpp", "host_pool.hpp", "host_pipe.hpp", "host_callers.hpp"]
DEPS = HOST_DEPS + LAUNCH_POLICY_DEPS + DEVICE_DEPS + [os.path.join("..", "..", "include", "hades252.h")]
# what the dominant kernel (k_perm_fast) is made of: profiles recorded for it stay valid while these are unchanged
PERM_FAST_DEPS = ["fr32.hpp", "staging.hpp", "hades_fast.hpp", "k_perm_fast.hpp"]
# ... plus these tables of hades_constants.inc (other kernels' tables may change without touching k_perm_fast)
PERM_FAST_TABLES = ("HADES_FAST_L", "HADES_FAST_MDS_SMALL", "HADES_NEG_P29", "HADES_TWO_P29", "HADES_P29", "HADES_FAST_ROUND_INIT",
                    "HADES_FAST_FINAL_F", "HADES_FAST_LIN_INIT", "HADES_FAST_FINAL_LIN")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-pthread",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def source_hash() -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read())
    return h.hexdigest()


def perm_fast_hash() -> str:
    """Hash of the sources + flags that determine k_perm_fast (keys committed PMC profiles to the kernel)."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in PERM_FAST_DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read())
    keep = False
    with open(os.path.join(CSRC, "hades_constants.inc")) as f:
        for line in f:
            if line.startswith("#define "):
                keep = line.split()[1] in PERM_FAST_TABLES
            if keep:
                h.update(line.encode())
                keep = line.rstrip().endswith("\\")
    return h.hexdigest()


def device_source_hash() -> str:
    """Hash of everything that determines what a launch (or a tree build: a SEQUENCE of launches) moves through HBM: the
    device code of every kernel (kernel headers, generated tables, flags) AND the host code that picks kernels, grids, LDS
    sizes and the level / fusion policy of a tree (launch.hpp, abi_*.hpp) -- not the host plumbing (pools, pipes, page
    locking, the fault hook), whose edits leave traffic per launch as it was.  Keys the committed counter records of the
    kernels other than k_perm_fast (profiles/hbm_traffic.json `secondary_kernels`)."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in sorted(DEVICE_DEPS + LAUNCH_POLICY_DEPS):
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read())
    return h.hexdigest()


def _fresh(want: str) -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == want


def build(force: bool = False, verbose: bool = True) -> str:
    want = source_hash()
    if not force and _fresh(want):
        return LIB
    with open(LOCK, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _fresh(want):          # another process built it while we waited
                return LIB
            tmp = LIB + ".tmp.%d" % os.getpid()
            cmd = [HIPCC] + FLAGS + ["-o", tmp] + SOURCES
            if verbose:
                print("[hades252_amd.build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, cwd=CSRC, check=True)
            os.replace(tmp, LIB)
            with open(STAMP + ".tmp", "w") as f:
                f.write(want + "\n")
            os.replace(STAMP + ".tmp", STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


# ---- native measurement tool of the host-pointer boundary (tools/host_path_bench.cpp) -------------------------
# A plain C++ caller of the C ABI, linked against the SYSTEM HIP runtime like a Rust / C host would be (a PyTorch
# process loads PyTorch's own bundled runtime, under which the same pipeline overlaps its copies far worse).
ROOT = os.path.dirname(os.path.dirname(CSRC))
TOOL_SRC = os.path.join(ROOT, "tools", "host_path_bench.cpp")
TOOL_BIN = os.path.join(ROOT, "build_tools", "host_path_bench")


def build_host_path_bench(verbose: bool = True) -> str:
    build(verbose=verbose)
    h = hashlib.sha256()
    for f in (TOOL_SRC, os.path.join(ROOT, "include", "hades252.h")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    want = h.hexdigest()
    stamp = TOOL_BIN + ".stamp"
    os.makedirs(os.path.dirname(TOOL_BIN), exist_ok=True)
    if os.path.exists(TOOL_BIN) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return TOOL_BIN
    with open(LOCK, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            tmp = TOOL_BIN + ".tmp.%d" % os.getpid()
            cmd = [HIPCC, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", tmp, TOOL_SRC, "-L" + CSRC,
                   "-lhades252", "-Wl,-rpath,$ORIGIN/../hades252_amd/csrc"]
            if verbose:
                print("[hades252_amd.build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            os.replace(tmp, TOOL_BIN)
            with open(stamp, "w") as f:
                f.write(want + "\n")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return TOOL_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_host_path_bench()
    print(LIB)
