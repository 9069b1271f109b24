"""Build libhades252.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m hades252_amd.build [--force]

The library is built into ``hades252_amd/csrc/libhades252.so`` so that it travels with the source
tree (it is git-ignored).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libhades252.so")
SOURCES = ["hades252.hip"]
DEPS = ["hades252.hip", "fr32.cuh", "staging.cuh", "hades_literal.cuh", "hades_fast.cuh",
        "hades_constants.inc", os.path.join("..", "..", "include", "hades252.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-pthread",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for d in DEPS:
        p = os.path.join(CSRC, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    cmd = [HIPCC] + FLAGS + ["-o", LIB] + SOURCES
    if verbose:
        print("[hades252_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
