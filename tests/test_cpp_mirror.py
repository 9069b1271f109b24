"""The C++ host-side mirror (include/hades252.hpp) of the reference's Strategy/ScalarStrategy."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build_tools", "readme_example")


def build_example():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "hades252_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "readme_example.cpp"), "-L", libdir, "-lhades252",
                    "-Wl,-rpath," + libdir, "-o", EXE], check=True)


def test_cpp_mirror_compiles_and_links(hades_lib):
    """Header-only mirror compiles as C++17 and links against the C ABI (no GPU needed)."""
    build_example()
    assert os.path.exists(EXE)


def test_c_header_is_plain_c(tmp_path):
    """include/hades252.h must be consumable from C (the FFI contract: no C++ in signatures)."""
    src = tmp_path / "t.c"
    src.write_text('#include "hades252.h"\nint (*fp)(void) = hades252_rounds;\nint main(void){return fp == (int (*)(void))0;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)


@pytest.mark.gpu
def test_readme_example_runs(hades_lib):
    """README.md:50-65 usage + hades_det (scalar.rs:62-74) through the C++ mirror, with a KAT."""
    build_example()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "known answer perm([17;5])[0]: ok" in r.stdout
    assert "len != k*WIDTH rejected" in r.stdout
    assert "merkle_root / DeviceBuffer / sponge_hash against perm: ok" in r.stdout
