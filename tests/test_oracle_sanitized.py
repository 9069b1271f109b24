"""CPU tier: the C oracle under AddressSanitizer + UBSan (sanitizers run on the CPU build only;
the GPU pool offers none)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_selftest")
    src = os.path.join(ROOT, "oracle")
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    os.path.join(src, "selftest.c"), os.path.join(src, "hades_oracle.c"), "-I", src, "-lpthread",
                    "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle selftest ok" in r.stdout
