"""CPU tier: `python bench.py --gpus N` without torchrun is a launcher -- it must spawn N rank processes with the
torchrun environment, never touch the GPU itself, and exit with the ranks' status.  On this GPU-less container every
rank fails loudly ("needs a GPU: no CPU fallback"), which is exactly what the launcher has to propagate."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU tier runs the real two-rank bench (tests/test_bench_contract.py)")
def test_launcher_spawns_ranks_and_propagates_failure():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--steps", "1", "--warmup", "0", "--perms-per-gpu", "64",
                        "--dist-backend", "gloo"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU: hades252_amd has no CPU fallback") == 3     # one per rank
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]                        # no result line


def test_rank_mode_reads_torchrun_environment():
    """With WORLD_SIZE set the process is a rank, not a launcher (it must not spawn anything)."""
    env = dict(os.environ, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    if torch.cuda.is_available():
        pytest.skip("needs a GPU-less machine to stop before the rendezvous")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and r.stderr.count("bench.py needs a GPU") == 1
