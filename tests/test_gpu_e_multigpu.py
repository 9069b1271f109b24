"""GPU tier, SURVEY section 8 row e: the multi-GPU decompositions on the one device a test box has -- Merkle trees sharded by
sub-tree (emulated ranks, virtual workers, against the oracle's committed roots) and the N > 1 path of bench.py rehearsed
with two ranks on one device."""
import ctypes
import hashlib
import json
import os
import random
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hades_spec as S  # noqa: E402,F401
from oracle_lib import P, R, limbs_of, int_of, digest_ref  # noqa: E402,F401
from gpu_common import *  # noqa: E402,F401,F403  (helpers shared by the GPU tier; fixtures torch_cuda / H: conftest.py)

pytestmark = pytest.mark.gpu


def test_merkle_sharded_emulated(torch_cuda, H, oracle):
    """Multi-GPU Merkle decomposition emulated on one device: every 'rank' builds
    its sub-tree roots, the gathered sub-roots are finished, result == single-device root."""
    torch = torch_cuda
    from hades252_amd import merkle
    tag = S.to_mont(15)
    n = 1 << 16
    leaves = H.gen_b(n, "cuda")
    ref = H.merkle4_root(leaves, tag, 1)
    for world in (1, 2, 4, 8):
        per_rank = n // world
        parts = [merkle.local_subroots(leaves[r * per_rank:(r + 1) * per_rank], n, world, tag, 1)
                 for r in range(world)]
        root = merkle.finish_from_subroots(torch.cat(parts), tag, 1)
        assert torch.equal(root.view(-1), ref.view(-1))
    assert torch.equal(merkle.merkle4_root_sharded(leaves, n, tag, 1).view(-1), ref.view(-1))


@pytest.mark.parametrize("workers", [1, 2, 3, 8, 16, 64])
def test_merkle_root_multi_workers_on_one_device(torch_cuda, hades_lib, H, oracle, workers):
    """The sub-tree sharding of SURVEY 8(e) behind the C ABI, with more workers than devices: the hipSetDevice threads, the
    sub-tree arithmetic (worker counts that are no power of the arity, more workers than sub-trees) and the final small tree."""
    for arity, k in ((4, 7), (2, 12), (3, 6), (4, 1), (2, 2)):
        n = arity ** k
        lv = oracle.gen_b(100 + n, n)
        exp = H.merkle_root_host(lv, arity, TAG[arity])
        assert (H.merkle_root_multi(lv, arity, TAG[arity], 1, workers, virtual=True) == exp).all(), (arity, k, workers)
    big = oracle.gen_b(5, 4 ** 10)                                  # >= 8 MiB: page-locked once for all workers
    assert (H.merkle_root_multi(big, 4, TAG[4], 3, workers, virtual=True) == H.merkle_root_host(big, 4, TAG[4], 3)).all()
    with pytest.raises(Exception):
        H.merkle_root_multi(oracle.gen_b(1, 100), 4, TAG[4], 1, workers, virtual=True)        # not a full tree
    ndev = hades_lib.hades252_device_count()
    if workers > ndev:
        with pytest.raises(Exception):
            H.merkle_root_multi(oracle.gen_b(1, 64), 4, TAG[4], 1, workers)                   # real devices only


def test_host_and_sharded_roots_equal_golden(torch_cuda, H, kat, oracle):
    """hades252_merkle_root (host memory, chunked upload) and hades252_merkle_root_multi (8 and 16 virtual workers: the
    8-GPU decomposition on one device) on 2^20 leaves against the committed root."""
    n = 4 ** 10
    gold = kat["merkle4_full_size"][str(n)]["root"]
    leaves = oracle.gen_b(0, n)
    assert hex(int_of(H.merkle_root_host(leaves, 4, TAG4, 1))) == gold
    for w in (8, 16, 3):
        assert hex(int_of(H.merkle_root_multi(leaves, 4, TAG4, 1, n_workers=w, virtual=True))) == gold


# ---------------------------------------------------------------------------------------------
# bench.py with N > 1, rehearsed on the one GPU of the test box (VERDICT r4 next #1e): two ranks sharing cuda:0 over gloo,
# both multi-rank records present and TRUE, the ranks' device identity gathered; without --single-device the same two
# ranks on one device must FAIL (exit 4) -- that is the check that makes a real 8-GPU run self-validating
# ---------------------------------------------------------------------------------------------
def _bench(args, expect_rc=0):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == expect_rc, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return (json.loads(lines[0]) if lines else None), r


def test_bench_two_ranks_rehearsal_prints_both_multi_rank_records(torch_cuda, hades_lib, kat):
    out, _ = _bench(["--gpus", "2", "--single-device", "--dist-backend", "gloo", "--perms-per-gpu", "1048576", "--steps", "2",
                     "--warmup", "1", "--cpu-sample", "16384"])
    assert out["n_gpus"] == 2 and out["parity_vs_cpu_sample"] is True
    d = out["dist"]
    assert d["backend"] == "gloo" and d["ranks_seen"] == 2 and d["single_device"] is True and d["distinct_devices"] is False
    assert len(out["per_gpu"]["device"]) == 2 and out["per_gpu"]["device"][0] == out["per_gpu"]["device"][1]
    assert "pci" in out["per_gpu"]["device"][0]
    c5 = out["secondary"]["config5_rehearsal"]
    assert c5["total_perms"] == 2 << 20 and len(c5["kernel_ms_per_rank"]) == 2 and c5["parity_vs_cpu_sample"] is True
    assert c5["digest_matches_one_device"] is True and c5["value"] > 0
    ms = out["secondary"]["merkle_2p24_sharded"]
    assert ms["root_matches_golden"] is True and "0x" + ms["root"].lstrip("0") == kat["merkle4_full_size"]["16777216"]["root"]
    assert ms["nodes"] == 5592405 and 0 < ms["tree_ms"] < 500 and "all_gather" in ms["exchange"]
    assert out["secondary"]["merkle_2p24"]["root_matches_golden"] is True        # the one-device tree of the same run
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_rehearsal_2ranks.json"), "w") as f:
        json.dump(out, f)


def test_bench_eight_ranks_rehearsal_is_the_production_split(torch_cuda, hades_lib, kat):
    """World size 8 -- the node BASELINE configs[4] names -- with all eight ranks on the one device of the test box, at
    rehearsal size: the tree splits into 16 sub-trees of 2^20 leaves, two per rank, the all_gather
    carries 16 sub-roots, every rank's device is reported, and both multi-rank records come out TRUE."""
    out, _ = _bench(["--gpus", "8", "--single-device", "--dist-backend", "gloo", "--perms-per-gpu", "262144", "--steps", "2",
                     "--warmup", "1", "--cpu-sample", "4096"])
    assert out["n_gpus"] == 8 and out["dist"]["ranks_seen"] == 8 and len(out["per_gpu"]["device"]) == 8
    assert out["parity_vs_cpu_sample"] is True and len(out["per_gpu"]["kernel_ms_per_rank"]) == 8
    c5 = out["secondary"]["config5_rehearsal"]
    assert c5["total_perms"] == 8 * 262144 and c5["parity_vs_cpu_sample"] is True and c5["digest_matches_one_device"] is True
    ms = out["secondary"]["merkle_2p24_sharded"]
    assert ms["root_matches_golden"] is True and "2 sub-tree(s) of 1048576 leaves per rank" in ms["workload"]
    assert "16 sub-roots" in ms["workload"] and "512 bytes" in ms["exchange"]
    with open(os.path.join(ROOT, "gpurun_out", "bench_rehearsal_8ranks.json"), "w") as f:
        json.dump(out, f)


def test_bench_config5_full_size_through_the_eight_rank_path(torch_cuda, hades_lib, kat):
    """BASELINE configs[4] through bench.py's own N = 8 code path at its REAL size -- 2^27 states per rank, all eight ranks on
    the one device of the test box (160 GiB resident) -- the sum of the ranks' shard digests against the CPU oracle's digest of
    all 2^30 outputs.  What an 8-GPU node will run, minus the seven other GPUs."""
    import gc
    gc.collect()
    torch_cuda.cuda.empty_cache()                      # what earlier tests of this process left in torch's caching allocator
    free, _ = torch_cuda.cuda.mem_get_info()
    if free < (200 << 30):
        pytest.skip("needs 200 GiB of free HBM (%d GiB free)" % (free >> 30))
    out, _ = _bench(["--gpus", "8", "--single-device", "--config5-full-size", "--dist-backend", "gloo", "--perms-per-gpu", "65536",
                     "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    c5 = out["secondary"]["config5_2p30"]
    assert c5["total_perms"] == 1 << 30 and c5["perms_per_gpu"] == 1 << 27 and len(c5["kernel_ms_per_rank"]) == 8
    assert c5["digest"] == kat["config5_2p30"]["oracle_digest"]
    assert c5["digest_matches_oracle_at_full_size"] is True and c5["parity_vs_cpu_sample"] is True
    with open(os.path.join(ROOT, "gpurun_out", "bench_config5_full_size_8ranks_one_device.json"), "w") as f:
        json.dump(out, f)


def test_bench_every_output_of_every_rank_against_the_oracle_at_headline_size(torch_cuda, hades_lib, kat):
    """The headline configuration (2^26 states per GPU) with two ranks: ALL outputs of each rank's first launch -- blocks 0
    and 1 of the global index space -- against the CPU oracle's committed digests of the same states
    (kat.json headline_2p26_blocks; tools/oracle_block_digests.py), not a sample.  One rank: block 0."""
    assert all(str(g) in kat["headline_2p26_blocks"]["blocks"] for g in range(8))
    two, _ = _bench(["--gpus", "2", "--single-device", "--dist-backend", "gloo", "--steps", "1", "--warmup", "0",
                     "--no-secondary", "--no-cpu-baseline"])
    assert two["config"]["perms_per_gpu"] == 1 << 26 and two["n_gpus"] == 2
    assert two["parity_all_outputs_first_launch"] is True and "covered: 2 of 2" in two["parity_all_outputs"]
    assert two["parity_vs_cpu_sample"] is True
    one, _ = _bench(["--steps", "1", "--warmup", "0", "--no-secondary", "--no-cpu-baseline"])
    assert one["parity_all_outputs_first_launch"] is True and "covered: 1 of 1" in one["parity_all_outputs"]
    small, _ = _bench(["--steps", "1", "--warmup", "0", "--no-secondary", "--no-cpu-baseline", "--perms-per-gpu", "65536"])
    assert small["parity_all_outputs_first_launch"] is None and small["parity_vs_cpu_sample"] is True


def test_bench_refuses_two_ranks_on_one_device(torch_cuda, hades_lib):
    """Two ranks that land on the same physical device without --single-device: every rank exits 4 and no line is printed."""
    import subprocess
    # both ranks are pointed at device 0 by making LOCAL_RANK 0 for both (what a mis-configured launcher would do)
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "bench.py", "--gpus", "2", "--dist-backend", "gloo", "--perms-per-gpu", "65536",
                                       "--steps", "1", "--warmup", "0", "--no-secondary", "--no-cpu-baseline"], cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [4, 4], outs
    assert not any(l.startswith("{") for o, _ in outs for l in o.splitlines())
    assert "same physical device" in outs[0][1]


def test_bench_strong_scaling_mode_same_batch_at_every_world_size(torch_cuda, hades_lib):
    """--total-perms T (round 6): T states IN ALL whatever the world size -- one rank, two ranks and three ranks (a world
    size that does not divide T: shards differ by one state) over the same 2^21 + 1 ... states must leave the same digest
    after the same number of passes, say `"scaling": "strong"` and report T per step."""
    common = ["--steps", "2", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"]
    total = (1 << 21) + 1
    one, _ = _bench(["--total-perms", str(total)] + common)
    assert one["scaling"] == "strong" and one["n_gpus"] == 1 and one["config"]["total_perms_per_step"] == total
    assert one["config"]["perms_per_gpu"] == total and one["parity_vs_cpu_sample"] is True
    assert abs(one["value"] - total * 2 / (one["ms_per_step"] * 2e-3)) < 1e-6 * one["value"]
    for world in (2, 3):
        out, _ = _bench(["--gpus", str(world), "--single-device", "--dist-backend", "gloo", "--total-perms", str(total)] + common)
        assert out["scaling"] == "strong" and out["n_gpus"] == world and out["config"]["total_perms_per_step"] == total
        assert out["config"]["perms_per_gpu"] == -(-total // world) and len(out["per_gpu"]["kernel_ms_per_rank"]) == world
        assert out["digest"] == one["digest"] and out["parity_vs_cpu_sample"] is True
        assert "IN ALL" in out["config"]["workload"] and "configs[4]" not in out["config"]["workload"]
    # --total-perms wins over --perms-per-gpu; the weak default is untouched
    both, _ = _bench(["--total-perms", str(total), "--perms-per-gpu", "4096"] + common)
    assert both["digest"] == one["digest"] and both["scaling"] == "strong"
    weak, _ = _bench(["--perms-per-gpu", str(total)] + common)
    assert weak["scaling"] == "weak" and weak["digest"] == one["digest"]
