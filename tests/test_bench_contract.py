"""bench.py contract: one JSON line with the required keys; the N>1 control flow (barrier, max over
ranks, digest combination, global shard indices) run as two ranks -- sharing cuda:0 over gloo, because
the test box has one GPU -- produces the same output digest as the single-rank job."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly one JSON line"
    return json.loads(lines[0])


def test_bench_single_and_two_ranks_agree():
    per = 1 << 18
    one = run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--perms-per-gpu", str(2 * per),
               "--cpu-sample", "4096"])
    for k in REQUIRED + ["cpu_baseline", "secondary"]:
        assert k in one, k
    # outside `value`: BASELINE configs[3] and the host-pointer boundary (native caller, PCIe-inclusive)
    mk, hp = one["secondary"]["merkle_2p24"], one["secondary"]["host_path"]
    assert mk["nodes"] == 5592405 and 0 < mk["tree_ms"] < 100 and mk["roofline"]["algorithmic_bytes_per_node"] == 160
    assert hp["bit_exact_vs_device_path"] is True and 0 < hp["frac_of_ceiling"] < 1.3 and hp["perms"] == 1 << 22   # (slightly > 1 happens: a box whose bare copies run below the pipeline; a mis-measured ceiling does not pass)
    sp = one["secondary"]["single_perm"]
    assert 20 < sp["device_us_median"] < 150 and sp["device_us_min"] <= sp["host_call_us_median"] < 400
    assert 20 < one["secondary"]["sponge_chain"]["us_per_block"] < 150
    assert mk["root_matches_golden"] is True                          # the timed tree == the CPU oracle's committed root
    wf = one["secondary"]["wire_format"]
    assert wf["round_trip_exact"] is True
    for k in ("to_bytes", "from_bytes"):
        assert wf[k]["roofline"]["bound"] == "hbm" and 0.3 < wf[k]["roofline"]["frac"] < 1.0
    gw = one["secondary"]["gadget_witness"]                           # f4: 972 wires per state, priced against HBM
    assert gw["last_rows_equal_perm"] is True and gw["roofline"]["bound"] == "hbm" and 0.15 < gw["roofline"]["frac"] < 1.0
    assert gw["trace"]["last_round_equals_perm"] is True and 0.1 < gw["trace"]["roofline"]["frac"] < 1.0
    ts = gw["trace_scaled"]                                            # the opt-in scaled form: faster, same information
    assert ts["last_round_times_mul_equals_perm"] is True and ts["roofline"]["frac"] > gw["trace"]["roofline"]["frac"]
    # BASELINE configs[4] at N = 1 (the first point of the strong-scaling curve): all 2^30 outputs against the oracle's digest
    c5 = one["secondary"]["config5_2p30"]
    if "skipped" in c5:                                                # (a device without 162 GiB free: the record says so)
        assert "GiB" in c5["skipped"]
    else:
        assert c5["scaling"] == "strong" and c5["total_perms"] == 1 << 30 and c5["digest_matches_oracle_at_full_size"] is True
        assert c5["shard_digests_match_oracle"] is True and c5["parity_vs_cpu_sample"] is True and c5["value"] > 1e8
    cb = one["cpu_baseline"]                                           # both builds of the same source, side by side
    assert cb["builds"]["portable"]["flags"].endswith("x86-64-v3") and cb["flags"] in (cb["builds"]["portable"]["flags"], "gcc -O3 -march=native")
    assert cb["value"] == max(b["value"] for b in cb["builds"].values() if "value" in b)
    assert 0 < hp["pageable_ms"] < 3 * hp["ms"]                       # ordinary memory on fresh pages: staging threads
    assert one["config"]["kernel"] == "k_perm_fast" and "valu_issue" in one and "frac_of_measured" not in one["valu_issue"]
    assert one["n_gpus"] == 1 and one["parity_vs_cpu_sample"] is True
    assert one["roofline"]["bound"] == "hbm" and 0 < one["roofline"]["frac"] < 1
    # N = 1 evidence block: one rank seen, its physical device named; the crossover table of the one-call-per-permutation shape
    assert one["dist"]["ranks_seen"] == 1 and len(one["per_gpu"]["device"]) == 1 and "pci" in one["per_gpu"]["device"][0]
    cx = one["secondary"]["crossover"]
    assert cx["rows"][0]["n"] == 1 and cx["gpu_beats_one_core_from_n"] is not None and cx["gpu_beats_all_cores_from_n"] is not None
    assert cx["gpu_beats_one_core_from_n"] <= cx["gpu_beats_all_cores_from_n"] <= 4096
    # counter-backed traffic of the secondary kernels comes from the keyed record or is null -- never a literal
    for rl in (mk["roofline"], wf["to_bytes"]["roofline"], wf["from_bytes"]["roofline"], gw["roofline"], gw["trace"]["roofline"],
               ts["roofline"]):
        assert "traffic" in rl and (rl["traffic"] is None or 0.9 < rl["traffic_over_algorithmic"] < 1.5)
    assert one["cpu_baseline"]["kind"] == "port" and one["cpu_baseline"]["cores"] >= 1
    assert one["vs_baseline"] is None and one["scaling"] == "weak" and one["config"]["workload"]
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200),
               "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--perms-per-gpu", str(per),
               "--dist-backend", "gloo", "--single-device", "--cpu-sample", "4096", "--no-secondary"])
    assert two["n_gpus"] == 2 and two["cpu_baseline"]["kind"] == "port" and "secondary" not in two
    # same global batch (2 x per states, same generator indices), same number of passes
    assert two["digest"] == one["digest"]
    assert two["parity_vs_cpu_sample"] is True and len(two["per_gpu"]["kernel_ms_per_rank"]) == 2
    # self-launch: `python bench.py --gpus 2` with no torchrun around it spawns its own ranks
    env_clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--perms-per-gpu",
                        str(per), "--dist-backend", "gloo", "--single-device", "--cpu-sample", "4096", "--no-secondary"],
                       cwd=ROOT, capture_output=True, text=True,
                       timeout=900, env=env_clean)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    self_launched = json.loads(lines[0])
    assert self_launched["n_gpus"] == 2 and self_launched["digest"] == one["digest"]
    assert self_launched["parity_vs_cpu_sample"] is True


def test_bench_reports_the_kernel_that_ran():
    """--perms-per-gpu <= 16384 with the default dispatch runs the five-waves kernel: the line must say so and must not
    attribute k_perm_fast's instruction counts / traffic profile to it."""
    small = run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--perms-per-gpu", "8192", "--cpu-sample",
                 "4096", "--no-secondary"])
    assert small["config"]["kernel"] == "k_perm_coop" and small["roofline"]["kernel"] == "k_perm_coop"
    assert "valu_issue" not in small and small["roofline"]["traffic"] is None and small["parity_vs_cpu_sample"] is True
    for per, name in ((512, "k_perm_lanes"), (3000, "k_perm_rows")):
        tiny = run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--perms-per-gpu", str(per), "--cpu-sample",
                    "4096", "--no-secondary"])
        assert tiny["config"]["kernel"] == name and tiny["parity_vs_cpu_sample"] is True
    mk = run([sys.executable, "bench.py", "--workload", "merkle", "--steps", "3"])
    assert mk["unit"] == "nodes/s" and mk["roofline"]["algorithmic_bytes_per_node"] == 160 and mk["value"] > 1e8


def test_bench_bookkeeping_over_rccl_at_world_size_one():
    """The RCCL path of the bookkeeping collectives (barrier with device_ids, all_reduce MAX / MIN / SUM and all_gather on GPU
    tensors: hades252_amd/sharding.py) has only ever run over gloo in the 2-rank tests; a one-GPU box can still run it for
    real at world size 1: bench.py under a torchrun-style environment with backend nccl (= RCCL)."""
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
           "MASTER_PORT": str(29900 + os.getpid() % 90), "HADES252_BENCH_FORCE_DIST": "1"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        one = run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--perms-per-gpu", str(1 << 18),
                   "--cpu-sample", "4096", "--no-secondary", "--dist-backend", "nccl"])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert one["n_gpus"] == 1 and one["parity_vs_cpu_sample"] is True and len(one["per_gpu"]["kernel_ms_per_rank"]) == 1
    assert len(one["digest"]) == 4 and one["value"] > 0
