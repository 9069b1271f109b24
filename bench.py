#!/usr/bin/env python3
"""Headline benchmark: Hades252 permutations/sec on N MI355X (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--perms-per-gpu P]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path (the batched `ScalarStrategy::perm`, through the C ABI
`hades252_perm_batch_dev`) over one synthetic batch already resident in HBM: 2^26 independent
width-5 permutations per GPU (BASELINE.json configs[2]; 10 GiB in place), generated on device by
the counter-based generator B.  With N GPUs every rank owns its own batch of the SAME size (weak
scaling, global element indices are disjoint; `--perms-per-gpu 134217728` gives BASELINE.json
configs[4], 2^30 over 8 GPUs); there is no data-path collective.  Rank 0 prints ONE JSON line.
Before the warm-up every rank runs ONE untimed parity launch and compares the digest of ALL its 2^26 outputs with the CPU
oracle's committed digest of the same states (`parity_all_outputs_first_launch`; blocks 0 .. 7 = N up to 8).

Launching: under torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process is one
rank.  Without it, `--gpus N` with N > 1 makes this process a LAUNCHER: it spawns N rank processes
(one per GPU, rendezvous on 127.0.0.1) before anything touches the GPU, waits for them, and exits
with their status -- the ranks do the work and rank 0 prints the line.

Also in the line:
  roofline      the dominant kernel (k_perm_fast) against the HBM roofline the metric names:
                algorithmic bytes = 320 B per permutation (160 B state in + 160 B out)
                x permutations per launch / average launch duration, measured live with HIP
                events on the launch stream.  `traffic` is the HBM byte count of one launch from
                the committed rocprofv3 PMC run (profiles/), or null.
  cpu_baseline  the CPU oracle (oracle/hades_oracle.c, a port: the Rust reference cannot be
                built in this image) timed on this host (rank 0) on a bounded sample of the same
                workload -- two builds of the same source side by side (-march=x86-64-v3, the checker, and
                -O3 -march=native compiled on this host), `value` = the better; the same sample is used
                to check the GPU output bit for bit.
  secondary     outside `value`: BASELINE.json configs[3] (arity-4 Merkle tree over 2^24 leaves: tree
                time, nodes/s, its own roofline with 160 B per node), `single_perm` (ONE permutation: device
                and host-call latency), `sponge_chain` (one message of 1000 blocks: us per dependent
                permutation), `wire_format` (to_bytes / from_bytes at 2^26 scalars against the HBM roofline), and
                `host_path`: the entry point a
                Rust `Strategy::perm` binds (`hades252_perm_batch`: host memory in, host memory out,
                PCIe-inclusive) on 2^22 states against this box's measured bidirectional copy ceiling.
  dist          N > 1 evidence: backend, ranks_seen (= the process group's world size), the physical device every rank
                sat on (PCI address / UUID, gathered) -- two ranks on one device fail the job unless --single-device.
                Every rank (at EVERY N, 1 included) also takes part in `secondary.config5_2p30`: BASELINE configs[4] as a
                strong-scaling measurement -- 2^30 states in all, 2^30 / N per GPU, 3 timed launches, oracle samples on every
                rank, every rank's digest of ALL its outputs against the oracle's committed shard digests and their sum against
                the oracle's digest of the whole batch (`config5_rehearsal` with --single-device: the same code on
                --perms-per-gpu states per rank) -- and, with N > 1, in `merkle_2p24_sharded` (the 2^24-leaf tree sharded by
                sub-tree: local sub-roots, ONE all_gather of 32-byte sub-roots -- the path's only exchange step -- and the top
                levels; root against the oracle's committed root).
  crossover     under `secondary`: the smallest batch for which one hades252_perm_batch call beats the CPU port
                (one core / all cores) -- the reference's own call shape is ONE permutation per call (README.md:60-61).
`--workload merkle` times the tree build itself as the step (development; the driver runs the default).
`--total-perms T` (strong scaling): T permutations IN ALL, rank g owns the contiguous range [g T / N, (g + 1) T / N);
`--total-perms 1073741824` is BASELINE configs[4] at any N (one device holds all 160 GiB at N = 1), the line then says
`"scaling": "strong"` and every rank's first launch is checked against the CPU oracle's digest of ALL 2^30 outputs.  The
default (weak) run carries the same measurement at every N -- N = 1 included -- as `secondary.config5_2p30`, so that the
driver's 1 / 2 / 4 / 8 runs give a strong-scaling curve for configs[4] beside the weak-scaling headline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PERM = 320
HBM_PEAK_GBS = 8000.0              # MI355X HBM3E peak, MI355X_MICROARCH.md
# The binding resource is VALU issue.  The 64-bit operations of k_perm_fast per permutation follow from its structure
# (DESIGN.md section 4.1); the TOTAL per wave is rocprofv3's SQ_INSTS_VALU / SQ_WAVES, read from the keyed profile record.
MADS_PER_PERM = 99 * 387 + 59 * 89 + 5 * 97 + 67 * 265       # 64-bit multiply-adds: S-boxes, K_r, FINAL_F, linear layers
SHIFTS_PER_PERM = 17 * 297 + 9 * 59 + 10 * 5 + 45 * 67       # 64-bit arithmetic shifts, one per column: same issue class
OPS64_PER_PERM = MADS_PER_PERM + SHIFTS_PER_PERM
# (32-bit operations = `valu_instructions_per_wave` of the keyed profile record - OPS64_PER_PERM: never a literal here)
N_SIMD = 1024
PEAK_CLOCK_HZ = 2.4e9
# Issue model: a wave64 instruction occupies its SIMD-16 for 4 cycles when 64-bit, 2 cycles when 32-bit
# (MI355X_MICROARCH.md).  IDEAL peak = one 64-bit op per SIMD per 4 cycles at the 2.4 GHz peak clock.
VALU_IDEAL_G_WI = N_SIMD * PEAK_CLOCK_HZ / 4 / 1e9  # 614.4 G 64-bit wave-instr/s
MERKLE_BYTES_PER_NODE = 160        # 4 x 32 B children in + 32 B digest out
CPU_SINGLE_THREAD_SECONDS = 2.5    # every CPU leg >= 2 s


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 256))


def cpu_baseline_and_check(H, torch, device, n_sample: int, kernel: int):
    """Time the CPU oracle on the first n_sample permutations of the workload and use its output to check the GPU path.
    The oracle is used here only as baseline + checker.  Two builds of the SAME source are timed: the portable one the
    tests check against (-march=x86-64-v3: it travels between machines) and one compiled on THIS host with -O3
    -march=native -- `value` is the better of the two, so the baseline is never handicapped by
    the flags; the GPU output is compared with the portable build's, and the native build's output with it too."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    orc = oracle_lib.load()
    cores = usable_cores()
    inp = orc.gen_b(0, 5 * n_sample)

    def time_build(o):
        # single thread on a slice sized for >= 2 s (calibrated on 2 048 permutations), all cores on the whole sample
        t0 = time.perf_counter()
        o.perm_batch(inp[:20 * 2048], 1)
        rate = 2048 / (time.perf_counter() - t0)
        n1 = int(min(n_sample, max(4096, rate * CPU_SINGLE_THREAD_SECONDS)))
        t0 = time.perf_counter()
        o.perm_batch(inp[:20 * n1], 1)
        t1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        res = o.perm_batch(inp, cores)
        return res, time.perf_counter() - t0, n1, t1

    exp, tall, n1, t1 = time_build(orc)
    builds = {"portable": {"flags": oracle_lib.PORTABLE_FLAGS, "value": n_sample / tall, "single_thread_value": n1 / t1}}
    native, native_flags = oracle_lib.load_native()
    native_ok = True
    if native is not None:
        exp_n, tall_n, n1_n, t1_n = time_build(native)
        native_ok = bool((exp_n == exp).all())
        builds["native"] = {"flags": native_flags, "value": n_sample / tall_n, "single_thread_value": n1_n / t1_n,
                            "equals_portable_build": native_ok}
    else:
        builds["native"] = {"flags": oracle_lib.NATIVE_FLAGS, "unavailable": native_flags}
    best = max((b for b in builds.values() if "value" in b), key=lambda b: b["value"])
    best_1t = max((b for b in builds.values() if "value" in b), key=lambda b: b["single_thread_value"])
    # parity of the GPU path on the same inputs
    buf = H.gen_b(5 * n_sample, device)
    H.ScalarStrategy(kernel).perm(buf)
    got = buf.cpu().numpy().view(np.uint64).reshape(-1)
    ok = bool((got == exp).all()) and native_ok
    return {
        "value": best["value"], "unit": "permutations/s", "cores": cores, "kind": "port",
        "flags": best["flags"], "host_cpu": oracle_lib.host_cpu_model(),
        "sample": "first %d permutations of the same generator-B workload, %d threads (C restatement of the reference CPU "
                  "path; the better of two builds of the same source, see `builds`)" % (n_sample, cores),
        "single_thread_value": best_1t["single_thread_value"], "single_thread_flags": best_1t["flags"],
        "single_thread_sample": "first %d permutations of the same workload, one thread, >= %.1f s" % (n1, CPU_SINGLE_THREAD_SECONDS),
        "all_cores_seconds": n_sample / best["value"],
        "builds": builds,
    }, ok


class ShardCheck:
    """Bit-exact check of THIS rank's shard against the oracle, on the very buffer and by the very launches that
    are timed: k sampled states (first k/2 + k/2 strided) are read before the first step (inputs: they must be what
    generator B defines) and again after the last one, and must then equal the oracle's permutation applied as many
    times as the kernel was launched (warm-up + timed steps).  The oracle is the checker only."""

    def __init__(self, torch, states, first_perm: int, n: int, k: int):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as np
        import oracle_lib
        self.np, self.orc, self.states = np, oracle_lib.load(), states
        k = max(2, min(k, n))
        idx = np.unique(np.concatenate([np.arange(k // 2), np.arange(0, n, max(1, n // (k // 2)))[: k // 2]]))
        self.tidx = torch.from_numpy(idx).to(states.device)
        self.inp = states[self.tidx].contiguous().cpu().numpy().view(np.uint64).reshape(-1).copy()
        e0 = 5 * (first_perm + int(idx[-1]))
        self.inputs_ok = bool((self.inp[-20:] == self.orc.gen_b(e0, 5)).all())

    def after(self, n_applications: int) -> bool:
        got = self.states[self.tidx].contiguous().cpu().numpy().view(self.np.uint64).reshape(-1)
        exp = self.inp
        for _ in range(n_applications):
            exp = self.orc.perm_batch(exp, min(usable_cores(), 8))
        return self.inputs_ok and bool((got == exp).all())


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without torchrun: spawn the N ranks.  This process never touches the GPU
    (a process that has initialised HIP must not be replaced or forked into GPU work on this pool)."""
    import socket
    import subprocess
    from hades252_amd import build
    build.build(verbose=False)          # once, so the ranks do not race to compile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.build()                  # the checker, likewise
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll ALL ranks: if one dies (before or inside a collective) the others would wait for it until the backend's
    # timeout -- end them instead and report the first failure
    rc = 0
    try:
        live = list(procs)
        deadline = None                   # set by the first failure: the others get a few seconds to fail by themselves
        while live and (deadline is None or time.time() < deadline):
            time.sleep(0.05)
            for p in list(live):
                if p.poll() is not None:
                    live.remove(p)
                    rc = rc or p.returncode
            if rc != 0 and deadline is None:
                deadline = time.time() + 5.0
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def kernel_of(kernel_arg: int, n: int) -> str:
    """The kernel `hades252_perm_batch_dev_ex(.., kernel)` launches for n states: asked of the library, whose dispatch
    rule lives in one place (`hades252_kernel_name` / `hades252_kernel_for`, include/hades252.h)."""
    from hades252_amd import strategy as H
    return H.kernel_name(kernel_arg, n)


def golden_merkle_root(n_leaves: int):
    """The CPU oracle's committed root of the arity-4 tree over the first n_leaves generator-B leaves (tag 15, word 1):
    tests/golden/kat.json `merkle4_full_size`, written by tools/gen_constants.py.  None if that size is not pinned."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
            rec = json.load(f)["merkle4_full_size"].get(str(n_leaves))
        return int(rec["root"], 16) if rec else None
    except Exception:
        return None


def golden_range_digest(first_perm: int, n: int):
    """The CPU oracle's digest (hades252_digest_dev / oracle_lib.digest_ref, GLOBAL word indices) of ALL outputs of
    perm(generator-B states [first_perm, first_perm + n)), from the committed oracle runs -- or None when the range is not
    a union of committed pieces.  Digests are additive over disjoint index ranges (limb-wise wrapping sums), and two
    independent oracle runs are committed in tests/golden/kat.json: `headline_2p26_blocks` (2^26-state blocks 0 .. 7 =
    states [0, 2^29): what rank g of a weak-scaling run holds after its first launch; tools/oracle_block_digests.py) and
    `config5_2p30.oracle_shard_digests` (2^27-state shards 0 .. 7 = all of configs[4]; tools/oracle_config5_digest.py)."""
    if n <= 0:
        return None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
            kat = json.load(f)
        blocks = kat["headline_2p26_blocks"]["blocks"]
        tables = (((1 << 26), [blocks.get(str(i)) for i in range(8)]),
                  ((1 << 27), list(kat["config5_2p30"]["oracle_shard_digests"])))
    except Exception:
        return None
    for piece, table in tables:
        if first_perm % piece or n % piece:
            continue
        ids = range(first_perm // piece, (first_perm + n) // piece)
        if ids[-1] >= len(table) or any(table[i] is None for i in ids):
            continue
        acc = [0, 0, 0, 0]
        for i in ids:
            acc = [(a + int(h, 16)) & 0xFFFFFFFFFFFFFFFF for a, h in zip(acc, table[i])]
        return ["%016x" % a for a in acc]
    return None


def merkle_record(H, torch, device, log_leaves: int, reps: int = 5):
    """BASELINE.json configs[3]: arity-4 Poseidon Merkle tree over 2^log_leaves leaves resident in HBM (generator B,
    tag 15, digest = word 1 -- the external convention, a parameter), root only.  HIP events around each build."""
    from hades252_amd import _lib
    n = 1 << log_leaves
    p_mod = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    tag = 15 * ((1 << 256) % p_mod) % p_mod
    leaves = H.gen_b(n, device)
    scratch = torch.empty(max(_lib.lib().hades252_merkle_scratch_bytes(n, 4) // 8, 2), dtype=torch.int64, device=device)
    H.merkle_root(leaves, 4, tag, 1, scratch)
    builds = 1                                      # (counted, so that a counter pass can divide its sums by it)
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        root = H.merkle_root(leaves, 4, tag, 1, scratch)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
        builds += 1
    med = sorted(ms)[len(ms) // 2]
    nodes = (n - 1) // 3
    ach = MERKLE_BYTES_PER_NODE * nodes / (med * 1e-3) / 1e9
    root_hex = "".join("%016x" % (int(v) & 0xFFFFFFFFFFFFFFFF) for v in reversed(root.cpu().tolist()))
    gold = golden_merkle_root(n)
    return {"workload": "arity-4 Poseidon Merkle tree over 2^%d leaves in HBM, root only (BASELINE configs[3]; tag 15, "
                        "digest word 1: external convention, parameters)" % log_leaves,
            "tree_ms": med, "tree_ms_all": ms, "trees_built": builds, "nodes": nodes, "nodes_per_s": nodes / (med * 1e-3),
            "root": root_hex,
            # the root of the LAST timed build against the CPU oracle's committed root of the same tree
            "root_matches_golden": None if gold is None else int(root_hex, 16) == gold,
            "golden": "tests/golden/kat.json merkle4_full_size (C oracle, %d permutations)" % nodes,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_node": MERKLE_BYTES_PER_NODE}}, leaves


def wire_format_record(H, torch, device, log_n=26):
    """BlsScalar::to_bytes / from_bytes on the device, 2^log_n scalars (2 GiB in + 2 GiB out at
    26: far beyond the 256 MB Infinity Cache), HIP events, median of 5.  Pure HBM streaming kernels: their roofline IS the
    HBM one, 64 algorithmic bytes per scalar (counter traffic = algorithmic x 1.000: profiles/r4/wire_bw_last_session.txt)."""
    n = 1 << log_n
    limbs = H.gen_b(n, device)
    out = torch.empty_like(limbs)
    canon = H.to_bytes(limbs)
    launches = {"to_bytes": 1, "from_bytes": 0}     # (counted, so that a counter pass can divide its sums by it)
    rec = {"workload": "2^%d scalars per launch, 32 B in + 32 B out each" % log_n}
    for name, fn in (("to_bytes", lambda: H.to_bytes(limbs, out)), ("from_bytes", lambda: H.from_bytes(canon, out))):
        fn()
        launches[name] += 1
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
            launches[name] += 1
        med = sorted(ms)[2]
        ach = 64.0 * n / (med * 1e-3) / 1e9
        rec[name] = {"ms": med, "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_scalar": 64}}
    rec["round_trip_exact"] = bool(torch.equal(out, limbs))          # from_bytes(to_bytes(x)) == x on all 2^log_n scalars
    rec["launches"] = launches
    return rec


def gadget_witness_record(H, torch, device, log_n=20):
    """every gate output GadgetStrategy assigns (972 per permutation,
    src/strategies/gadget.rs:41-133) for 2^log_n states -- 31 104 bytes written per state, so the kernel is priced against
    the HBM roofline.  HIP events, median of 5; the last round's rows must equal the permutation of the same states."""
    n = 1 << log_n
    st = H.gen_b(5 * n, device)
    wires = torch.empty((H.witness_wires(), n, 4), dtype=torch.int64, device=device)
    H.perm_witness(st, out=wires)
    launches = 1
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        H.perm_witness(st, out=wires)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
        launches += 1
    med = sorted(ms)[2]
    nw = wires.shape[0]
    last = torch.stack([wires[nw - 10 + 2 * j + 1] for j in range(5)], dim=1).reshape(-1)
    out = st.clone()
    H.ScalarStrategy().perm(out)
    ach = (160.0 + 32.0 * nw) * n / (med * 1e-3) / 1e9
    rec = {"workload": "2^%d states, %d wires of 32 B each per state" % (log_n, nw), "ms": med, "launches": launches,
           "perms_per_s": n / (med * 1e-3), "last_rows_equal_perm": bool(torch.equal(last, out.reshape(-1))),
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_state": 160 + 32 * nw}}
    del wires
    # ... and the per-round trace (the state after each of the 67 rounds, 160 B each), same states
    trace = torch.empty((67, n, 5, 4), dtype=torch.int64, device=device)
    stv = st.view(n, 5, 4)
    H.perm_trace(stv, out=trace)
    launches = 1
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        H.perm_trace(stv, out=trace)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
        launches += 1
    med = sorted(ms)[2]
    ach = 160.0 * 68 * n / (med * 1e-3) / 1e9
    rec["trace"] = {"workload": "2^%d states, 67 states of 160 B written per state" % log_n, "ms": med, "launches": launches,
                    "perms_per_s": n / (med * 1e-3), "last_round_equals_perm": bool(torch.equal(trace[66].reshape(-1), out.reshape(-1))),
                    "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_state": 160 * 68}}
    # ... and the same trace in SCALED form (opt-in: the consumer applies one multiplier per round + the deferred constants
    # lazily): the rounds of the throughput kernel, every word leaves through `finalize` alone
    true_last = trace[66].clone()
    H.perm_trace_scaled(stv, out=trace)
    launches = 1
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        H.perm_trace_scaled(stv, out=trace)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
        launches += 1
    med = sorted(ms)[2]
    ach = 160.0 * 68 * n / (med * 1e-3) / 1e9
    mul, add = H.trace_scale_table()                                 # un-scale the last round with the library's own field ops
    import numpy as np
    m66 = torch.from_numpy(np.tile(mul[66], 5 * n).view(np.int64)).to(device).view(-1, 4)
    last = H.fr_op(H.FR_MUL, trace[66].reshape(-1, 4).contiguous(), m66)
    rec["trace_scaled"] = {"workload": "2^%d states, 67 states of 160 B written per state, SCALED form (true = scaled * mul[r] "
                                       "+ add[r][w], hades252_perm_trace_scale_table)" % log_n, "ms": med, "launches": launches,
                           "perms_per_s": n / (med * 1e-3),
                           "last_round_times_mul_equals_perm": bool(torch.equal(last.reshape(-1), true_last.reshape(-1))),
                           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_state": 160 * 68}}
    return rec


def sponge_chain_record(H, torch, device, blocks=1000):
    """The reference's consumer (dusk-poseidon's sponge, README.md:9) at its hardest shape for a GPU: ONE message, a chain
    of `blocks` dependent permutations.  Whole call (HIP events), microseconds per block; the CPU port's time for one
    permutation on one thread is what a CPU core would need per block."""
    pool = H.gen_b(4 * blocks, device)
    off = torch.zeros(1, dtype=torch.int64, device=device)
    ln = torch.full((1,), 4 * blocks - 1, dtype=torch.int64, device=device)      # + the padding scalar = `blocks` blocks
    cap = 1                                                                      # any capacity word: timing only
    H.sponge_hash_var(pool, off, ln, cap, 1)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        H.sponge_hash_var(pool, off, ln, cap, 1)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    return {"workload": "sponge hash of ONE message of %d blocks (rate 4): %d dependent permutations, one wave" % (blocks, blocks),
            "ms": ms, "us_per_block": ms * 1e3 / blocks}


def single_perm_record(H, torch, device):
    """The reference's own call shape: ONE permutation (README.md:60-61).  Device-side: HIP events around
    hades252_perm_batch_dev on one resident state (default dispatch = the lane-split kernel; the events include the launch
    gap of a few microseconds -- profiles/r3/ holds the rocprofv3 kernel durations); host call: hades252_perm_batch on 20
    limbs of ordinary memory, wall clock.  The CPU port's time for one permutation on one thread stands beside them."""
    import numpy as np
    strat = H.ScalarStrategy()
    buf = H.gen_b(5, device)
    host = buf.cpu().numpy().view(np.uint64).reshape(-1).copy()
    for _ in range(20):
        strat.perm(buf)
    torch.cuda.synchronize()
    ev = []
    for _ in range(101):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        strat.perm(buf)
        b.record()
        torch.cuda.synchronize()
        ev.append(a.elapsed_time(b) * 1e3)
    for _ in range(20):
        strat.perm(host)
    ts = []
    for _ in range(101):
        t0 = time.perf_counter()
        strat.perm(host)
        ts.append((time.perf_counter() - t0) * 1e6)
    return {"workload": "one WIDTH=5 permutation (BASELINE configs[0] shape), default dispatch: k_perm_lanes",
            "device_us_median": sorted(ev)[50], "device_us_min": min(ev),
            "host_call_us_median": sorted(ts)[50], "host_call_us_min": min(ts)}


def host_path_record(log_n: int = 22):
    """The boundary a Rust `Strategy::perm` binds: `hades252_perm_batch` on host memory (PCIe-inclusive; never
    `value`), measured by a NATIVE caller -- tools/host_path_bench.cpp, a plain C++ program linked against the
    system HIP runtime like a Rust / C host, started here as a child process.  (Inside this PyTorch process the
    library runs on PyTorch's own bundled HIP runtime, under which the same copy / kernel pipeline overlaps far worse:
    profiles/r3/host_path.txt.)  The ceiling is measured by that same program in the same run."""
    import subprocess
    from hades252_amd import build
    exe = build.build_host_path_bench(verbose=False)
    res = subprocess.run([exe, str(log_n)], capture_output=True, text=True, timeout=300)
    if res.returncode != 0:
        raise RuntimeError("host_path_bench failed: " + res.stderr[-500:])
    rec = json.loads(res.stdout.strip().splitlines()[-1])
    rec["entry_point"] = ("hades252_perm_batch on page-locked host memory from hades252_host_alloc: chunks copied in, "
                          "permuted and copied out on three streams chained by events (native C++ caller, system HIP "
                          "runtime)")
    rec["ceiling"] = ("hipMemcpyAsync of the same bytes in both directions at once (10 / 20 / 40 MiB pieces, both kinds of "
                      "page-locked memory, best of 3 stream pairs each), measured by the same process in this run")
    return rec


def profile_record(build, kernel_name: str, n: int):
    """The committed rocprofv3 PMC record of the dominant kernel (profiles/hbm_traffic.json, written by
    tools/summarize_profile.py on the GPU box and stamped with the commit by tools/collect_profiles.sh) -- only while it is
    keyed to the very kernel this process runs: same sources + tables + flags (`build.perm_fast_hash()`), same launch
    size.  Returns (record or None, device-record or None): the second is the per-kernel table of the other kernels
    (`secondary_kernels`), keyed by `build.device_source_hash()`."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(tpath))
    except Exception:
        return None, None
    main_rec = None
    if (kernel_name == "k_perm_fast" and rec.get("kernel") == "k_perm_fast"
            and rec.get("kernel_source_hash") == build.perm_fast_hash() and "hash_note" not in rec):
        main_rec = dict(rec)
        if rec.get("perms_per_launch") != n:          # instructions per wave do not depend on the launch size, bytes do
            main_rec["hbm_bytes_per_launch"] = None
    sec = rec.get("secondary_kernels")
    if not (isinstance(sec, dict) and sec.get("device_source_hash") == build.device_source_hash()):
        sec = None
    return main_rec, sec


def attach_traffic(roofline: dict, sec, key: str, algorithmic: float):
    """Counter-backed HBM bytes of a secondary kernel (per launch / per tree, like `achieved`), replayed from the keyed
    record; null when the record does not belong to this build."""
    ent = (sec or {}).get(key)
    if ent and ent.get("hbm_bytes"):
        roofline["traffic"] = ent["hbm_bytes"]
        roofline["traffic_over_algorithmic"] = ent["hbm_bytes"] / algorithmic
        roofline["traffic_source"] = ("NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of commit %s "
                                      "(profiles/hbm_traffic.json secondary_kernels.%s), same device sources"
                                      % (str(sec.get("measured_at_commit"))[:12], key))
    else:
        roofline["traffic"] = None


def ranks_agree(sharding, ok: bool, device) -> bool:
    """AND over ranks of a LOCAL outcome, taken before a record's first collective: either every rank goes on or every rank
    skips -- a rank that failed alone (out of memory on one GPU) must not leave the others waiting in a barrier."""
    return sharding.reduce_min_int(1 if ok else 0, device) == 1


def config5_record(args, H, torch, device, sharding, rank, world):
    """BASELINE configs[4] on the ranks of this job, at EVERY world size (every rank calls this): 2^30 states in all, rank g
    owns [g 2^30 / N, (g + 1) 2^30 / N) -- 160 GiB on the one device at N = 1, 20 GiB per GPU at N = 8 -- so the driver's
    1 / 2 / 4 / 8 runs give a STRONG-scaling curve for the config BASELINE names.  3 timed launches; 2 048 states of every
    rank's shard against the CPU oracle; every rank's digest of ALL its outputs after the FIRST launch against the sum of
    the oracle's committed shard digests it covers, and the wrapping sum over ranks against the oracle's digest of all
    2^30 outputs (tests/golden/kat.json `config5_2p30`).  With --single-device (the one-GPU rehearsal of the multi-rank
    path) the shard is --perms-per-gpu states and rank 0 computes the one-device digest itself."""
    full = (not args.single_device) or args.config5_full_size
    total = (1 << 30) if full else (args.perms_per_gpu or (1 << 20)) * world
    first_perm, end = sharding.strong_shard(rank, world, total)
    n = end - first_perm
    st, check, err = None, None, None
    try:                                             # local work only: nothing here may leave another rank waiting
        free, _ = torch.cuda.mem_get_info()
        if free < 160 * n + (2 << 30):
            raise MemoryError("%d GiB of HBM free, the shard needs %d" % (free >> 30, (160 * n + (2 << 30)) >> 30))
        st = torch.empty((n, 5, 4), dtype=torch.int64, device=device)
        H.gen_b(5 * n, device, first_elem=5 * first_perm, out=st.view(-1, 4))
        check = ShardCheck(torch, st, first_perm, n, args.verify_sample)
        torch.cuda.synchronize()
    except Exception as e:
        err = repr(e)
    if not ranks_agree(sharding, err is None, device):
        del st, check
        torch.cuda.empty_cache()
        return {"skipped": "a rank could not set its shard up (rank 0: %s)" % err} if rank == 0 else None
    strat = H.ScalarStrategy(args.kernel)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    digest1 = None
    sharding.barrier(device)
    t0 = time.perf_counter()
    for i, (a, b) in enumerate(evs):
        a.record()
        strat.perm(st)
        b.record()
        if i == 0:
            digest1 = H.digest(st, first_index=20 * first_perm)       # (outside the events, inside the wall clock)
    torch.cuda.synchronize()
    sharding.barrier(device)
    wall = sharding.reduce_max(time.perf_counter() - t0, device)
    ms = sharding.gather_floats(sum(a.elapsed_time(b) for a, b in evs) / 3, device)
    ok = sharding.reduce_min_int(1 if check.after(3) else 0, device) == 1
    # this rank's outputs, ALL of them, against the oracle's committed digests of the same index range
    mine = ["%016x" % (d & 0xFFFFFFFFFFFFFFFF) for d in digest1]
    gold_mine = golden_range_digest(first_perm, n) if total == 1 << 30 else None
    shards_covered = sharding.reduce_sum_int(1 if gold_mine is not None else 0, device)
    shards_ok = sharding.reduce_sum_int(1 if (gold_mine is not None and mine == gold_mine) else 0, device)
    combined = ["%016x" % d for d in sharding.combine_digests(digest1, device)]
    del st, check
    torch.cuda.empty_cache()
    gold, gold_src = None, None
    if total == 1 << 30:
        try:
            with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
                gold = json.load(f)["config5_2p30"]["oracle_digest"]
            gold_src = ("tests/golden/kat.json config5_2p30: the C ORACLE's digest of all 2^30 outputs (= the digest of the whole "
                        "batch permuted on ONE device, recorded since round 2)")
        except Exception:
            gold = None
    elif total <= 1 << 24 and rank == 0:
        whole = H.gen_b(5 * total, device)
        H.ScalarStrategy(args.kernel).perm(whole)
        gold = ["%016x" % (d & 0xFFFFFFFFFFFFFFFF) for d in H.digest(whole, first_index=0)]
        gold_src = "the same %d states permuted as ONE batch on rank 0's device in this run" % total
        del whole
        torch.cuda.empty_cache()
    if rank != 0:
        return None
    n_max = -(-total // world)
    return {"workload": "%s states in all over %d GPU(s), %d per GPU (BASELINE configs[4]%s), generator B, in place, 3 timed "
                        "launches" % (("2^%d" % (total.bit_length() - 1)) if total & (total - 1) == 0 else str(total), world,
                                      n_max, "" if total == 1 << 30 else " at rehearsal size"),
            "scaling": "strong", "perms_per_gpu": n_max, "total_perms": total, "kernel": kernel_of(args.kernel, n_max),
            "kernel_ms_per_rank": ms, "perms_per_s_per_rank": [n_max / (x * 1e-3) for x in ms],
            "value": total / (max(ms) * 1e-3), "unit": "permutations/s (whole job, slowest rank's mean launch)",
            "wall_s_3_launches_and_digest": wall,
            "parity_vs_cpu_sample": ok, "digest": combined,
            "digest_matches_one_device": None if gold is None else combined == list(gold),
            # at full size the committed digest is the CPU oracle's over all 2^30 outputs: every output bit-exact, not a sample
            "digest_matches_oracle_at_full_size": (combined == list(gold)) if (gold is not None and total == 1 << 30) else None,
            # ... and rank by rank (a rank's range is a union of the oracle's eight 2^27-state shards when N divides 8)
            "shard_digests_match_oracle": None if shards_covered == 0 else shards_ok == shards_covered,
            "shards_covered": "%d of %d ranks" % (shards_covered, world),
            "golden": gold_src}


def merkle_sharded_record(H, torch, device, sharding, rank, world, log_leaves=24, reps=5):
    """BASELINE configs[3] sharded by sub-tree (every rank calls this): rank g generates the leaves
    [g n / W, (g + 1) n / W) of the 2^24-leaf tree, builds the roots of its whole sub-trees (no communication), ONE
    all_gather moves the 32-byte sub-roots -- the path's only exchange step -- and every rank hashes the top levels.
    tree_ms = median over `reps` of the max over ranks of (barrier -> root in hand), the all_gather included."""
    from hades252_amd import merkle
    n = 1 << log_leaves
    p_mod = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    tag = 15 * ((1 << 256) % p_mod) % p_mod
    per_sub, subs = merkle.subtree_split(n, world)
    shard = H.gen_b(n // world, device, first_elem=rank * (n // world))
    root = merkle.merkle4_root_sharded(shard, n, tag, 1)                      # warm-up (code object, communicator)
    ms = []
    for _ in range(reps):
        torch.cuda.synchronize()
        sharding.barrier(device)
        t0 = time.perf_counter()
        root = merkle.merkle4_root_sharded(shard, n, tag, 1)
        torch.cuda.synchronize()
        ms.append(sharding.reduce_max((time.perf_counter() - t0) * 1e3, device))
    root_hex = "".join("%016x" % (int(v) & 0xFFFFFFFFFFFFFFFF) for v in reversed(root.cpu().tolist()))
    gold = golden_merkle_root(n)
    mine_ok = 1 if (gold is None or int(root_hex, 16) == gold) else 0
    all_ok = sharding.reduce_min_int(mine_ok, device) == 1                  # every rank holds the root: all must agree
    del shard
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    med = sorted(ms)[len(ms) // 2]
    nodes = (n - 1) // 3
    ach = MERKLE_BYTES_PER_NODE * nodes / (med * 1e-3) / 1e9
    return {"workload": "arity-4 Poseidon Merkle tree over 2^%d leaves sharded over %d ranks: %d sub-tree(s) of %d leaves "
                        "per rank, one all_gather of %d sub-roots (32 B each), top levels on every rank"
                        % (log_leaves, world, subs, per_sub, subs * world),
            "exchange": "all_gather, %s backend, %d bytes in all" % (sharding.backend_name(), 32 * subs * world),
            "tree_ms": med, "tree_ms_all": ms, "nodes": nodes, "nodes_per_s": nodes / (med * 1e-3), "root": root_hex,
            "root_matches_golden": None if gold is None else all_ok,
            "golden": "tests/golden/kat.json merkle4_full_size (C oracle, %d permutations)" % nodes,
            "timing": "wall clock from a barrier to the root on every rank (device synchronised), max over ranks, median "
                      "of %d; includes the all_gather and, on one shared device (--single-device), the ranks' serialisation"
                      % reps,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_node": MERKLE_BYTES_PER_NODE,
                         "note": "whole-job figure against ONE device's peak x %d" % world}}


def crossover_record(H, cb):
    """The reference's real call shape is ONE permutation per call (README.md:60-61).  From what this run measured: the time
    of one hades252_perm_batch call on n states in ordinary host memory (n = 1 .. 4 096, median of 21 calls), against n
    permutations on one core of the CPU port and against n spread over all its cores; the crossover is the smallest n from
    which the GPU call wins for every larger n measured."""
    import numpy as np
    strat = H.ScalarStrategy()
    one_core = 1e6 / cb["single_thread_value"]                     # us per permutation, one thread
    all_cores = 1e6 / cb["value"]                                  # us per permutation, all threads busy
    sizes = [1, 2, 3, 4, 8, 12, 16, 17, 20, 24, 32, 48, 64, 128, 256, 512, 1024, 4096]
    rows = []
    rng = np.random.default_rng(5)
    for n in sizes:
        host = rng.integers(0, 1 << 62, size=20 * n, dtype=np.uint64)
        for _ in range(3):
            strat.perm(host)
        ts = []
        for _ in range(21):
            t0 = time.perf_counter()
            strat.perm(host)
            ts.append((time.perf_counter() - t0) * 1e6)
        rows.append({"n": n, "gpu_call_us": sorted(ts)[10], "cpu_one_core_us": n * one_core,
                     "cpu_all_cores_us": max(one_core, n * all_cores)})

    def first_win(key):
        win = None
        for r in reversed(rows):
            if r["gpu_call_us"] < r[key]:
                win = r["n"]
            else:
                break
        return win
    return {"workload": "one hades252_perm_batch call on n states in ordinary host memory (PCIe-inclusive), median of 21",
            "cpu": "the C port of the reference CPU path: %.1f us per permutation on one core, %.2f us with all %d cores busy "
                   "(a batch smaller than the core count costs one permutation's time)" % (one_core, all_cores, cb["cores"]),
            "rows": rows,
            "gpu_beats_one_core_from_n": first_win("cpu_one_core_us"),
            "gpu_beats_all_cores_from_n": first_win("cpu_all_cores_us"),
            "note": "below the first figure a literal drop-in under an un-batched caller is a slow-down: keep ScalarStrategy "
                    "there and batch (INTEGRATION.md, 'when NOT to route through HipStrategy')"}



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--perms-per-gpu", type=int, default=0,
                    help="default: 2^26 at every N (BASELINE configs[2]); 134217728 = configs[4] (2^30 over 8 GPUs)")
    ap.add_argument("--total-perms", type=int, default=0,
                    help="strong scaling: this many permutations IN ALL, sharded over the ranks (1073741824 = BASELINE "
                         "configs[4] at any N); overrides --perms-per-gpu")
    ap.add_argument("--kernel", type=int, default=0,
                    help="0 default dispatch (k_perm_fast above 16384 states), 1 literal, 2 fast, 3 coop (five waves "
                         "per state), 4 lanes (lane-split low latency)")
    ap.add_argument("--workload", default="perm", choices=["perm", "merkle"],
                    help="perm: the headline (BASELINE configs[2]); merkle: a step = one arity-4 tree build over "
                         "2^24 leaves (configs[3]; development -- the default run reports it under `secondary`)")
    ap.add_argument("--cpu-sample", type=int, default=1 << 20)
    ap.add_argument("--verify-sample", type=int, default=2048, help="states per rank checked against the oracle (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Merkle / host-path records")
    # test hooks for boxes with fewer GPUs than ranks (control-flow check of the N>1 path only)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--single-device", action="store_true", help="every rank uses cuda:0 (testing)")
    ap.add_argument("--config5-full-size", action="store_true",
                    help="with --single-device at world size 8: run secondary.config5_2p30 at its real size (2^27 states per "
                         "rank, 160 GiB on the one device) instead of the rehearsal size (testing)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))

    import torch
    from hades252_amd import build, sharding, _lib
    build.build(verbose=False)
    from hades252_amd import strategy as H

    rank, local_rank, world = sharding.env_world()
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: hades252_amd has no CPU fallback")
    dev_index = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # (HADES252_BENCH_FORCE_DIST=1: initialise the process group at world size 1 too -- how the RCCL bookkeeping path, barrier /
    #  max / gather / digest sum on GPU tensors, is exercised on a one-GPU box: tests/test_bench_contract.py)
    if world > 1 or os.environ.get("HADES252_BENCH_FORCE_DIST") == "1":
        try:
            sharding.init_process_group(args.dist_backend)
        except Exception as e:                     # no silent fallback, no re-exec: say why and leave with a failure
            print("bench.py: rank %d: init_process_group(%r) failed: %r" % (rank, args.dist_backend, e), file=sys.stderr,
                  flush=True)
            raise SystemExit(3)

    # N ranks must sit on N distinct physical devices: every rank's PCI address / UUID, gathered
    devices = sharding.gather_strings(sharding.device_identity(torch, dev_index), device)
    ranks_seen = sharding.world_size()
    if ranks_seen != world:
        raise SystemExit("bench.py: rank %d: process group has %d ranks, WORLD_SIZE says %d" % (rank, ranks_seen, world))
    if world > 1 and not args.single_device and not sharding.distinct_devices(devices):
        if rank == 0:
            print("bench.py: two ranks report the same physical device (pass --single-device to allow it): %r" % devices,
                  file=sys.stderr, flush=True)
        raise SystemExit(4)

    if args.workload == "merkle":
        return bench_merkle(args, H, torch, device, sharding, rank, world)

    strong = args.total_perms > 0
    if strong:
        first_perm, end = sharding.strong_shard(rank, world, args.total_perms)
        n, n_max = end - first_perm, -(-args.total_perms // world)
    else:
        n = n_max = args.perms_per_gpu or (1 << 26)
        first_perm, _ = sharding.weak_shard(rank, n)
    kernel_name = kernel_of(args.kernel, n_max)
    strat = H.ScalarStrategy(args.kernel)
    states = torch.empty((n, 5, 4), dtype=torch.int64, device=device)
    H.gen_b(5 * n, device, first_elem=5 * first_perm, out=states.view(-1, 4))
    check = ShardCheck(torch, states, first_perm, n, args.verify_sample)

    # the parity launch (untimed, before the warm-up): EVERY output of this rank's block against the oracle, through the
    # 256-bit position-dependent digest the oracle computed over the same 2^26 states (headline configuration only)
    strat.perm(states)
    gold_block = golden_range_digest(first_perm, n)
    block_ok = None
    if gold_block is not None:
        block_ok = ["%016x" % (d & 0xFFFFFFFFFFFFFFFF) for d in H.digest(states, first_index=20 * first_perm)] == list(gold_block)
    for _ in range(args.warmup):
        strat.perm(states)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    sharding.barrier(device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        strat.perm(states)
        b.record()
    torch.cuda.synchronize()
    sharding.barrier(device)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.reduce_max(elapsed, device)

    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs))
    kernel_ms_max = sharding.reduce_max(kernel_ms, device)
    per_rank_ms = sharding.gather_floats(kernel_ms, device)
    # the timed launches themselves, checked on every rank: sampled states after warm-up + timed steps
    rank_ok = check.after(1 + args.warmup + args.steps) and block_ok is not False
    all_ok = sharding.reduce_min_int(1 if rank_ok else 0, device) == 1
    blocks_covered = sharding.reduce_sum_int(1 if block_ok is not None else 0, device)
    blocks_ok = sharding.reduce_sum_int(1 if block_ok else 0, device)
    digest = sharding.combine_digests(H.digest(states, first_index=20 * first_perm), device)

    del states, check
    torch.cuda.empty_cache()
    # the records every rank takes part in (N > 1): configs[4] and the sharded tree with its one exchange step
    multi = {}
    if not args.no_secondary:
        # BASELINE configs[4] at this world size (N = 1 included: the strong-scaling curve needs its first point) unless the
        # headline already IS that measurement; with --single-device the same code at rehearsal size
        recs = []
        if not (strong and args.total_perms == 1 << 30):
            c5_name = "config5_2p30" if ((not args.single_device) or args.config5_full_size) else "config5_rehearsal"
            recs.append((c5_name, lambda: config5_record(args, H, torch, device, sharding, rank, world)))
        if world > 1:
            recs.append(("merkle_2p24_sharded", lambda: merkle_sharded_record(H, torch, device, sharding, rank, world)))
        for name, fn in recs:
            try:
                multi[name] = fn()
            except Exception as e:           # (local failures are agreed on inside the records, before their first collective;
                multi[name] = {"error": repr(e)}   # a ValueError -- a world size the tree does not split over -- is the same on every rank)
    backend_name = sharding.backend_name()
    sharding.shutdown(device)                # the last collective: from here on rank 0 works alone
    if rank != 0:
        if not rank_ok:
            raise SystemExit("rank %d: GPU output differs from the CPU oracle" % rank)
        return
    per_step = args.total_perms if strong else n * world
    total_perms = per_step * args.steps
    value = total_perms / elapsed
    n = n_max                                   # (strong scaling with N not dividing T: shards differ by one state)
    achieved = ALGO_BYTES_PER_PERM * n / (kernel_ms_max * 1e-3) / 1e9
    traffic, traffic_source = None, None
    prof, sec_prof = profile_record(build, kernel_name, n)
    if prof and prof.get("hbm_bytes_per_launch"):
        traffic = prof.get("hbm_bytes_per_launch")
        traffic_source = ("NOT measured in this run: replayed from the rocprofv3 PMC passes of commit %s "
                          "(profiles/hbm_traffic.json; same kernel source hash %s, same launch size)"
                          % (str(prof.get("measured_at_commit"))[:12], build.perm_fast_hash()[:12]))
    pow2 = n & (n - 1) == 0
    out = {
        "metric": "Hades252 permutations/sec (WIDTH=5, BLS12-381 Fr)",
        "value": value, "unit": "permutations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": (("%s independent WIDTH=5 permutations IN ALL, sharded over %d GPU(s): %d per GPU, in place in "
                                 "HBM (%sgenerator B, Montgomery-limb AoS records)"
                                 % (("2^%d" % (per_step.bit_length() - 1)) if per_step & (per_step - 1) == 0 else str(per_step),
                                    world, n, "BASELINE configs[4]; " if per_step == 1 << 30 else "")) if strong else
                                ("2^%d independent WIDTH=5 permutations per GPU, in place in HBM (BASELINE %s; generator "
                                 "B, Montgomery-limb AoS records)"
                                 % (n.bit_length() - 1, "configs[4]: 2^30 over 8 GPUs" if (world == 8 and n == 1 << 27)
                                    else "configs[2]")) if pow2 else "%d permutations per GPU" % n),
                   "perms_per_gpu": n, "total_perms_per_step": per_step, "state_bytes": 160,
                   "kernel": kernel_name,
                   "sharding": "contiguous range per rank, no collective"},
        "per_gpu": {"value": value / world, "unit": "permutations/s",
                    "kernel_ms_per_rank": per_rank_ms,
                    "perms_per_s_per_rank": [n / (ms * 1e-3) for ms in per_rank_ms],
                    "device": devices},
        "dist": {"backend": backend_name, "ranks_seen": ranks_seen, "world_size_env": world,
                 "distinct_devices": sharding.distinct_devices(devices), "single_device": bool(args.single_device),
                 "data_path_collectives": "none (bookkeeping only: barrier, max of times, AND of checks, sum of digests)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": kernel_name, "kernel_ms": kernel_ms_max,
                     "algorithmic_bytes_per_perm": ALGO_BYTES_PER_PERM,
                     "note": "HBM traffic equals the algorithmic bytes; the kernel is VALU-issue bound "
                             "(~84 k instructions per 64 x 320 B), see valu_issue and DESIGN.md"},
        "digest": ["%016x" % d for d in digest],
        "parity_vs_cpu_sample": all_ok,
        "parity_sample": "%d states of every rank's shard, read back after the %d launches (parity launch + warm-up + timed) of "
                         "the timed kernel and compared with the CPU oracle applied as many times; AND over ranks"
                         % (args.verify_sample, 1 + args.warmup + args.steps),
        # all 2^26 outputs of the first launch, on every rank whose block the oracle's digests cover (blocks 0 .. 7)
        # (null = no rank's range is covered by a committed oracle digest; false = a covered rank's digest differs)
        "parity_all_outputs_first_launch": (blocks_ok == blocks_covered) if blocks_covered else None,
        "parity_all_outputs": "digest (hades252_digest_dev, global indices) of ALL outputs of each rank's first launch == the CPU "
                              "oracle's digest of the same states (tests/golden/kat.json headline_2p26_blocks / config5_2p30 "
                              "shards, summed over the pieces a rank's range covers); ranks covered: %d of %d, equal: %d; a "
                              "mismatch fails the job through parity_vs_cpu_sample" % (blocks_covered, world, blocks_ok),
    }
    if prof and prof.get("valu_instructions_per_wave"):
        # instructions per wave (= per 64 permutations... per lane: per permutation) from the counters of the keyed record
        ops32 = float(prof["valu_instructions_per_wave"]) - OPS64_PER_PERM
        eq = (OPS64_PER_PERM + 0.5 * ops32) * n / (kernel_ms_max * 1e-3) / 64 / 1e9
        out["valu_issue"] = {
            "bound": "VALU issue: 64-bit integer multiply-add pipe (the binding bound; HBM idles at 1.8 %)",
            "achieved": eq, "unit": "G 64-bit-equivalent wave-instr/s",
            "peak": VALU_IDEAL_G_WI, "frac": eq / VALU_IDEAL_G_WI,
            "valu_instructions_per_wave": prof["valu_instructions_per_wave"],
            "instructions_source": "SQ_INSTS_VALU / SQ_WAVES of the same keyed record as roofline.traffic",
            "mads_per_perm": MADS_PER_PERM, "ops64_per_perm": OPS64_PER_PERM, "ops32_per_perm": ops32,
            "note": "achieved = (64-bit ops + 0.5 x 32-bit ops) per permutation x permutations/s / 64 lanes; peak = 1024 "
                    "SIMDs x 2.4 GHz / 4 cycles (ideal pipe at the peak clock); only fewer instructions can make the "
                    "kernel faster (DESIGN.md section 5)"}
    if not args.no_cpu_baseline:
        # checked with the kernel that was timed, whatever the sample size would make the default dispatch pick
        timed_kernel = args.kernel or H.kernel_for(n)
        cb, ok = cpu_baseline_and_check(H, torch, device, args.cpu_sample, timed_kernel)
        out["cpu_baseline"] = cb
        out["parity_vs_cpu_sample"] = all_ok = all_ok and ok
    wrong = []
    if not args.no_secondary:
        sec = dict(multi)
        try:
            sec["merkle_2p24"], leaves = merkle_record(H, torch, device, 24)
            del leaves
            torch.cuda.empty_cache()
            attach_traffic(sec["merkle_2p24"]["roofline"], sec_prof, "merkle_2p24_tree", MERKLE_BYTES_PER_NODE * sec["merkle_2p24"]["nodes"])
            sec["single_perm"] = single_perm_record(H, torch, device)
            if "cpu_baseline" in out:
                sec["crossover"] = crossover_record(H, out["cpu_baseline"])
            sec["sponge_chain"] = sponge_chain_record(H, torch, device)
            sec["wire_format"] = wire_format_record(H, torch, device)
            for k in ("to_bytes", "from_bytes"):
                attach_traffic(sec["wire_format"][k]["roofline"], sec_prof, "wire_" + k, 64.0 * (1 << 26))
            torch.cuda.empty_cache()
            sec["gadget_witness"] = gadget_witness_record(H, torch, device)
            attach_traffic(sec["gadget_witness"]["roofline"], sec_prof, "witness",
                           sec["gadget_witness"]["roofline"]["algorithmic_bytes_per_state"] * float(1 << 20))
            attach_traffic(sec["gadget_witness"]["trace"]["roofline"], sec_prof, "trace", 160.0 * 68 * (1 << 20))
            attach_traffic(sec["gadget_witness"]["trace_scaled"]["roofline"], sec_prof, "trace_scaled", 160.0 * 68 * (1 << 20))
            torch.cuda.empty_cache()
            sec["host_path"] = host_path_record(22)
        except Exception as e:                       # secondary records never take the headline down
            sec["error"] = repr(e)
        out["secondary"] = sec
        # ... but a WRONG result does fail the job (after the line is out)
        if sec.get("merkle_2p24", {}).get("root_matches_golden") is False:
            wrong.append("the 2^24-leaf Merkle root differs from the CPU oracle's committed root")
        ms_ = sec.get("merkle_2p24_sharded") or {}
        if ms_.get("root_matches_golden") is False:
            wrong.append("the SHARDED 2^24-leaf Merkle root differs from the CPU oracle's committed root")
        c5 = sec.get("config5_2p30") or sec.get("config5_rehearsal") or {}
        if (c5.get("parity_vs_cpu_sample") is False or c5.get("digest_matches_one_device") is False
                or c5.get("shard_digests_match_oracle") is False):
            wrong.append("configs[4]: the shards differ from the oracle sample, from the oracle's shard digests or from the "
                         "one-device digest")
    print(json.dumps(out), flush=True)
    if not all_ok:
        raise SystemExit("GPU output differs from the CPU oracle")
    if wrong:
        raise SystemExit("; ".join(wrong))


def bench_merkle(args, H, torch, device, sharding, rank, world):
    """--workload merkle: a step = one tree build (root only) over 2^24 leaves per GPU, every rank its own tree."""
    rec, leaves = merkle_record(H, torch, device, 24, reps=max(1, args.steps))
    ms = sharding.reduce_max(rec["tree_ms"], device)
    if rank != 0:
        return
    nodes = rec["nodes"]
    out = {"metric": "Poseidon Merkle tree nodes/sec (arity 4, 2^24 leaves; Hades252 WIDTH=5)",
           "value": nodes * world / (ms * 1e-3), "unit": "nodes/s", "n_gpus": world, "steps": args.steps,
           "warmup": 1, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int64", "data": "synthetic", "config": {"workload": rec["workload"]},
           "roofline": rec["roofline"], "root": rec["root"], "root_matches_golden": rec["root_matches_golden"]}
    print(json.dumps(out), flush=True)
    if rec["root_matches_golden"] is False:
        raise SystemExit("the Merkle root differs from the CPU oracle's committed root")


if __name__ == "__main__":
    main()
