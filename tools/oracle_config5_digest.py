#!/usr/bin/env python3
"""BASELINE configs[4] against the ORACLE at full size: the additive digest (include/hades252.h, hades252_digest_dev; numpy
restatement tests/oracle_lib.py::digest_ref) of perm(generator-B states 0 .. 2^30 - 1), computed by the C oracle on this
host's CPU cores -- about 2.5 h on 6 threads -- shard by shard (the 8 shards of the 8-GPU decomposition), resumable.

    nice -n 19 python tools/oracle_config5_digest.py [threads] [log2_total]   ->  gpurun_out/oracle_config5_digest.json

tests/golden/kat.json `config5_2p30` holds the result: `oracle_digest` (whole range) and `oracle_shard_digests`; the
device-derived digest recorded there since round 2 must equal it (tests/test_oracle.py::test_config5_golden_digest_is_the_oracles)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

M64 = (1 << 64) - 1
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
log_total = int(sys.argv[2]) if len(sys.argv) > 2 else 30
total, world, chunk = 1 << log_total, 8, 1 << 18
out_path = os.path.join(ROOT, "gpurun_out", "oracle_config5_digest_2p%d.json" % log_total)
os.makedirs(os.path.dirname(out_path), exist_ok=True)
state = {"n": total, "world": world, "chunk": chunk, "next_state": 0, "shard_digests": [[0, 0, 0, 0] for _ in range(world)],
         "seconds": 0.0, "threads": threads}
# resume: from the output file itself, or (a GPU box does not receive gpurun_out/) from a copy of an earlier call's partial
# result placed at tools/oracle_config5_digest.resume.json
for cand in (out_path, os.path.join(ROOT, "tools", "oracle_config5_digest.resume.json")):
    if os.path.exists(cand):
        old = json.load(open(cand))
        if old.get("n") == total and old.get("chunk") == chunk and old.get("next_state", 0) > state["next_state"]:
            state = old
import numpy as np  # noqa: E402


def digest_fast(words, first_index):
    """digest_ref for a range that starts and ends on a multiple of 4 words: the four lanes are the columns of a reshape."""
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    assert first_index % 4 == 0 and w.size % 4 == 0
    idx = np.arange(w.size, dtype=np.uint64) + np.uint64(first_index)
    with np.errstate(over="ignore"):
        z = w ^ (idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0xD1B54A32D192ED03))
        z = (z ^ (z >> np.uint64(32))) * np.uint64(0xD6E8FEB86659FD93)
        z = (z ^ (z >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(32)
        return [int(x) for x in z.reshape(-1, 4).sum(axis=0, dtype=np.uint64)]


orc = oracle_lib.load()
probe = orc.perm_batch(orc.gen_b(5 * 12345, 5 * 1000), 1)
assert digest_fast(probe, 20 * 12345) == oracle_lib.digest_ref(probe, 20 * 12345)
per = total // world
t_last = time.time()
while state["next_state"] < total:
    b = state["next_state"]
    n = min(chunk, total - b)
    t0 = time.time()
    outp = orc.perm_batch(orc.gen_b(5 * b, 5 * n), threads)
    d = digest_fast(outp, 20 * b)
    g = b // per
    assert (b + n - 1) // per == g, "a chunk never straddles two shards"
    state["shard_digests"][g] = [(x + y) & M64 for x, y in zip(state["shard_digests"][g], d)]
    state["next_state"] = b + n
    state["seconds"] += time.time() - t0
    if time.time() - t_last > 60 or state["next_state"] == total:
        t_last = time.time()
        with open(out_path + ".tmp", "w") as f:
            json.dump(state, f)
        os.replace(out_path + ".tmp", out_path)
        print("%.2f %% done, %.0f s of CPU wall so far" % (100.0 * state["next_state"] / total, state["seconds"]), flush=True)
whole = [0, 0, 0, 0]
for sd in state["shard_digests"]:
    whole = [(x + y) & M64 for x, y in zip(whole, sd)]
state["digest"] = ["%016x" % x for x in whole]
state["shard_digests_hex"] = [["%016x" % x for x in sd] for sd in state["shard_digests"]]
with open(out_path, "w") as f:
    json.dump(state, f, indent=1)
print("DONE", state["digest"])
