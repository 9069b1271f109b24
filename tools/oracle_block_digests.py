#!/usr/bin/env python3
"""The headline batch against the ORACLE, every output: digests (include/hades252.h hades252_digest_dev; global word
indices) of perm(generator-B states) per BLOCK of 2^26 states -- block g = states [g 2^26, (g + 1) 2^26) = what rank g of
`bench.py --gpus N` holds after its first launch.  ~4 min per block on 16 host threads, ~16 min on 8.

    nice -n 19 python tools/oracle_block_digests.py <first_block> <end_block> [threads]  ->  gpurun_out/oracle_blocks_2p26_<a>_<b>.json

tests/golden/kat.json `headline_2p26_blocks` holds blocks 0 .. 7 (N = 1, 2, 4, 8 at 2^26 states per GPU).  Cross-check built
in: blocks 2 g and 2 g + 1 add up to shard g of `config5_2p30.oracle_shard_digests`, computed by an independent run."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

M64 = (1 << 64) - 1
a, b = int(sys.argv[1]), int(sys.argv[2])
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
BLOCK, CHUNK = 1 << 26, 1 << 18


def digest_fast(words, first_index):
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
probe = orc.perm_batch(orc.gen_b(5 * 777, 5 * 500), 1)
assert digest_fast(probe, 20 * 777) == oracle_lib.digest_ref(probe, 20 * 777)
out_path = os.path.join(ROOT, "gpurun_out", "oracle_blocks_2p26_%d_%d.json" % (a, b))
os.makedirs(os.path.dirname(out_path), exist_ok=True)
res = {"block_states": BLOCK, "blocks": {}, "threads": threads}
t0 = time.time()
for g in range(a, b):
    acc = [0, 0, 0, 0]
    for first in range(g * BLOCK, (g + 1) * BLOCK, CHUNK):
        outp = orc.perm_batch(orc.gen_b(5 * first, 5 * CHUNK), threads)
        acc = [(x + y) & M64 for x, y in zip(acc, digest_fast(outp, 20 * first))]
    res["blocks"][str(g)] = ["%016x" % x for x in acc]
    res["seconds"] = time.time() - t0
    with open(out_path + ".tmp", "w") as f:
        json.dump(res, f, indent=1)
    os.replace(out_path + ".tmp", out_path)
    print("block %d: %s  (%.0f s so far)" % (g, " ".join(res["blocks"][str(g)]), res["seconds"]), flush=True)
