"""The kernels behind bench.py's `secondary` rooflines at the sizes bench.py runs them: the 2^24-leaf arity-4 tree, the
wire-format kernels at 2^26 scalars, the gadget witness and the per-round trace at 2^20 states -- bench.py's own record
functions, nothing else.  Run plain for the timings (stdout: one JSON line), and under `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` in separate passes (tools/profile_round.sh) for the HBM byte counts tools/summarize_profile.py turns
into `secondary_kernels` of profiles/hbm_traffic.json.  The number of launches of every kind is COUNTED by the record
functions themselves and printed, so that the summary divides the counter sums by what really ran."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from hades252_amd import strategy as H  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
out = {}
rec, leaves = bench.merkle_record(H, torch, dev, 24, reps=5)
del leaves
torch.cuda.empty_cache()
out["merkle_2p24"] = {"tree_ms": rec["tree_ms"], "trees_built": rec["trees_built"], "root_matches_golden": rec["root_matches_golden"]}
w = bench.wire_format_record(H, torch, dev)
torch.cuda.empty_cache()
# (to_bytes runs once more than from_bytes: it also produces the canonical input of from_bytes)
out["wire_format"] = {"to_bytes_ms": w["to_bytes"]["ms"], "from_bytes_ms": w["from_bytes"]["ms"], "to_bytes_launches": w["launches"]["to_bytes"],
                      "from_bytes_launches": w["launches"]["from_bytes"]}
g = bench.gadget_witness_record(H, torch, dev)
out["gadget_witness"] = {"witness_ms": g["ms"], "trace_ms": g["trace"]["ms"], "witness_launches": g["launches"],
                         "trace_launches": g["trace"]["launches"], "trace_scaled_ms": g["trace_scaled"]["ms"],
                         "trace_scaled_launches": g["trace_scaled"]["launches"]}
print(json.dumps(out), flush=True)
