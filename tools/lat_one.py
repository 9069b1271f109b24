"""200 launches each of the two low-latency kernels on ONE resident state, and 50 builds of a 2^16-leaf arity-4 tree:
run under `rocprofv3 --kernel-trace --stats` (tools/profile_round.sh) for the kernels' own durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hades252_amd import strategy as H

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
buf = H.gen_b(5, "cuda")
for k in (4, 3, 2):
    s = H.ScalarStrategy(k)
    for _ in range(200):
        s.perm(buf)
    torch.cuda.synchronize()
leaves = H.gen_b(1 << 16, "cuda")
tag = 15 * ((1 << 256) % P) % P
for _ in range(50):
    H.merkle_root(leaves, 4, tag, 1)
torch.cuda.synchronize()
