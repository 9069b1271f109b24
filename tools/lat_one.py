"""200 launches each of the low-latency kernels on ONE resident state, 50 builds of a 2^16-leaf arity-4 tree, and the
chain entry points (sponge, path verification, tree update) on ONE message / opening / leaf:
run under `rocprofv3 --kernel-trace --stats` (tools/profile_round.sh) for the kernels' own durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hades252_amd import strategy as H

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
buf = H.gen_b(5, "cuda")
for k in (4, 5, 3, 2):          # lanes, rows, five waves, one state per lane
    s = H.ScalarStrategy(k)
    for _ in range(200):
        s.perm(buf)
    torch.cuda.synchronize()
leaves = H.gen_b(1 << 16, "cuda")
tag = 15 * ((1 << 256) % P) % P
for _ in range(50):
    H.merkle_root(leaves, 4, tag, 1)
torch.cuda.synchronize()
# the dependent-permutation entry points on small batches (one chain per wave): ONE message of 200 blocks, one opening
# and one updated leaf of a 4^8-leaf tree
pool = H.gen_b(800, "cuda")
off = torch.zeros(1, dtype=torch.int64, device="cuda")
ln = torch.full((1,), 799, dtype=torch.int64, device="cuda")
for _ in range(20):
    H.sponge_hash_var(pool, off, ln, 1, 1)
torch.cuda.synchronize()
tree = H.merkle_build(leaves, 4, tag, 1)
idx = torch.tensor([12345], dtype=torch.int64, device="cuda")
paths = H.merkle_open(leaves, tree, 4, idx)
lv = leaves[idx].contiguous()
for _ in range(50):
    H.merkle_verify(lv, idx, paths, 4, tag, 1)
    H.merkle_update(leaves, tree, 4, idx, tag, 1)
torch.cuda.synchronize()
