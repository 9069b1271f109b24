"""Soak / differential test on the GPU (development + evidence, not part of pytest):
  1. literal kernel vs shipped kernel vs five-waves-per-state kernel on N_CHUNKS x 2^26 random states with
     different seeds (digest of all outputs must agree);
  2. every 5-tuple over a set of edge values (0, 1, p-1, R, ...), literal vs shipped vs cooperative, all bits.
The literal kernel is the reference's schedule verbatim and is itself compared with the CPU
oracle in tests/."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hades252_amd import strategy as H

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % P
n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
n = 1 << 26
a = torch.empty((n, 5, 4), dtype=torch.int64, device=dev)
t0 = time.time()
bad = 0
for c in range(n_chunks):
    seed = 0x1234567 + 977 * c
    H.gen_b(5 * n, dev, seed=seed, out=a.view(-1, 4))
    H.ScalarStrategy(2).perm(a)
    d_fast = H.digest(a)
    H.gen_b(5 * n, dev, seed=seed, out=a.view(-1, 4))
    H.ScalarStrategy(1).perm(a)
    d_lit = H.digest(a)
    H.gen_b(5 * n, dev, seed=seed, out=a.view(-1, 4))
    H.ScalarStrategy(3).perm(a)
    d_coop = H.digest(a)
    ok = d_fast == d_lit == d_coop
    bad += not ok
    print("chunk %2d seed %#x  %s  digest %016x" % (c, seed, "ok" if ok else "MISMATCH", d_fast[0]), flush=True)
print("random soak: %d x 2^26 = %.3g states, mismatching chunks: %d, %.1f s" % (n_chunks, n_chunks * n, bad, time.time() - t0))
del a

edge = [0, 1, 2, P - 1, P - 2, R, P - R, (1 << 255) % P, (1 << 254) - 1, 0xFFFFFFFF, P - (1 << 32),
        0xFFFFFFFF00000000, (P - 1) // 2, (1 << 128) - 1]
tab = np.array([[(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for v in edge], dtype=np.uint64)
idx = np.array(list(itertools.product(range(len(edge)), repeat=5)), dtype=np.int64)       # 14^5 tuples
states = tab[idx]                                                                          # [N,5,4]
x = torch.from_numpy(states.view(np.int64)).to(dev).contiguous()
y = x.clone()
z = x.clone()
H.ScalarStrategy(2).perm(x)
H.ScalarStrategy(1).perm(y)
H.ScalarStrategy(3).perm(z)
same = torch.equal(x, y) and torch.equal(x, z)
print("edge 5-tuples: %d states, literal == shipped == cooperative: %s" % (idx.shape[0], same))
sys.exit(0 if (bad == 0 and same) else 1)
