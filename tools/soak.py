"""Soak: the barrier-carrying kernels (helped lane-split forms, five-waves chains) and the host pipelines, many repetitions on
fresh data, every result compared with the throughput kernel's -- looks for rare races, not for arithmetic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hades252_amd import strategy as H, _lib

torch.manual_seed(1)
bad = 0
fast = H.ScalarStrategy(_lib.KERNEL_FAST)
for it in range(300):
    n = int(torch.randint(1, 4200, (1,)).item())
    a = H.gen_b(5 * n, "cuda", first_elem=it * 100003).view(n, 5, 4)
    ref = a.clone()
    fast.perm(ref)
    for k in (_lib.KERNEL_DEFAULT, _lib.KERNEL_LANES, _lib.KERNEL_ROWS, _lib.KERNEL_COOP):
        if k == _lib.KERNEL_LANES and n > 2048:
            continue
        b = a.clone()
        H.ScalarStrategy(k).perm(b)
        if not torch.equal(b, ref):
            bad += 1
            print("MISMATCH perm it=%d n=%d kernel=%d" % (it, n, k))
print("perm forms: 300 batches, mismatches:", bad)
# chains: sponge / verify at sizes on each side of the thresholds vs the per-lane kernels (forced by padding the batch)
cap = 12345
for it in range(60):
    n = int(torch.randint(1, 3000, (1,)).item())
    ln = int(torch.randint(0, 30, (1,)).item())
    big = 20000
    pool = H.gen_b(big * max(ln, 1) + 8, "cuda", first_elem=it * 7919)
    offs = (torch.arange(big, dtype=torch.int64, device="cuda") * max(ln, 1))
    lens = torch.full((big,), ln, dtype=torch.int64, device="cuda")
    ref = H.sponge_hash_var(pool, offs, lens, cap, 1)                       # 20 000 messages: one per lane
    got = H.sponge_hash_var(pool, offs[:n].contiguous(), lens[:n].contiguous(), cap, 1)
    if not torch.equal(got, ref[:n]):
        bad += 1
        print("MISMATCH sponge it=%d n=%d len=%d" % (it, n, ln))
print("sponge chains: 60 batches, total mismatches:", bad)
# host pipelines
for it in range(12):
    n = int(torch.randint(1, 1 << 20, (1,)).item())
    a = H.gen_b(5 * n, "cuda", first_elem=it).view(n, 5, 4)
    host = a.cpu().numpy().view(np.uint64).reshape(-1).copy()
    H.ScalarStrategy().perm(host)
    fast.perm(a)
    if not (host == a.cpu().numpy().view(np.uint64).reshape(-1)).all():
        bad += 1
        print("MISMATCH host perm it=%d n=%d" % (it, n))
    lv = H.gen_b(max(n, 2), "cuda", first_elem=it * 31)
    r_dev = H.merkle_root(lv, 4, 15, 1).cpu().numpy().view(np.uint64)
    r_host = H.merkle_root_host(lv.cpu().numpy().view(np.uint64).reshape(-1).copy(), 4, 15, 1)
    if not (r_dev == r_host).all():
        bad += 1
        print("MISMATCH host merkle it=%d n=%d" % (it, n))
print("host paths: 12 rounds, total mismatches:", bad)
# gadget witness / per-round trace (true-form rounds) against the throughput kernel and each other, ragged sizes
for it in range(24):
    n = int(torch.randint(1, 5000, (1,)).item())
    a = H.gen_b(5 * n, "cuda", first_elem=it * 65537).view(n, 5, 4)
    ref = a.clone()
    fast.perm(ref)
    w = H.perm_witness(a)
    tr = H.perm_trace(a)
    last = torch.stack([w[962 + 2 * j + 1] for j in range(5)], dim=1)
    if not (torch.equal(last, ref) and torch.equal(tr[66], ref)):
        bad += 1
        print("MISMATCH witness / trace vs perm it=%d n=%d" % (it, n))
print("witness / trace: 24 batches, total mismatches:", bad)
sys.exit(1 if bad else 0)
