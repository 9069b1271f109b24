"""Wire-format kernels (BlsScalar::to_bytes / from_bytes on device) at a working set far beyond the 256 MB
Infinity Cache: 2^26 scalars = 2 GiB in + 2 GiB out per launch.  Run plain for timings, and under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) for the HBM byte counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hades252_amd import strategy as H

dev = torch.device("cuda", 0)
n = 1 << 26
limbs = H.gen_b(n, dev)
out = torch.empty_like(limbs)
canon = H.to_bytes(limbs)
for name, fn in (("to_bytes", lambda: H.to_bytes(limbs, out)), ("from_bytes", lambda: H.from_bytes(canon, out))):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
    print("%-10s n=2^26 scalars (2 GiB in + 2 GiB out)  median %7.3f ms  %7.1f GB/s algorithmic (64 B/scalar)"
          % (name, ms, 64 * n / (ms * 1e-3) / 1e9))
