import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from hades252_amd import strategy as H
s = H.ScalarStrategy()
rng = np.random.default_rng(1)
for n in (1, 2, 16, 256):
    host = rng.integers(0, 1 << 62, size=20 * n, dtype=np.uint64)
    for _ in range(50): s.perm(host)
    ts = []
    for _ in range(301):
        t0 = time.perf_counter(); s.perm(host); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort(); print("n=%4d  host call median %.1f us  min %.1f  p90 %.1f" % (n, ts[150], ts[0], ts[270]), flush=True)
