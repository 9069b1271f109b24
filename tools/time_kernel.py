"""Quick device-side timing of the permutation kernels (development helper)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hades252_amd import strategy as H, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
kernels = [int(k) for k in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2]
buf = H.gen_b(5 * n, "cuda")
for k in kernels:
    t = torch.zeros(20, dtype=torch.int64, device="cuda")
    if _lib.lib().hades252_perm_batch_dev_ex(t.data_ptr(), 1, None, k) != 0:
        print("kernel", k, "not built"); continue
    s = H.ScalarStrategy(k)
    s.perm(buf); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): s.perm(buf)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("kernel %d  n=%d  %.3f ms  %.2f Mperm/s" % (k, n, ms, n / ms / 1e3))
