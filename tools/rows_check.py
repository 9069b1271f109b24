import sys, time
sys.path.insert(0, "/root/repo")
import torch
from hades252_amd import strategy as H, _lib
for n in (1, 3, 4, 5, 16, 17, 1000, 1025, 4095, 4096, 4097, 10000):
    a = H.gen_b(5 * n, "cuda").view(n, 5, 4)
    b = a.clone()
    H.ScalarStrategy(_lib.KERNEL_ROWS).perm(a)
    H.ScalarStrategy(_lib.KERNEL_FAST).perm(b)
    print(n, bool(torch.equal(a, b)))
for n in (1, 1025, 2048, 4096, 8192):
    a = H.gen_b(5 * n, "cuda").view(n, 5, 4)
    for name, k in (("rows", _lib.KERNEL_ROWS), ("coop", _lib.KERNEL_COOP)):
        s = H.ScalarStrategy(k)
        for _ in range(5): s.perm(a)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); s.perm(a); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print("n=%d %s median %.1f us min %.1f us" % (n, name, sorted(ts)[10], min(ts)))
