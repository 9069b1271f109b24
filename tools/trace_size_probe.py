"""Per-round trace kernels against the batch size: 2^20 states are 16 waves per SIMD at 3 resident (LDS slab), i.e. six
passes of which the last is one third full.  HIP events, median of 5."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hades252_amd import strategy as H  # noqa: E402

dev = torch.device("cuda", 0)
for n in (1 << 18, 3 << 16, 1 << 19, 3 << 17, 983040, 1 << 20, 3 << 18, 1 << 21, 3 << 19, 1 << 22):
    st = H.gen_b(5 * n, dev).view(n, 5, 4)
    trace = torch.empty((67, n, 5, 4), dtype=torch.int64, device=dev)
    row = "n %8d (%5.2f waves/SIMD)" % (n, n / 64 / 1024)
    for name, fn in (("true", lambda: H.perm_trace(st, out=trace)), ("scaled", lambda: H.perm_trace_scaled(st, out=trace))):
        fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        med = sorted(ms)[2]
        row += "  %s %7.3f ms %6.1f M perms/s %5.3f of HBM" % (name, med, n / med / 1e3, 160.0 * 68 * n / (med * 1e-3) / 8e12)
    print(row, flush=True)
    del trace, st
    torch.cuda.empty_cache()
