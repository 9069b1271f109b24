"""The two per-round trace kernels and the throughput kernel at 2^21 states, a few launches each (for rocprofv3 --pmc)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hades252_amd import strategy as H  # noqa: E402

dev = torch.device("cuda", 0)
n = 1 << 21
st = H.gen_b(5 * n, dev).view(n, 5, 4)
trace = torch.empty((67, n, 5, 4), dtype=torch.int64, device=dev)
for _ in range(3):
    H.perm_trace(st, out=trace)
    H.perm_trace_scaled(st, out=trace)
    H.ScalarStrategy(2).perm(st.clone())
torch.cuda.synchronize()
