"""Random soak of the arithmetic (not of races: tools/soak.py): for `seconds` of wall clock, chunk after chunk of 2^24 fresh
states -- three chunks out of four spread over the WHOLE field [0, p) (generator-B scalars squared by the device's 8 x 32
arithmetic), one kept as generator B defines it (below 2^254) -- permuted by the reference's literal schedule on the
saturated 8 x 32 CIOS arithmetic (k_states_literal) and by the shipped scale-tracked radix-2^29 kernel (k_perm_fast), compared
bit for bit; the first 2^18 / 2^14 / 2^12 states of every chunk also through the five-waves, the rows and the lane-split
kernel, and 256 states of every chunk against the CPU oracle.  Round 6: the first 2^16 states of every chunk also through
the two per-round trace kernels -- the scaled trace, un-scaled with the library's table and its own field operations, must
equal the true-form trace in all 67 rounds (the exit routine of the scaled kernel over the whole field).  Prints one line
per 16 chunks and a summary.

    python tools/soak_random.py [seconds=600]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle_lib  # noqa: E402
from hades252_amd import strategy as H, _lib  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
n = 1 << 24
orc = oracle_lib.load()
lit, fast = H.ScalarStrategy(_lib.KERNEL_LITERAL), H.ScalarStrategy(_lib.KERNEL_FAST)
small = ((_lib.KERNEL_COOP, 1 << 18), (_lib.KERNEL_ROWS, 1 << 14), (_lib.KERNEL_LANES, 1 << 12))
mul_t, add_t = H.trace_scale_table()
m_tr = 1 << 16
mul_dev = [torch.from_numpy(np.tile(mul_t[r], 5 * m_tr).view(np.int64)).cuda().view(-1, 4) for r in range(67)]
add_dev = [torch.from_numpy(np.tile(add_t[r].reshape(-1), m_tr).view(np.int64)).cuda().view(-1, 4) for r in range(67)]
t0 = time.time()
chunks = bad = checked_oracle = top_max = 0
rng = np.random.default_rng(20250101)
while time.time() - t0 < seconds:
    seed = 0x5EED0000 + chunks
    a = H.gen_b(5 * n, "cuda", seed=seed)
    if chunks % 4 != 3:
        a = H.fr_op(H.FR_SQUARE, a, None, H.FR_IMPL_SATURATED32)
    a = a.view(n, 5, 4)
    ref = a.clone()
    lit.perm(ref)
    b = a.clone()
    fast.perm(b)
    ok = torch.equal(ref, b)
    for k, m in small:
        c = a[:m].clone()
        H.ScalarStrategy(k).perm(c)
        ok = ok and torch.equal(c, ref[:m])
    scaled, true = H.perm_trace_scaled(a[:m_tr]), H.perm_trace(a[:m_tr])
    for r in range(67):
        un = H.fr_op(H.FR_ADD, H.fr_op(H.FR_MUL, scaled[r].reshape(-1, 4), mul_dev[r]), add_dev[r])
        ok = ok and torch.equal(un.view(-1), true[r].reshape(-1))
    ok = ok and torch.equal(true[66], ref[:m_tr])
    del scaled, true
    idx = torch.from_numpy(np.sort(rng.choice(n, 256, replace=False))).cuda()
    inp = a[idx].contiguous().cpu().numpy().view(np.uint64).reshape(-1)
    got = b[idx].contiguous().cpu().numpy().view(np.uint64).reshape(-1)
    ok = ok and bool((orc.perm_batch(inp.copy()) == got).all())
    checked_oracle += 256
    chunks += 1
    if not ok:
        bad += 1
        print("MISMATCH in chunk %d (seed 0x%x)" % (chunks - 1, seed), flush=True)
    top_max = max(top_max, int(a.view(-1, 4)[:, 3].max().item()))       # (Montgomery limbs: the whole range is [0, 0x73ed...])
    if chunks % 16 == 0:
        print("chunk %5d  %.3e states so far  mismatching chunks %d  largest top limb so far 0x%016x  %.0f s"
              % (chunks, chunks * n, bad, top_max, time.time() - t0), flush=True)
print("random soak: %d x 2^24 = %.3e states literal (8 x 32 CIOS) vs shipped (radix 2^29, scale-tracked), + the three latency "
      "kernels on 2^18 / 2^14 / 2^12 states of every chunk, + scaled trace x table == true trace (67 rounds) on 2^16 states of "
      "every chunk, + %d states vs the CPU oracle: mismatching chunks %d, %.0f s"
      % (chunks, chunks * n, checked_oracle, bad, time.time() - t0))
sys.exit(1 if bad else 0)
