"""witness_probe.py -- time hades252_perm_witness_dev (or, with a third argument `trace`, hades252_perm_trace_dev) alone
   (HIP events), for profiling runs (round 4).
   python tools/witness_probe.py [log2 n] [reps] [trace]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hades252_amd.strategy as H


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n = 1 << logn
    dev = torch.device("cuda:0")
    st = H.gen_b(5 * n, dev)
    trace = len(sys.argv) > 3 and sys.argv[3] == "trace"
    planes, name = (67 * 5, "trace") if trace else (972, "witness")
    wires = torch.empty((67, n, 5, 4) if trace else (972, n, 4), dtype=torch.int64, device=dev)
    run = (lambda: H.perm_trace(st.view(n, 5, 4), out=wires)) if trace else (lambda: H.perm_witness(st, out=wires))
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    ms = ts[len(ts) // 2]
    print("%s n=2^%d: %.3f ms  %.2f M perms/s  %.1f GB/s written" % (name, logn, ms, n / ms / 1e3, planes * 32 * n / ms / 1e6))


main()
