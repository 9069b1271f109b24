"""Does splitting a forest over several streams hide the latency-bound small levels behind other parts' bulk levels?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hades252_amd import strategy as H, _lib

dev = torch.device("cuda", 0)
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
tag = 15 * ((1 << 256) % P) % P


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for nt, k in ((4096, 4), (10000, 4), (65536, 2), (1024, 6)):
    per = 4 ** k
    leaves = H.gen_b(nt * per, dev)
    nodes = nt * (per - 1) // 3
    ref = H.merkle_forest(leaves, nt, 4, tag, 1)
    for parts in (1, 2, 3, 4, 8):
        if nt % parts: continue
        streams = [torch.cuda.Stream() for _ in range(parts)]
        chunk = nt // parts
        scr = [torch.empty(max(_lib.lib().hades252_merkle_forest_scratch_bytes(chunk, per, 4) // 8, 2), dtype=torch.int64, device=dev) for _ in range(parts)]
        outs = [None] * parts

        def run():
            cur = torch.cuda.current_stream()
            for i, st in enumerate(streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    outs[i] = H.merkle_forest(leaves[i * chunk * per:(i + 1) * chunk * per], chunk, 4, tag, 1, scr[i])
            for st in streams:
                cur.wait_stream(st)
        dt = timed(run)
        ok = torch.equal(torch.cat(outs), ref)
        print("forest %6d x 4^%d, %d stream(s): %8.3f ms  %8.2f Mperm/s  %s" % (nt, k, parts, dt * 1e3, nodes / dt / 1e6, "ok" if ok else "MISMATCH"))
    del leaves
