"""Does building the 4 quarter-trees of a 2^24-leaf tree on 4 streams (so that one sub-tree's drains and
latency-bound top levels overlap with another's bulk work) beat the single-stream builder?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hades252_amd import strategy as H, _lib

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
tag = 15 * ((1 << 256) % P) % P
dev = torch.device("cuda", 0)
n = 1 << 24
leaves = H.gen_b(n, dev)

def single():
    return H.merkle4_root(leaves, tag, 1)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], r

for parts in (1, 4, 16):
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    q = n // parts
    scr = [torch.empty(_lib.lib().hades252_merkle4_scratch_bytes(q) // 8, dtype=torch.int64, device=dev) for _ in range(parts)]
    subs = torch.empty((parts, 4), dtype=torch.int64, device=dev)
    def multi():
        main = torch.cuda.current_stream(dev)
        ev0 = torch.cuda.Event(); ev0.record(main)
        for i, s in enumerate(streams):
            s.wait_event(ev0)
            with torch.cuda.stream(s):
                subs[i] = H.merkle4_root(leaves[i * q:(i + 1) * q], tag, 1, scr[i])
            e = torch.cuda.Event(); e.record(s); main.wait_event(e)
        return H.merkle4_root(subs, tag, 1) if parts > 1 else subs[0]
    dt, r = timed(multi)
    print("parts=%2d  %8.3f ms  root %s" % (parts, dt * 1e3, "".join("%016x" % (int(x) & (2**64 - 1)) for x in r.view(-1).cpu().tolist()[::-1])))
dt, r = timed(single)
print("single     %8.3f ms  root %s" % (dt * 1e3, "".join("%016x" % (int(x) & (2**64 - 1)) for x in r.view(-1).cpu().tolist()[::-1])))
