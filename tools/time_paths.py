"""Development / DESIGN.md numbers: host-pointer (PCIe-inclusive) rate, Merkle tree build time,
literal-vs-fast kernel, per-size throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hades252_amd import strategy as H, _lib

dev = torch.device("cuda", 0)

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

print("== device-resident kernels")
for logn in (0, 6, 10, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26):
    n = 1 << logn
    buf = H.gen_b(5 * n, dev)
    for k, name in ((2, "fast"), (3, "coop"), (1, "literal")):
        if k == 1 and logn > 24: continue
        if k == 3 and logn > 22: continue
        s = H.ScalarStrategy(k)
        dt = timed(lambda: s.perm(buf), reps=3 if logn > 20 else 20)
        print("n=2^%-2d %-8s %9.3f ms  %8.2f Mperm/s  %7.2f GB/s algorithmic" % (logn, name, dt * 1e3, n / dt / 1e6, 320 * n / dt / 1e9))
    del buf

print("== host-pointer path (hades252_perm_batch: H2D + kernel + D2H, pageable numpy memory)")
for logn in (16, 20, 22):
    n = 1 << logn
    host = H.gen_b(5 * n, dev).cpu().numpy().view(np.uint64).reshape(-1).copy()
    s = H.ScalarStrategy()
    dt = timed(lambda: s.perm(host), reps=2)
    print("n=2^%-2d host path %9.3f ms  %8.2f Mperm/s (%.2f GB/s over PCIe each way)" % (logn, dt * 1e3, n / dt / 1e6, 160 * n / dt / 1e9))

print("== Merkle arity-4 (tag 15, out word 1): bulk levels one parent per lane, last <= 65536 nodes fused in CUs")
tag = 15 * ((1 << 256) % 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001) % 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
for logn in (8, 12, 16, 18, 20, 24):
    n = 1 << logn
    leaves = H.gen_b(n, dev)
    scratch = torch.empty(_lib.lib().hades252_merkle4_scratch_bytes(n) // 8, dtype=torch.int64, device=dev)
    dt = timed(lambda: H.merkle4_root(leaves, tag, 1, scratch), reps=5)
    nodes = (n - 1) // 3
    root = H.merkle4_root(leaves, tag, 1, scratch).cpu().numpy().view(np.uint64)
    print("leaves=2^%-2d %9.3f ms  %d perms  %8.2f Mperm/s  root %s" % (logn, dt * 1e3, nodes, nodes / dt / 1e6, "".join("%016x" % int(x) for x in root[::-1])))

n = 1 << 24
leaves = H.gen_b(n, dev)
scratch = torch.empty(_lib.lib().hades252_merkle4_scratch_bytes(n) // 8, dtype=torch.int64, device=dev)
ts = []
for _ in range(7):
    ts.append(timed(lambda: H.merkle4_root(leaves, tag, 1, scratch), reps=3))
print("leaves=2^24 root, 7 x 3 runs: min %.3f ms  median %.3f ms" % (min(ts) * 1e3, sorted(ts)[3] * 1e3))
dt = timed(lambda: H.merkle_build(leaves, 4, tag, 1), reps=3)
tree = H.merkle_build(leaves, 4, tag, 1)
idx = torch.randint(0, n, (1 << 16,), dtype=torch.int64, device=dev)
dto = timed(lambda: H.merkle_open(leaves, tree, 4, idx), reps=5)
print("build (all levels kept) leaves=2^24 %9.3f ms;  2^16 openings (12 levels x 3 siblings) %8.3f ms" % (dt * 1e3, dto * 1e3))
del tree, leaves
tag2 = 3 * ((1 << 256) % 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001) % 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
for logn in (16, 20):
    n = 1 << logn
    leaves = H.gen_b(n, dev)
    dt = timed(lambda: H.merkle_root(leaves, 2, tag2, 1), reps=5)
    print("arity 2 leaves=2^%-2d %9.3f ms  %d perms  %8.2f Mperm/s" % (logn, dt * 1e3, n - 1, (n - 1) / dt / 1e6))

print("== host-pointer path, large batches (in-place page-locking when HADES252_HOST_PIN != 0)")
for logn in (20, 22, 24):
    n = 1 << logn
    host = H.gen_b(5 * n, dev).cpu().numpy().view(np.uint64).reshape(-1).copy()
    s = H.ScalarStrategy()
    dt = timed(lambda: s.perm(host), reps=2)
    print("n=2^%-2d host path %9.3f ms  %8.2f Mperm/s (%.2f GB/s each way)" % (logn, dt * 1e3, n / dt / 1e6, 160 * n / dt / 1e9))

print("== small-call latency, host-pointer path (pooled stream + buffer)")
for n in (1, 64, 4096, 16384):
    host = H.gen_b(5 * n, dev).cpu().numpy().view(np.uint64).reshape(-1).copy()
    for k, name in ((0, "default"), (2, "fast")):
        s = H.ScalarStrategy(k)
        if k == 2:
            # host path always uses the default dispatch; time the device-resident call + sync instead
            buf = H.gen_b(5 * n, dev)
            dt = timed(lambda: (s.perm(buf), torch.cuda.synchronize()), reps=50)
            print("n=%-5d device call + sync, per-lane kernel %8.1f us" % (n, dt * 1e6))
        else:
            dt = timed(lambda: s.perm(host), reps=50)
            print("n=%-5d host call (H2D + kernel + D2H)      %8.1f us" % (n, dt * 1e6))

print("== wire format on device (BlsScalar::from_bytes / to_bytes), per-round trace")
n = 1 << 24
limbs = H.gen_b(n, dev)
canon = H.to_bytes(limbs)
out = torch.empty_like(limbs)
dt = timed(lambda: H.to_bytes(limbs, out), reps=5)
print("to_bytes   n=2^24 scalars %8.3f ms  %8.2f G scalars/s  %7.1f GB/s (64 B/scalar)" % (dt * 1e3, n / dt / 1e9, 64 * n / dt / 1e9))
dt = timed(lambda: H.from_bytes(canon, out), reps=5)
print("from_bytes n=2^24 scalars %8.3f ms  %8.2f G scalars/s  %7.1f GB/s (64 B/scalar)" % (dt * 1e3, n / dt / 1e9, 64 * n / dt / 1e9))
for lognt in (18, 20):
    nt = 1 << lognt
    st = H.gen_b(5 * nt, dev)
    trace = torch.empty((67, nt, 5, 4), dtype=torch.int64, device=dev)
    for k, name in ((2, "fast"), (1, "literal")):
        if k == 1 and lognt > 18: continue
        dt = timed(lambda: H.perm_trace(st, kernel=k, out=trace), reps=3)
        print("perm_trace %-7s n=2^%d states  %8.3f ms  %8.2f Mperm/s  (67 x 160 B written per state: %.1f GB/s)" % (name, lognt, dt * 1e3, nt / dt / 1e6, 67 * 160 * nt / dt / 1e9))
    del trace

nw = 1 << 18
st = H.gen_b(5 * nw, dev)
wires = torch.empty((972, nw, 4), dtype=torch.int64, device=dev)
dt = timed(lambda: H.perm_witness(st, out=wires), reps=3)
print("perm_witness (972 gadget wires) n=2^18 states  %8.3f ms  %8.2f Mperm/s  (31 104 B written per state: %.1f GB/s)" % (dt * 1e3, nw / dt / 1e6, 972 * 32 * nw / dt / 1e9))
del wires

print("== the trait's per-operation methods, batched (n = 2^22 states / 2^24 scalars)")
n = 1 << 22
stt = H.gen_b(5 * n, dev)
sc = H.gen_b(1 << 24, dev)
strat = H.ScalarStrategy()
for name, fn, units in (("add_round_key", lambda: strat.add_round_key(H.RoundConstantsIter(7), stt), n),
                        ("mul_matrix", lambda: strat.mul_matrix(H.RoundConstantsIter(), stt), n),
                        ("apply_full_round", lambda: strat.apply_full_round(H.RoundConstantsIter(0), stt), n),
                        ("apply_partial_round", lambda: strat.apply_partial_round(H.RoundConstantsIter(20), stt), n),
                        ("quintic_s_box", lambda: strat.quintic_s_box(sc), 1 << 24)):
    dt = timed(fn, reps=5)
    print("%-20s %8.3f ms  %8.2f G units/s" % (name, dt * 1e3, units / dt / 1e9))
del stt, sc

print("== batched fixed-length sponge (rate 4, pad with 1)")
cap = (1 << 64) * ((1 << 256) % 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001) % 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
for length, nmsg in ((3, 1 << 22), (4, 1 << 22), (16, 1 << 20)):
    msgs = H.gen_b(nmsg * length, dev)
    dt = timed(lambda: H.sponge_hash(msgs, length, cap, 1), reps=3)
    perms = nmsg * ((length + 1 + 3) // 4)
    print("len=%-2d n=2^%-2d %8.3f ms  %8.2f Mhash/s  %8.2f Mperm/s" % (length, nmsg.bit_length() - 1, dt * 1e3, nmsg / dt / 1e6, perms / dt / 1e6))
print("== batched variable-length sponge (lengths uniform in 0..32, packed, pad with 1)")
nmsg = 1 << 21
g = torch.Generator(device="cpu"); g.manual_seed(1)
lens = torch.randint(0, 33, (nmsg,), generator=g, dtype=torch.int64)
offs = torch.cumsum(lens, 0) - lens
pool = H.gen_b(int(lens.sum().item()) + 1, dev)
dl, do = lens.to(dev), offs.to(dev)
dt = timed(lambda: H.sponge_hash_var(pool, do, dl, cap, 1), reps=3)
perms = int(((lens + 1 + 3) // 4).sum().item())
wave_max = int(((lens + 1 + 3) // 4).view(-1, 256).max(dim=1).values.sum().item()) * 256
print("n=2^21 ragged %8.3f ms  %8.2f Mhash/s  %8.2f M useful perm/s  (%.2f M lane-perm/s incl. masked lanes)" % (dt * 1e3, nmsg / dt / 1e6, perms / dt / 1e6, wave_max / dt / 1e6))
