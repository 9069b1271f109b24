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

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
RM = (1 << 256) % P
tag = 15 * RM % P
tag2 = 3 * RM % P
cap = (1 << 64) * RM % P

def sec_device():
    print("== device-resident kernels")
    for logn in (0, 2, 6, 8, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 26):
        n = 1 << logn
        buf = H.gen_b(5 * n, dev)
        for k, name in ((2, "fast"), (3, "coop"), (4, "lanes"), (1, "literal")):
            if k == 1 and logn > 24: continue
            if k == 3 and logn > 22: continue
            if k == 4 and logn > 18: continue
            s = H.ScalarStrategy(k)
            dt = timed(lambda: s.perm(buf), reps=3 if logn > 20 else 20)
            print("n=2^%-2d %-8s %9.3f ms  %8.2f Mperm/s  %7.2f GB/s algorithmic" % (logn, name, dt * 1e3, n / dt / 1e6, 320 * n / dt / 1e9))
        del buf


def sec_host_small():
    print("== host-pointer path (hades252_perm_batch: H2D + kernel + D2H, pageable numpy memory)")
    for logn in (16, 20, 22):
        n = 1 << logn
        host = H.gen_b(5 * n, dev).cpu().numpy().view(np.uint64).reshape(-1).copy()
        s = H.ScalarStrategy()
        dt = timed(lambda: s.perm(host), reps=2)
        print("n=2^%-2d host path %9.3f ms  %8.2f Mperm/s (%.2f GB/s over PCIe each way)" % (logn, dt * 1e3, n / dt / 1e6, 160 * n / dt / 1e9))


def sec_merkle():
    print("== Merkle arity-4 (tag 15, out word 1): bulk levels one parent per lane, last <= 65536 nodes fused in CUs")
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
    dt = dtb = timed(lambda: H.merkle_build(leaves, 4, tag, 1), reps=3)
    tree = H.merkle_build(leaves, 4, tag, 1)
    idx = torch.randint(0, n, (1 << 16,), dtype=torch.int64, device=dev)
    dto = timed(lambda: H.merkle_open(leaves, tree, 4, idx), reps=5)
    print("build (all levels kept) leaves=2^24 %9.3f ms;  2^16 openings (12 levels x 3 siblings) %8.3f ms" % (dt * 1e3, dto * 1e3))
    pths = H.merkle_open(leaves, tree, 4, idx)
    lv = leaves[idx].contiguous()
    roots = H.merkle_verify(lv, idx, pths, 4, tag, 1)
    dtv = timed(lambda: H.merkle_verify(lv, idx, pths, 4, tag, 1), reps=5)
    print("verify 2^16 openings of the 2^24-leaf tree (12 permutations each): %8.3f ms  %8.2f Mperm/s  all roots ok: %s"
          % (dtv * 1e3, 12 * (1 << 16) / dtv / 1e6, bool((roots == tree[-1:]).all())))
    g = torch.Generator(device="cpu")
    g.manual_seed(11)
    for k_upd in (1, 256, 768, 1024, 1 << 14, 1 << 18):
        ui = torch.sort(torch.randint(0, n, (k_upd,), generator=g, dtype=torch.int64))[0].to(dev)
        dtu = timed(lambda: H.merkle_update(leaves, tree, 4, ui, tag, 1), reps=5)
        print("update %7d leaves of the 2^24-leaf tree (their 12 ancestors each, sorted indices): %8.3f ms  (full rebuild %.1f ms)"
              % (k_upd, dtu * 1e3, dtb * 1e3))
    del tree, leaves, pths, lv
    for nt, k in ((4096, 4), (10 ** 4, 4), (1 << 16, 2)):
        per = 4 ** k
        fl = H.gen_b(nt * per, dev)
        sc = torch.empty(max(_lib.lib().hades252_merkle_forest_scratch_bytes(nt, per, 4) // 8, 2), dtype=torch.int64, device=dev)
        dt = timed(lambda: H.merkle_forest(fl, nt, 4, tag, 1, sc), reps=10)
        nodes = nt * (per - 1) // 3
        print("forest of %d arity-4 trees of 4^%d leaves: %8.3f ms  %d perms  %8.2f Mperm/s" % (nt, k, dt * 1e3, nodes, nodes / dt / 1e6))
        del fl
    for n_any, ar in ((3 ** 9, 3), (4 ** 7 * 3, 4), (10 ** 6, 4), (10 ** 6 + 1, 2)):
        lf = H.gen_b(n_any, dev)
        tg = (2 ** ar - 1) * RM % P
        pad = H.merkle_empty_digests(ar, H.merkle_depth(n_any, ar), 0, tg, 1)
        dt = timed(lambda: H.merkle_root(lf, ar, tg, 1, pad=pad), reps=5)
        nodes = sum(H.merkle_level_sizes(n_any, ar))
        print("arity %d, %d leaves (padding table of empty subtrees): %8.3f ms  %d perms  %8.2f Mperm/s" % (ar, n_any, dt * 1e3, nodes, nodes / dt / 1e6))
        del lf
    for logn in (16, 20):
        n = 1 << logn
        leaves = H.gen_b(n, dev)
        dt = timed(lambda: H.merkle_root(leaves, 2, tag2, 1), reps=5)
        print("arity 2 leaves=2^%-2d %9.3f ms  %d perms  %8.2f Mperm/s" % (logn, dt * 1e3, n - 1, (n - 1) / dt / 1e6))


def sec_host():
    print("== host-pointer path, large batches, against the box's own PCIe ceiling")
    # Ceiling: the same bytes moved by bare hipMemcpyAsync from / to page-locked memory in 20 MiB pieces, host->device
    # and device->host at the same time (what a perfect pipeline of this call would be bound by).  The runtime maps
    # streams to DMA engines; an unlucky pair shares one engine and gets half the rate, so several fresh stream pairs
    # are tried and the best one counts.  tools/pcie_probe.hip is the same measurement in plain HIP.
    def pcie_ceiling(nbytes):
        h_in = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        h_out = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        d_in = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        d_out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        piece = 20 << 20
        best = {"h2d": 1e9, "d2h": 1e9, "both": 1e9}
        for pair in range(4):
            s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
            for _ in range(3):
                for mode in ("h2d", "d2h", "both"):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for off in range(0, nbytes, piece):
                        if mode in ("h2d", "both"):
                            with torch.cuda.stream(s1): d_in[off:off + piece].copy_(h_in[off:off + piece], non_blocking=True)
                        if mode in ("d2h", "both"):
                            with torch.cuda.stream(s2): h_out[off:off + piece].copy_(d_out[off:off + piece], non_blocking=True)
                    torch.cuda.synchronize()
                    best[mode] = min(best[mode], time.perf_counter() - t0)
        return best

    # (1) a NATIVE caller (tools/host_path_bench.cpp: plain C++, system HIP runtime -- what a Rust / C host links)
    import json, subprocess
    from hades252_amd import build as hb
    exe = hb.build_host_path_bench(verbose=False)
    for line in subprocess.run([exe, "20", "22", "24"], capture_output=True, text=True).stdout.strip().splitlines():
        r = json.loads(line)
        print("native caller n=2^%-2d page-locked (hades252_host_alloc) %9.3f ms  %8.2f Mperm/s  %.2f GB/s each way = %.1f %% of the ceiling "
              "(both directions at once %.2f GB/s each way; alone H2D %.2f, D2H %.2f);  pageable memory %9.3f ms %8.2f Mperm/s;  bit-exact %s"
              % (r["perms"].bit_length() - 1, r["ms"], r["perms_per_s"] / 1e6, r["gbs_each_way"], 100 * r["frac_of_ceiling"],
                 r["pcie_ceiling_gbs_each_way"], r["h2d_alone_gbs"], r["d2h_alone_gbs"], r["pageable_ms"],
                 r["pageable_perms_per_s"] / 1e6, r["bit_exact_vs_device_path"]))
    # (2) the same library inside THIS PyTorch process (PyTorch's bundled HIP runtime): for the record
    for logn in (20, 22, 24):
        n = 1 << logn
        nbytes = 160 * n
        ceil = pcie_ceiling(nbytes)
        src = H.gen_b(5 * n, dev).cpu().numpy().view(np.uint64).reshape(-1).copy()
        s = H.ScalarStrategy()
        print("n=2^%-2d ceiling: H2D alone %.2f GB/s, D2H alone %.2f GB/s, both at once %.2f GB/s each way (%.3f ms)"
              % (logn, nbytes / ceil["h2d"] / 1e9, nbytes / ceil["d2h"] / 1e9, nbytes / ceil["both"] / 1e9, ceil["both"] * 1e3))
        with H.HostBuffer(n) as hb:
            hb.array[:] = src
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); s.perm(hb.array); ts.append(time.perf_counter() - t0)
            dt = sorted(ts[1:])[len(ts[1:]) // 2]
            print("n=2^%-2d host path, hades252_host_alloc memory   %9.3f ms  %8.2f Mperm/s  %.2f GB/s each way = %.1f %% of the ceiling"
                  % (logn, dt * 1e3, n / dt / 1e6, nbytes / dt / 1e9, 100 * ceil["both"] / dt))
        mine = src.copy()
        H.host_register(mine)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); s.perm(mine); ts.append(time.perf_counter() - t0)
        dt = sorted(ts[1:])[len(ts[1:]) // 2]
        H.host_unregister(mine)
        print("n=2^%-2d host path, hades252_host_register once  %9.3f ms  %8.2f Mperm/s  %.2f GB/s each way = %.1f %% of the ceiling"
              % (logn, dt * 1e3, n / dt / 1e6, nbytes / dt / 1e9, 100 * ceil["both"] / dt))
        plain = src.copy()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); s.perm(plain); ts.append(time.perf_counter() - t0)
        dt = sorted(ts[1:])[len(ts[1:]) // 2]
        print("n=2^%-2d host path, pageable memory (through the staging threads; in-process numbers: see host_path_native.txt) %9.3f ms  %8.2f Mperm/s  %.2f GB/s each way"
              % (logn, dt * 1e3, n / dt / 1e6, nbytes / dt / 1e9))
        for w in (2, 8):
            with H.HostBuffer(n) as hb:
                hb.array[:] = src
                H.perm_multi(hb.array, w, virtual=True)
                t0 = time.perf_counter(); H.perm_multi(hb.array, w, virtual=True); dt = time.perf_counter() - t0
                print("n=2^%-2d perm_batch_multi_ex, %d workers on this one device (pinned) %9.3f ms  %8.2f Mperm/s" % (logn, w, dt * 1e3, n / dt / 1e6))
        del src, mine, plain


def sec_latency():
    print("== small-call latency: host-pointer call (hades252_perm_batch, default dispatch) and device call + sync per kernel")
    for n in (1, 4, 64, 256, 768, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
        host = H.gen_b(5 * n, dev).cpu().numpy().view(np.uint64).reshape(-1).copy()
        s = H.ScalarStrategy(0)
        ts = []
        for _ in range(7):
            ts.append(timed(lambda: s.perm(host), reps=50))
        print("n=%-5d host call (in + kernel + out), default dispatch (%s): median %8.1f us  min %8.1f us"
              % (n, H.kernel_name(0, n), sorted(ts)[3] * 1e6, min(ts) * 1e6))
        buf = H.gen_b(5 * n, dev)
        for k, name in ((4, "lanes (one state per wave)"), (5, "rows (four states per wave)"), (3, "coop (five waves per state)"),
                        (2, "fast (one state per lane)")):
            if (k == 4 and n > 4096) or (k == 5 and n > 16384):
                continue
            sk = H.ScalarStrategy(k)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev = []
            for _ in range(30):
                a.record(); sk.perm(buf); b.record(); torch.cuda.synchronize(); ev.append(a.elapsed_time(b) * 1e3)
            dt = timed(lambda: (sk.perm(buf), torch.cuda.synchronize()), reps=50)
            print("n=%-5d %-30s kernel (HIP events) median %8.1f us  min %8.1f us;  call + sync %8.1f us" % (n, name, sorted(ev)[15], min(ev), dt * 1e6))


def sec_wire():
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

    for lognw in (18, 20):
        nw = 1 << lognw
        st = H.gen_b(5 * nw, dev)
        wires = torch.empty((972, nw, 4), dtype=torch.int64, device=dev)
        dt = timed(lambda: H.perm_witness(st, out=wires), reps=3)
        print("perm_witness (972 gadget wires) n=2^%d states  %8.3f ms  %8.2f Mperm/s  (31 104 B written per state: %.1f GB/s)" % (lognw, dt * 1e3, nw / dt / 1e6, 972 * 32 * nw / dt / 1e9))
        del wires


def sec_perop():
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


def sec_sponge():
    print("== batched fixed-length sponge (rate 4, pad with 1)")
    for length, nmsg in ((3, 1 << 22), (4, 1 << 22), (16, 1 << 20)):
        msgs = H.gen_b(nmsg * length, dev)
        dt = timed(lambda: H.sponge_hash(msgs, length, cap, 1), reps=3)
        perms = nmsg * ((length + 1 + 3) // 4)
        print("len=%-2d n=2^%-2d %8.3f ms  %8.2f Mhash/s  %8.2f Mperm/s" % (length, nmsg.bit_length() - 1, dt * 1e3, nmsg / dt / 1e6, perms / dt / 1e6))

def sec_sponge_var():
    print("== batched variable-length sponge (lengths uniform in 0..32, packed, pad with 1)")
    nmsg = 1 << 21
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    lens = torch.randint(0, 33, (nmsg,), generator=g, dtype=torch.int64)
    offs = torch.cumsum(lens, 0) - lens
    pool = H.gen_b(int(lens.sum().item()) + 1, dev)
    dl, do = lens.to(dev), offs.to(dev)
    perms = int(((lens + 1 + 3) // 4).sum().item())
    wave_max = int(((lens + 1 + 3) // 4).view(-1, 64).max(dim=1).values.sum().item()) * 64
    for srt in (False, True):
        dt = timed(lambda: H.sponge_hash_var(pool, do, dl, cap, 1, sort=srt), reps=5)
        print("n=2^21 ragged, %-28s %8.3f ms  %8.2f Mhash/s  %8.2f M useful perm/s%s"
              % ("sorted by block count first" if srt else "message order", dt * 1e3, nmsg / dt / 1e6, perms / dt / 1e6,
                 "" if srt else "  (%.2f M lane-perm/s issued incl. idle lanes: %.0f %% useful)" % (wave_max / dt / 1e6, 100.0 * perms / wave_max)))
    st = H.SpongeStates(1 << 22, cap)
    blk = H.gen_b((1 << 22) * 4, dev).view(1 << 22, 1, 4, 4)
    dt = timed(lambda: st.absorb(blk), reps=5)
    print("streaming absorb, 2^22 states x 1 block: %8.3f ms  %8.2f Mperm/s" % (dt * 1e3, (1 << 22) / dt / 1e6))


def sec_small():
    print("== small batches of dependent-permutation work: one message / state / query per wave (<= 1024) vs one per lane")
    pool = H.gen_b(4 * 4096 + 8, dev)
    for nmsg, blocks in ((1, 1000), (1, 100), (64, 100), (768, 100), (1024, 100), (1025, 100), (4096, 100), (4097, 100), (16384, 100), (16385, 100)):
        ln = 4 * blocks - 1                                           # + the padding 1 = `blocks` blocks exactly
        offs = (torch.arange(nmsg, dtype=torch.int64) % 7).to(dev)
        lens = torch.full((nmsg,), ln, dtype=torch.int64, device=dev)
        dt = timed(lambda: H.sponge_hash_var(pool, offs, lens, cap, 1), reps=3)
        print("sponge: %5d message(s) x %4d blocks: %9.3f ms = %7.1f us per block  (%s)"
              % (nmsg, blocks, dt * 1e3, dt * 1e6 / blocks, "one message per wave" if nmsg <= 1024 else ("four messages per wave" if nmsg <= 4096 else ("five waves per message" if nmsg <= 16384 else "one message per lane"))))
    for n in (1, 768, 1024, 1025, 4096, 4097, 16384, 16385):
        st = H.SpongeStates(n, cap)
        blk = H.gen_b(n * 50 * 4, dev).view(n, 50, 4, 4)
        dt = timed(lambda: st.absorb(blk), reps=3)
        print("streaming absorb: %5d state(s) x 50 blocks: %9.3f ms = %7.1f us per block" % (n, dt * 1e3, dt * 1e6 / 50))
    n = 1 << 24
    leaves = H.gen_b(n, dev)
    tree = H.merkle_build(leaves, 4, tag, 1)
    for nq in (1, 64, 768, 1024, 1025, 4096, 4097, 16384, 16385, 1 << 16):
        idx = torch.randint(0, n, (nq,), dtype=torch.int64, device=dev)
        pths = H.merkle_open(leaves, tree, 4, idx)
        lv = leaves[idx].contiguous()
        dt = timed(lambda: H.merkle_verify(lv, idx, pths, 4, tag, 1), reps=5)
        print("verify %6d opening(s) of the 2^24-leaf tree (12 dependent permutations each): %9.3f ms" % (nq, dt * 1e3))


if __name__ == "__main__":
    want = sys.argv[1:] or ['device', 'host_small', 'merkle', 'host', 'latency', 'wire', 'perop', 'sponge', 'sponge_var', 'small']
    for nm in want:
        globals()["sec_" + nm]()
        torch.cuda.empty_cache()
