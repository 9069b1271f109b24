// host_path_bench.cpp -- the host-pointer boundary as a native caller sees it (what a Rust `Strategy::perm` binds):
// hades252_host_alloc + hades252_perm_batch from a plain C++ process linked against the system HIP runtime, next to
// this box's bidirectional copy ceiling measured by the same process.  One JSON line per batch size.
//   hipcc -O2 -Iinclude -o build_tools/host_path_bench tools/host_path_bench.cpp -Lhades252_amd/csrc -lhades252 \
//         -Wl,-rpath,'$ORIGIN/../hades252_amd/csrc'
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <algorithm>
#include <chrono>
#include <vector>
#include "hades252.h"

// A buffer the driver has never seen: fresh anonymous pages (a std::vector reused across repetitions is page-locked for
// free from the second call on -- the driver caches the pinning -- which flatters every "pageable" number).
static void *fresh_pages(size_t bytes) {
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(1); }
    return p;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hades252_strerror(r_)); exit(1); } } while (0)

// bare hipMemcpyAsync both ways at once, page-locked memory; best over fresh stream pairs, piece sizes (10 / 20 / 40 MiB)
// and BOTH kinds of page-locked allocation (the runtime's default and the portable + mapped kind hades252_host_alloc
// hands out: on some boxes of the pool device -> host copies into the default kind run at half rate, 27 instead of
// 57 GB/s, and a "ceiling" measured on it alone came out below what the library's pipeline achieved)
static double ceiling_seconds(size_t bytes, double *h2d_alone, double *d2h_alone) {
    uint8_t *d_in, *d_out;
    CK(hipMalloc((void **)&d_in, bytes));
    CK(hipMalloc((void **)&d_out, bytes));
    double best[3] = {1e9, 1e9, 1e9};
    for (unsigned flags : {(unsigned)hipHostMallocDefault, (unsigned)(hipHostMallocPortable | hipHostMallocMapped)}) {
        uint8_t *h_in, *h_out;
        CK(hipHostMalloc((void **)&h_in, bytes, flags));
        CK(hipHostMalloc((void **)&h_out, bytes, flags));
        memset(h_in, 1, bytes);
        memset(h_out, 1, bytes);
        for (int pair = 0; pair < 3; pair++) {
            hipStream_t s1, s2;
            CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
            CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            for (size_t piece : {(size_t)10 << 20, (size_t)20 << 20, (size_t)40 << 20})
                for (int mode = 0; mode < 3; mode++) {
                    CK(hipDeviceSynchronize());
                    double t0 = now();
                    for (size_t off = 0; off < bytes; off += piece) {
                        size_t n = std::min(piece, bytes - off);
                        if (mode != 1) CK(hipMemcpyAsync(d_in + off, h_in + off, n, hipMemcpyHostToDevice, s1));
                        if (mode != 0) CK(hipMemcpyAsync(h_out + off, d_out + off, n, hipMemcpyDeviceToHost, s2));
                    }
                    CK(hipDeviceSynchronize());
                    best[mode] = std::min(best[mode], now() - t0);
                }
            CK(hipStreamDestroy(s1));
            CK(hipStreamDestroy(s2));
        }
        CK(hipHostFree(h_in));
        CK(hipHostFree(h_out));
    }
    CK(hipFree(d_in));
    CK(hipFree(d_out));
    *h2d_alone = best[0];
    *d2h_alone = best[1];
    return best[2];
}

// `callers`: the one-shot host forms of the Merkle root and the sponge (hades252_merkle_root / hades252_sponge_hash)
// against their device-resident twins and the one-way copy rate of the same bytes
static int callers() {
    const uint64_t tag[4] = {0x00000020ffffffdfull, 0x348ddb9d00362421ull, 0x658b26f6c2232750ull, 0x0e5d6e47a2b2d9b1ull};  // 15 R mod p
    const uint64_t cap[4] = {1, 0, 0, 0};
    for (int logn : {16, 20, 24}) {
        const size_t n = (size_t)1 << logn, bytes = n * 32;
        uint64_t *lv;
        HK(hades252_host_alloc((void **)&lv, bytes));
        void *d, *d_scr, *d_root;
        CK(hipMalloc(&d, bytes));
        CK(hipMalloc(&d_scr, hades252_merkle_scratch_bytes(n, 4) + 64));
        CK(hipMalloc(&d_root, 32));
        HK(hades252_gen_b_dev(d, 0, n, 0x4861646573323532ull, nullptr));
        CK(hipMemcpy(lv, d, bytes, hipMemcpyDeviceToHost));
        uint64_t want[4], got[4];
        double td = 1e9, th = 1e9, tp = 1e9, tc = 1e9;
        for (int rep = 0; rep < 7; rep++) {
            CK(hipDeviceSynchronize());
            double t0 = now();
            HK(hades252_merkle_root_dev(d, n, 4, d_scr, hades252_merkle_scratch_bytes(n, 4), tag, 1, d_root, nullptr));
            CK(hipDeviceSynchronize());
            td = std::min(td, now() - t0);
        }
        for (int rep = 0; rep < 5; rep++) {
            double t0 = now();
            CK(hipMemcpy(d, lv, bytes, hipMemcpyHostToDevice));
            tc = std::min(tc, now() - t0);
        }
        CK(hipMemcpy(want, d_root, 32, hipMemcpyDeviceToHost));
        for (int rep = 0; rep < 5; rep++) {
            double t0 = now();
            HK(hades252_merkle_root(lv, n, 4, tag, 1, nullptr, got));
            th = std::min(th, now() - t0);
        }
        bool ok = memcmp(want, got, 32) == 0;
        std::vector<uint64_t> plain(lv, lv + n * 4);              // ordinary memory the driver already knows (a reused Vec)
        for (int rep = 0; rep < 3; rep++) {
            double t0 = now();
            HK(hades252_merkle_root(plain.data(), n, 4, tag, 1, nullptr, got));
            tp = std::min(tp, now() - t0);
        }
        ok = ok && memcmp(want, got, 32) == 0;
        printf("merkle root, arity 4, 2^%d leaves in HOST memory: page-locked %.3f ms, pageable %.3f ms;  device-resident %.3f ms, "
               "bare upload of the leaves %.3f ms (%.1f GB/s);  roots equal: %s\n",
               logn, th * 1e3, tp * 1e3, td * 1e3, tc * 1e3, bytes / tc / 1e9, ok ? "yes" : "NO");
        fflush(stdout);
        HK(hades252_host_free(lv));
        CK(hipFree(d)); CK(hipFree(d_scr)); CK(hipFree(d_root));
    }
    for (int logn : {10, 16, 22}) {
        const size_t n = (size_t)1 << logn, len = 4, bytes = n * len * 32;
        uint64_t *ms, *dg;
        HK(hades252_host_alloc((void **)&ms, bytes));
        HK(hades252_host_alloc((void **)&dg, n * 32));
        void *d, *d_dig;
        CK(hipMalloc(&d, bytes));
        CK(hipMalloc(&d_dig, n * 32));
        HK(hades252_gen_b_dev(d, 0, n * len, 0x4861646573323532ull, nullptr));
        CK(hipMemcpy(ms, d, bytes, hipMemcpyDeviceToHost));
        double td = 1e9, th = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipDeviceSynchronize());
            double t0 = now();
            HK(hades252_sponge_hash_dev(d, n, len, cap, 1, d_dig, nullptr));
            CK(hipDeviceSynchronize());
            td = std::min(td, now() - t0);
            t0 = now();
            HK(hades252_sponge_hash(ms, n, len, cap, 1, dg));
            th = std::min(th, now() - t0);
        }
        std::vector<uint64_t> want(n * 4);
        CK(hipMemcpy(want.data(), d_dig, n * 32, hipMemcpyDeviceToHost));
        printf("sponge, 2^%d messages of 4 scalars (2 permutations each) in HOST memory: %.3f ms = %.1f M hashes/s (%.1f GB/s in);  "
               "device-resident %.3f ms;  digests equal: %s\n",
               logn, th * 1e3, n / th / 1e6, bytes / th / 1e9, td * 1e3, memcmp(want.data(), dg, n * 32) == 0 ? "yes" : "NO");
        fflush(stdout);
        HK(hades252_host_free(ms)); HK(hades252_host_free(dg));
        CK(hipFree(d)); CK(hipFree(d_dig));
    }
    // last: the Merkle
    // root of leaves in ordinary memory THE DRIVER HAS NEVER SEEN -- fresh pages for every repetition
    for (int logn : {20, 24}) {
        const size_t n = (size_t)1 << logn, bytes = n * 32;
        void *d;
        CK(hipMalloc(&d, bytes));
        HK(hades252_gen_b_dev(d, 0, n, 0x4861646573323532ull, nullptr));
        uint64_t want[4], got[4];
        std::vector<uint64_t> keep(n * 4);
        CK(hipMemcpy(keep.data(), d, bytes, hipMemcpyDeviceToHost));
        HK(hades252_merkle_root(keep.data(), n, 4, tag, 1, nullptr, want));
        double tp = 1e9;
        bool ok = true;
        for (int rep = 0; rep < 3; rep++) {
            uint64_t *plain = (uint64_t *)fresh_pages(bytes);
            memcpy(plain, keep.data(), bytes);
            double t0 = now();
            HK(hades252_merkle_root(plain, n, 4, tag, 1, nullptr, got));
            tp = std::min(tp, now() - t0);
            ok = ok && memcmp(want, got, 32) == 0;
            munmap(plain, bytes);
        }
        printf("merkle root, arity 4, 2^%d leaves in ordinary memory on FRESH pages every call: %.3f ms;  roots equal: %s\n", logn, tp * 1e3,
               ok ? "yes" : "NO");
        CK(hipFree(d));
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && strcmp(argv[1], "callers") == 0) return callers();
    std::vector<int> logs;
    for (int i = 1; i < argc; i++) logs.push_back(atoi(argv[i]));
    if (logs.empty()) logs = {20, 22, 24};
    for (int logn : logs) {
        const size_t n = (size_t)1 << logn, bytes = n * 160;
        double h2d, d2h;
        const double ceil = ceiling_seconds(bytes, &h2d, &d2h);
        uint64_t *st;
        HK(hades252_host_alloc((void **)&st, bytes));
        void *d;
        CK(hipMalloc(&d, bytes));
        HK(hades252_gen_b_dev(d, 0, 5 * n, 0x4861646573323532ull, nullptr));
        CK(hipMemcpy(st, d, bytes, hipMemcpyDeviceToHost));
        // reference result computed on device, for a full compare of the host path's output
        HK(hades252_perm_batch_dev(d, n, nullptr));
        std::vector<uint64_t> expect(n * 20);
        CK(hipMemcpy(expect.data(), d, bytes, hipMemcpyDeviceToHost));
        std::vector<double> ts;
        bool ok = true;
        for (int rep = 0; rep < 7; rep++) {
            if (rep > 0) {                                            // restore the inputs
                HK(hades252_gen_b_dev(d, 0, 5 * n, 0x4861646573323532ull, nullptr));
                CK(hipMemcpy(st, d, bytes, hipMemcpyDeviceToHost));
            }
            double t0 = now();
            HK(hades252_perm_batch(st, n));
            ts.push_back(now() - t0);
            if (rep == 0 || rep == 6) ok = ok && memcmp(st, expect.data(), bytes) == 0;
        }
        std::sort(ts.begin() + 1, ts.end());
        const double med = ts[1 + (ts.size() - 1) / 2];
        // the same batch from ordinary (pageable) memory the driver has never seen: fresh pages for every repetition
        double tp = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            uint64_t *plain = (uint64_t *)fresh_pages(bytes);
            HK(hades252_gen_b_dev(d, 0, 5 * n, 0x4861646573323532ull, nullptr));
            CK(hipMemcpy(plain, d, bytes, hipMemcpyDeviceToHost));
            double t0 = now();
            HK(hades252_perm_batch(plain, n));
            tp = std::min(tp, now() - t0);
            ok = ok && memcmp(plain, expect.data(), bytes) == 0;
            munmap(plain, bytes);
        }
        printf("{\"perms\": %zu, \"ms\": %.3f, \"perms_per_s\": %.4g, \"gbs_each_way\": %.2f, \"pcie_ceiling_gbs_each_way\": %.2f, "
               "\"frac_of_ceiling\": %.3f, \"h2d_alone_gbs\": %.2f, \"d2h_alone_gbs\": %.2f, \"first_call_ms\": %.3f, "
               "\"pageable_ms\": %.3f, \"pageable_perms_per_s\": %.4g, \"bit_exact_vs_device_path\": %s}\n",
               n, med * 1e3, n / med, bytes / med / 1e9, bytes / ceil / 1e9, ceil / med, bytes / h2d / 1e9, bytes / d2h / 1e9,
               ts[0] * 1e3, tp * 1e3, n / tp, ok ? "true" : "false");
        fflush(stdout);
        HK(hades252_host_free(st));
        CK(hipFree(d));
    }
    return 0;
}
