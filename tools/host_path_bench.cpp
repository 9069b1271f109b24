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
#include <algorithm>
#include <chrono>
#include <vector>
#include "hades252.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hades252_strerror(r_)); exit(1); } } while (0)

// bare hipMemcpyAsync both ways at once, 20 MiB pieces, page-locked memory; best of several fresh stream pairs
static double ceiling_seconds(size_t bytes, double *h2d_alone, double *d2h_alone) {
    uint8_t *h_in, *h_out, *d_in, *d_out;
    CK(hipHostMalloc((void **)&h_in, bytes, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&h_out, bytes, hipHostMallocDefault));
    CK(hipMalloc((void **)&d_in, bytes));
    CK(hipMalloc((void **)&d_out, bytes));
    memset(h_in, 1, bytes);
    memset(h_out, 1, bytes);
    const size_t piece = (size_t)20 << 20;
    double best[3] = {1e9, 1e9, 1e9};
    for (int pair = 0; pair < 4; pair++) {
        hipStream_t s1, s2;
        CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        for (int rep = 0; rep < 3; rep++)
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
    CK(hipHostFree(h_in)); CK(hipHostFree(h_out)); CK(hipFree(d_in)); CK(hipFree(d_out));
    *h2d_alone = best[0];
    *d2h_alone = best[1];
    return best[2];
}

int main(int argc, char **argv) {
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
        // the same batch from ordinary (pageable) memory: page-locked and released inside the call
        std::vector<uint64_t> plain(n * 20);
        double tp = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            HK(hades252_gen_b_dev(d, 0, 5 * n, 0x4861646573323532ull, nullptr));
            CK(hipMemcpy(plain.data(), d, bytes, hipMemcpyDeviceToHost));
            double t0 = now();
            HK(hades252_perm_batch(plain.data(), n));
            tp = std::min(tp, now() - t0);
        }
        ok = ok && memcmp(plain.data(), expect.data(), bytes) == 0;
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
