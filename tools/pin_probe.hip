// pin_probe.hip -- what page-locking ordinary memory costs on this box, and whether it overlaps with DMA (round 4).
// hades252_perm_batch on a pageable buffer has to lock it (or copy it); this probe times hipHostRegister / Unregister of
// 640 MiB whole and in 32 MiB slices, alone and beside a bidirectional DMA stream from another page-locked buffer, and
// a CPU memcpy into page-locked memory for comparison (1 and 4 threads).
//   hipcc -O2 -std=c++17 -pthread -o build_tools/pin_probe tools/pin_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t bytes = (size_t)640 << 20, slice = (size_t)32 << 20;
    uint8_t *plain = (uint8_t *)aligned_alloc(4096, bytes);
    memset(plain, 1, bytes);
    uint8_t *pinned, *d_a, *d_b;
    CK(hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault));
    memset(pinned, 2, bytes);
    CK(hipMalloc((void **)&d_a, bytes));
    CK(hipMalloc((void **)&d_b, bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        CK(hipHostRegister(plain, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipHostUnregister(plain));
        double t2 = now();
        printf("whole buffer (640 MiB): register %.3f ms (%.1f GB/s), unregister %.3f ms (%.1f GB/s)\n", (t1 - t0) * 1e3,
               bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        for (size_t o = 0; o < bytes; o += slice) CK(hipHostRegister(plain + o, slice, hipHostRegisterDefault));
        double t1 = now();
        for (size_t o = 0; o < bytes; o += slice) CK(hipHostUnregister(plain + o));
        double t2 = now();
        printf("32 MiB slices: register %.3f ms (%.1f GB/s), unregister %.3f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9,
               (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9);
    }
    // DMA alone: both directions at once from the page-locked buffer, 20 MiB pieces
    auto dma = [&]() {
        const size_t piece = (size_t)20 << 20;
        for (size_t o = 0; o < bytes; o += piece) {
            CK(hipMemcpyAsync(d_a + o, pinned + o, piece, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync(pinned + o, d_b + o, piece, hipMemcpyDeviceToHost, s2));
        }
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
    };
    dma();
    double t0 = now();
    dma();
    double t_dma = now() - t0;
    printf("DMA alone, 640 MiB each way at once: %.3f ms (%.1f GB/s each way)\n", t_dma * 1e3, bytes / t_dma / 1e9);
    // register / unregister slices on a helper thread beside the DMA
    for (int rep = 0; rep < 2; rep++) {
        double t_reg = 0, t_unreg = 0;
        std::thread helper([&]() {
            double a = now();
            for (size_t o = 0; o < bytes; o += slice) CK(hipHostRegister(plain + o, slice, hipHostRegisterDefault));
            double b = now();
            for (size_t o = 0; o < bytes; o += slice) CK(hipHostUnregister(plain + o));
            t_reg = b - a;
            t_unreg = now() - b;
        });
        double a = now();
        dma();
        dma();
        double t_d = now() - a;
        helper.join();
        printf("beside each other: 2 x DMA %.3f ms (alone %.3f); slices register %.3f ms, unregister %.3f ms\n", t_d * 1e3, 2 * t_dma * 1e3,
               t_reg * 1e3, t_unreg * 1e3);
    }
    // CPU memcpy of the same bytes into page-locked memory (the staging alternative)
    for (int threads : {1, 4, 8}) {
        double a = now();
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; t++)
            ts.emplace_back([&, t]() { memcpy(pinned + bytes / threads * t, plain + bytes / threads * t, bytes / threads); });
        for (auto &t : ts) t.join();
        double d = now() - a;
        printf("CPU memcpy pageable -> page-locked, %d thread(s): %.3f ms (%.1f GB/s)\n", threads, d * 1e3, bytes / d / 1e9);
    }
    return 0;
}
