// host_pipe_probe.hip -- timeline of the host-pointer pipeline: when does each chunk's copy-in, kernel and copy-out
// actually run?  Links libhades252.so for the real permutation kernel (hades252_perm_batch_dev).
//   hipcc -O3 --offload-arch=gfx950 -Iinclude -o build_tools/host_pipe_probe tools/host_pipe_probe.hip \
//         -Lhades252_amd/csrc -lhades252 -Wl,-rpath,$PWD/hades252_amd/csrc
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "hades252.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_copy_out(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

int main(int argc, char **argv) {
    const size_t chunk = argc > 1 ? atol(argv[1]) : (1 << 17);
    const size_t n_chunks = argc > 2 ? atol(argv[2]) : 16;
    const int out_mode = argc > 3 ? atoi(argv[3]) : 1;        // 0 = hipMemcpyAsync D2H, 1 = copy kernel, 2 = no copy-out
    const int kern = argc > 4 ? atoi(argv[4]) : 1;            // 0 = no permutation kernel
    const int prio = argc > 5 ? atoi(argv[5]) : 0;
    const int ogrid = argc > 6 ? atoi(argv[6]) : 256;
    const int quiet = argc > 7 ? atoi(argv[7]) : 0;
    const int kSlots = 6;
    const size_t n = chunk * n_chunks, cb = chunk * 160;
    uint8_t *h, *d;
    CK(hipHostMalloc((void **)&h, n * 160, hipHostMallocPortable | hipHostMallocMapped));
    CK(hipMalloc((void **)&d, cb * kSlots));
    for (size_t i = 0; i < n * 20; i++) ((uint64_t *)h)[i] = (i * 0x9E3779B97F4A7C15ull) >> 3;
    hipStream_t s_in, s_k, s_out;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (prio) {
        CK(hipStreamCreateWithPriority(&s_in, hipStreamNonBlocking, lo));
        CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
        CK(hipStreamCreateWithPriority(&s_out, hipStreamNonBlocking, hi));
    } else {
        CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    }
    printf("chunk %zu states (%.1f MiB), %zu chunks, out_mode %d (copy grid %d), kernel %d, priorities %d (range %d..%d)\n", chunk,
           cb / 1048576.0, n_chunks, out_mode, ogrid, kern, prio, lo, hi);
    std::vector<hipEvent_t> ev(n_chunks * 6);
    for (auto &e : ev) CK(hipEventCreate(&e));
    hipEvent_t t0;
    CK(hipEventCreate(&t0));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, s_in));
        CK(hipStreamWaitEvent(s_k, t0, 0));
        CK(hipStreamWaitEvent(s_out, t0, 0));
        for (size_t c = 0; c < n_chunks; c++) {
            hipEvent_t *e = &ev[c * 6];
            int k = (int)(c % kSlots);
            uint8_t *slot = d + (size_t)k * cb;
            if (c >= (size_t)kSlots) CK(hipEventSynchronize(ev[(c - kSlots) * 6 + 5]));
            CK(hipEventRecord(e[0], s_in));
            CK(hipMemcpyAsync(slot, h + c * cb, cb, hipMemcpyHostToDevice, s_in));
            CK(hipEventRecord(e[1], s_in));
            CK(hipStreamWaitEvent(s_k, e[1], 0));
            CK(hipEventRecord(e[2], s_k));
            if (kern && hades252_perm_batch_dev(slot, chunk, s_k) != 0) return 1;
            CK(hipEventRecord(e[3], s_k));
            CK(hipStreamWaitEvent(s_out, e[3], 0));
            CK(hipEventRecord(e[4], s_out));
            if (out_mode == 0) CK(hipMemcpyAsync(h + c * cb, slot, cb, hipMemcpyDeviceToHost, s_out));
            if (out_mode == 1) hipLaunchKernelGGL(k_copy_out, dim3(ogrid), dim3(256), 0, s_out, (const uint4 *)slot, (uint4 *)(h + c * cb), cb / 16);
            CK(hipEventRecord(e[5], s_out));
        }
        CK(hipDeviceSynchronize());
    }
    float total = 0;
    CK(hipEventElapsedTime(&total, t0, ev[(n_chunks - 1) * 6 + 5]));
    printf("total %.3f ms = %.2f GB/s each way, %.1f Mperm/s\n", total, n * 160 / (total * 1e-3) / 1e9, n / (total * 1e-3) / 1e6);
    if (quiet) return 0;
    printf("chunk |   copy-in start..end   |   kernel start..end    |  copy-out start..end   (ms since start)\n");
    for (size_t c = 0; c < n_chunks; c++) {
        float t[6];
        for (int i = 0; i < 6; i++) CK(hipEventElapsedTime(&t[i], t0, ev[c * 6 + i]));
        printf("%5zu | %8.3f .. %8.3f (%5.3f) | %8.3f .. %8.3f (%5.3f) | %8.3f .. %8.3f (%5.3f)\n", c, t[0], t[1], t[1] - t[0], t[2], t[3],
               t[3] - t[2], t[4], t[5], t[5] - t[4]);
    }
    return 0;
}
