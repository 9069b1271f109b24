// pcie_probe.hip -- what the host link of THIS box sustains, by every mechanism the host-pointer path could use.
//   hipcc -O3 --offload-arch=gfx950 -o build_tools/pcie_probe tools/pcie_probe.hip && ./build_tools/pcie_probe
// (1) hipMemcpyAsync (SDMA engines) from / to page-locked memory: one direction alone, both directions at once, with
//     1, 2 and 4 streams per direction (the runtime maps streams to engines);
// (2) a kernel that reads / writes the page-locked host memory directly (what the <= 256-state path already does).
// The host-pointer entry points are reported against the best bidirectional figure found here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) dst[i] = src[i];
}
// both directions inside ONE kernel: in -> dev_in and dev_out -> out
__global__ void k_copy2(const uint4 *__restrict__ h_in, uint4 *__restrict__ d_in, const uint4 *__restrict__ d_out,
                        uint4 *__restrict__ h_out, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) {
        uint4 a = h_in[i], b = d_out[i];
        d_in[i] = a;
        h_out[i] = b;
    }
}

// VALU-only busy kernel: `iters` dependent multiply-adds per thread (no memory traffic), to stand in for k_perm_fast
__global__ void k_busy(uint32_t *sink, int iters) {
    uint64_t a = threadIdx.x + 1, b = blockIdx.x + 3;
    for (int i = 0; i < iters; i++) a = a * b + (a >> 7);
    if (a == 0x123456789abcdefull) sink[0] = (uint32_t)a;
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 640) << 20;     // 640 MiB = 2^22 states
    const size_t piece = (size_t)20 << 20;                                   // 20 MiB pieces (2^17 states)
    uint8_t *h_in, *h_out, *d_in, *d_out;
    CK(hipHostMalloc((void **)&h_in, bytes, hipHostMallocPortable | hipHostMallocMapped));
    CK(hipHostMalloc((void **)&h_out, bytes, hipHostMallocPortable | hipHostMallocMapped));
    CK(hipMalloc((void **)&d_in, bytes));
    CK(hipMalloc((void **)&d_out, bytes));
    for (size_t i = 0; i < bytes; i += 4096) h_in[i] = (uint8_t)i, h_out[i] = 1;
    CK(hipMemset(d_out, 7, bytes));
    std::vector<hipStream_t> sin(4), sout(4);
    for (auto &s : sin) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto &s : sout) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    printf("pcie_probe: %zu MiB each way, %zu MiB pieces, page-locked host memory\n", bytes >> 20, piece >> 20);
    for (int ns = 1; ns <= 4; ns *= 2) {
        for (int mode = 0; mode < 3; mode++) {               // 0 h2d, 1 d2h, 2 both
            double best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipDeviceSynchronize());
                double t0 = now();
                size_t k = 0;
                for (size_t off = 0; off < bytes; off += piece, k++) {
                    size_t n = bytes - off < piece ? bytes - off : piece;
                    if (mode != 1) CK(hipMemcpyAsync(d_in + off, h_in + off, n, hipMemcpyHostToDevice, sin[k % ns]));
                    if (mode != 0) CK(hipMemcpyAsync(h_out + off, d_out + off, n, hipMemcpyDeviceToHost, sout[k % ns]));
                }
                CK(hipDeviceSynchronize());
                double dt = now() - t0;
                if (dt < best) best = dt;
            }
            printf("hipMemcpyAsync %d stream(s) per direction, %-4s: %7.3f ms  %6.2f GB/s each way%s\n", ns,
                   mode == 0 ? "h2d" : mode == 1 ? "d2h" : "both", best * 1e3, bytes / best / 1e9,
                   mode == 2 ? "  <- bidirectional" : "");
        }
    }
    uint8_t *hd_in, *hd_out;
    CK(hipHostGetDevicePointer((void **)&hd_in, h_in, 0));
    CK(hipHostGetDevicePointer((void **)&hd_out, h_out, 0));
    for (int grid : {256, 1024, 4096}) {
        for (int mode = 0; mode < 4; mode++) {               // 0 h2d, 1 d2h, 2 both (two kernels, two streams), 3 both (one kernel)
            double best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipDeviceSynchronize());
                double t0 = now();
                if (mode == 0 || mode == 2)
                    hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, sin[0], (const uint4 *)hd_in, (uint4 *)d_in, bytes / 16);
                if (mode == 1 || mode == 2)
                    hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, sout[0], (const uint4 *)d_out, (uint4 *)hd_out, bytes / 16);
                if (mode == 3)
                    hipLaunchKernelGGL(k_copy2, dim3(grid), dim3(256), 0, sin[0], (const uint4 *)hd_in, (uint4 *)d_in,
                                       (const uint4 *)d_out, (uint4 *)hd_out, bytes / 16);
                CK(hipDeviceSynchronize());
                double dt = now() - t0;
                if (dt < best) best = dt;
            }
            printf("kernel copy, grid %4d x 256, %-18s: %7.3f ms  %6.2f GB/s each way\n", grid,
                   mode == 0 ? "h2d" : mode == 1 ? "d2h" : mode == 2 ? "both (2 kernels)" : "both (1 kernel)",
                   best * 1e3, bytes / best / 1e9);
        }
    }
    // (3) the pipeline of the host-pointer path, emulated: copy-in stream -> kernel stream -> copy-out stream chained by
    //     events over 4 slots, the kernel replaced by a VALU-only busy kernel of a chosen size (0 = no kernel at all)
    uint32_t *sink;
    CK(hipMalloc((void **)&sink, 4));
    hipStream_t sk;
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    const int kSlots = 4;
    hipEvent_t ein[kSlots], ek[kSlots], eout[kSlots];
    for (int i = 0; i < kSlots; i++) {
        CK(hipEventCreateWithFlags(&ein[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ek[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&eout[i], hipEventDisableTiming));
    }
    // calibrate the busy kernel: time of 3072 blocks x 256 threads x 20000 iterations
    double t_busy;
    {
        hipLaunchKernelGGL(k_busy, dim3(3072), dim3(256), 0, sk, sink, 20000);
        CK(hipDeviceSynchronize());
        double t0 = now();
        hipLaunchKernelGGL(k_busy, dim3(3072), dim3(256), 0, sk, sink, 20000);
        CK(hipDeviceSynchronize());
        t_busy = now() - t0;
        printf("busy kernel: 3072 blocks x 256 threads x 20000 iterations = %.3f ms\n", t_busy * 1e3);
    }
    for (size_t pc : {(size_t)10 << 20, (size_t)20 << 20, (size_t)40 << 20}) {
        for (int kmode = 0; kmode < 3; kmode++) {          // 0 none, 1 busy kernel ~0.6 x the piece's copy time, 2 tiny kernel
            double copy_t = (double)pc / 47e9;
            int iters = kmode == 1 ? (int)(20000 * (0.6 * copy_t / t_busy)) : 100;
            int grid = kmode == 1 ? 3072 : 1;
            double best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipDeviceSynchronize());
                double t0 = now();
                size_t c = 0;
                for (size_t off = 0; off < bytes; off += pc, c++) {
                    int k = (int)(c % kSlots);
                    size_t n = bytes - off < pc ? bytes - off : pc;
                    uint8_t *d = d_in + (size_t)k * pc;
                    if (c >= (size_t)kSlots) CK(hipStreamWaitEvent(sin[0], eout[k], 0));
                    CK(hipMemcpyAsync(d, h_in + off, n, hipMemcpyHostToDevice, sin[0]));
                    CK(hipEventRecord(ein[k], sin[0]));
                    CK(hipStreamWaitEvent(sk, ein[k], 0));
                    if (kmode) hipLaunchKernelGGL(k_busy, dim3(grid), dim3(256), 0, sk, sink, iters);
                    CK(hipEventRecord(ek[k], sk));
                    CK(hipStreamWaitEvent(sout[0], ek[k], 0));
                    CK(hipMemcpyAsync(h_out + off, d, n, hipMemcpyDeviceToHost, sout[0]));
                    CK(hipEventRecord(eout[k], sout[0]));
                }
                CK(hipDeviceSynchronize());
                double dt = now() - t0;
                if (dt < best) best = dt;
            }
            printf("pipeline emulation, %2zu MiB pieces, %-34s: %7.3f ms  %6.2f GB/s each way\n", pc >> 20,
                   kmode == 0 ? "no kernel" : kmode == 1 ? "all-CU busy kernel (0.6 x copy time)" : "one-block kernel", best * 1e3,
                   bytes / best / 1e9);
        }
    }
    return 0;
}
