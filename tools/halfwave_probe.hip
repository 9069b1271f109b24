// halfwave_probe.hip -- does a wave64 with only 32 (or 16) active lanes issue its VALU instructions faster on gfx950?
// If it did, the lone-wave regime of k_perm_fast (16 385 .. 65 536 states: one wave per SIMD, 168-175 us) could run
// half-filled waves on twice as many SIMDs.  One wave per SIMD (1024 blocks of 64 threads... 4 waves per CU), a chain
// of dependent 64-bit multiply-adds and 32-bit ops like the kernel's, active lanes = 64 / 32 / 16.
//   hipcc -O3 --offload-arch=gfx950 -o build_tools/halfwave_probe tools/halfwave_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int ACTIVE>
__global__ void __launch_bounds__(256) k_chain(int64_t *out, int iters, int32_t b0, int32_t b1) {
    const int lane = threadIdx.x & 63;
    if (lane >= ACTIVE) return;
    int64_t acc = threadIdx.x;
    int32_t x = (int32_t)blockIdx.x + 3, y = b1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            acc += (int64_t)x * b0;                  // v_mad_i64_i32, dependent
            asm volatile("" : "+v"(acc));
            acc += (int64_t)y * b1;
            asm volatile("" : "+v"(acc));
            x = ((int32_t)acc & 0x1fffffff) + k;     // 32-bit ops, dependent
            acc >>= 29;
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + x;
}

int main() {
    int64_t *d;
    CK(hipMalloc(&d, (size_t)1024 * 256 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;                          // 2000 x 16 x 5 = 160 k instructions per wave
    for (int blocks : {256, 512}) {                  // 256 blocks x 4 waves = one wave per SIMD; 512 = two
        for (int active : {64, 32, 16}) {
            std::vector<float> ts;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                if (active == 64) hipLaunchKernelGGL(k_chain<64>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345, 678);
                if (active == 32) hipLaunchKernelGGL(k_chain<32>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345, 678);
                if (active == 16) hipLaunchKernelGGL(k_chain<16>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345, 678);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            printf("%d blocks of 4 waves, %2d active lanes per wave: %8.3f ms for %d instructions per wave = %.2f ns per instruction\n", blocks,
                   active, ts[2], iters * 16 * 5, ts[2] * 1e6 / (iters * 16 * 5.0));
        }
    }
    return 0;
}
