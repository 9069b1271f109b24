// lonewave_probe.hip -- A/B of the lone-wave instantiation of the throughput kernel.
// MEASURED AND NOT BUILT (profiles/r5/lonewave_probe.txt): the variants are a patch of hades_fast.hpp, not part of it --
//   git apply tools/lonewave_variants.patch && <the hipcc line below> && git checkout hades252_amd/csrc/hades_fast.hpp
//
// Between 2^14 and 2^17 states the grid of k_perm_fast puts at most ONE wave on a SIMD (65 536 states = 1 024 waves = 1 024
// SIMDs), and a lone wave issued one instruction per 4.97 cycles against 4.15 under saturation.
// The variants below are the same device code (hades_fast.hpp, template parameter V of fast_perm) with
//   bit 0   the limb products of every S-box column on a second accumulator, interleaved with the reduction terms
//   bit 1   the linear layer column-major: five independent row accumulators interleaved (no extra instruction)
//   bit 2   the K_r linear map on two accumulators (odd / even limbs)
//   bit 4   (16) PROBE ONLY, wrong results: every round reads the same table lines (no scalar-cache misses)
//   bit 9   (512) full rounds: the S-boxes of two words statement by statement (two independent chains, no extra instruction)
//   bit 5   (32) the lone-wave constant pipeline: K_r's columns three ahead + the next round's cache lines touched an S-box
//           ahead (hades_fast.hpp, fast_round)
// For every variant and batch size: microseconds per launch (median of 9, HIP events around ONE launch) and the digest
// of the output against variant 0 -- same limbs by construction, checked anyway.
//
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I hades252_amd/csrc -o build_tools/lonewave_probe tools/lonewave_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "hades_constants.inc"
#include "hades_fast.hpp"

using namespace hades;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__constant__ FastTables d_fast = {HADES_FAST_ROUND_INIT, HADES_FAST_FINAL_F, HADES_FAST_LIN_INIT, HADES_FAST_FINAL_LIN};

template <int V>
__global__ void __launch_bounds__(256, 3) k_perm_v(const uint8_t *in, uint8_t *out, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = lds + (threadIdx.x / kWave) * lds_wave_bytes(5);
    size_t rec0 = (size_t)blockIdx.x * 256 + (threadIdx.x / kWave) * kWave;
    Fr st[5];
    wave_load_records<5>(in, rec0, n, slab, st);
    Fr res[5];
    fast_perm<5, V>(&d_fast, st, res, 0);
    wave_store_records<5>(out, rec0, n, slab, res);
}

__global__ void k_fill(uint64_t *w, size_t n_u64) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_u64) return;
    uint64_t z = 0x4861646573323532ull + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    if ((i & 3) == 3) z &= 0x3fffffffffffffffull;
    w[i] = z;
}

__global__ void k_xor(const uint64_t *w, size_t n_u64, unsigned long long *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    for (; i < n_u64; i += (size_t)gridDim.x * blockDim.x) v += w[i] * (2 * i + 1);
    atomicAdd(out, v);
}

template <int V>
static void launch(const uint8_t *in, uint8_t *out, size_t n) {
    hipLaunchKernelGGL(k_perm_v<V>, dim3((unsigned)((n + 255) / 256)), dim3(256), 4 * lds_wave_bytes(5), 0, in, out, n);
}
typedef void (*launch_fn)(const uint8_t *, uint8_t *, size_t);

int main(int argc, char **argv) {
    const int variants[] = {0, 2, 512, 514, 0, 2, 514};
    launch_fn fns[] = {launch<0>, launch<2>, launch<512>, launch<514>, launch<0>, launch<2>, launch<514>};
    const size_t sizes[] = {(size_t)1 << 14, (size_t)3 << 13, (size_t)1 << 15, (size_t)3 << 14, (size_t)1 << 16, (size_t)3 << 15,
                            (size_t)1 << 17, (size_t)1 << 18, (size_t)1 << 20, (size_t)1 << 24};
    const size_t nmax = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 24);   // argv[1]: log2 of the largest batch (PMC runs: 16)
    uint8_t *in, *out;
    unsigned long long *dg;
    CHECK(hipMalloc(&in, nmax * 160));
    CHECK(hipMalloc(&out, nmax * 160));
    CHECK(hipMalloc(&dg, 8));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((nmax * 20 + 255) / 256)), dim3(256), 0, 0, (uint64_t *)in, nmax * 20);
    CHECK(hipDeviceSynchronize());
    for (size_t vi = 0; vi < sizeof(variants) / sizeof(variants[0]); vi++) {
        hipFuncAttributes a;
        const void *fp = nullptr;
        switch (variants[vi]) {
            case 0: fp = (const void *)k_perm_v<0>; break;
            case 2: fp = (const void *)k_perm_v<2>; break;
            case 512: fp = (const void *)k_perm_v<512>; break;
            default: fp = (const void *)k_perm_v<514>; break;
        }
        CHECK(hipFuncGetAttributes(&a, fp));
        printf("variant %d: %d VGPRs, %zu B scratch\n", variants[vi], a.numRegs, (size_t)a.localSizeBytes);
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%10s", "states");
    for (int v : variants) printf("   V=%d us", v);
    printf("   (digests equal)\n");
    for (size_t n : sizes) {
        if (n > nmax) continue;
        printf("%10zu", n);
        unsigned long long ref = 0;
        bool same = true;
        for (size_t vi = 0; vi < sizeof(variants) / sizeof(variants[0]); vi++) {
            std::vector<float> ts;
            for (int rep = 0; rep < 11; rep++) {
                CHECK(hipEventRecord(e0, 0));
                fns[vi](in, out, n);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            printf(" %8.1f", ts[ts.size() / 2] * 1e3);
            CHECK(hipMemset(dg, 0, 8));
            hipLaunchKernelGGL(k_xor, dim3(1024), dim3(256), 0, 0, (const uint64_t *)out, n * 20, dg);
            unsigned long long h;
            CHECK(hipMemcpy(&h, dg, 8, hipMemcpyDeviceToHost));
            if (vi == 0) ref = h;
            same = same && (h == ref || (variants[vi] & (16 | 64 | 128 | 256)));   // (variant 16 and up: fixed table lines, wrong results by design)
        }
        printf("   %s\n", same ? "yes" : "NO");
    }
    return 0;
}
