// residency.hip -- residency sweep of the shipped permutation kernel.
//
// Same device code as k_perm_fast (hades_fast.hpp + staging.hpp), instantiated with different block sizes,
// launch bounds and LDS footprints, so that the number of co-resident waves per SIMD varies from 1 to 5
// while everything else stays fixed.  For every variant: VGPRs (hipFuncGetAttributes), LDS per block,
// blocks per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor) -> waves per SIMD, and the time for 2^24
// permutations (median of 5).  Output is checked against the first variant (digest of all bytes).
//
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I hades252_amd/csrc -o build_tools/residency tools/residency.hip
#include <hip/hip_runtime.h>
#include <string.h>
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

// HALF = 1: the wave stages its 64 records in two halves of 32 through a half-size slab (5.6 KB per wave)
template <int BLOCK, int MINW, int HALF>
__global__ void __launch_bounds__(BLOCK, MINW) k_perm(uint8_t *states, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    size_t rec0 = (size_t)blockIdx.x * BLOCK + (threadIdx.x / kWave) * kWave;
    Fr st[5];
    if constexpr (HALF == 0) {
        uint8_t *slab = lds + (threadIdx.x / kWave) * lds_wave_bytes(5);
        wave_load_records<5>(states, rec0, n, slab, st);
        Fr out[5];
        fast_perm<5>(&d_fast, st, out, 0);
        wave_store_records<5>(states, rec0, n, slab, out);
    } else {
        constexpr int kRec = lds_rec_bytes(5);
        uint8_t *slab = lds + (threadIdx.x / kWave) * (32 * kRec);
        const uint4 *g = reinterpret_cast<const uint4 *>(states + rec0 * 160);
        const size_t total = n * 10, chunk0 = rec0 * 10;
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int k = 0; k < 5; k++) {
                int c = h * 320 + k * 64 + lane;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (chunk0 + c < total) v = g[c];
                int rec = (c - h * 320) / 10, part = (c - h * 320) - rec * 10;
                *reinterpret_cast<uint4 *>(slab + rec * kRec + part * 16) = v;
            }
            __syncthreads();
            if ((lane >> 5) == h) {
#pragma unroll
                for (int w = 0; w < 5; w++) {
                    const uint4 *p = reinterpret_cast<const uint4 *>(slab + (lane & 31) * kRec + w * 32);
                    uint4 lo = p[0], hi = p[1];
                    st[w].l[0] = lo.x; st[w].l[1] = lo.y; st[w].l[2] = lo.z; st[w].l[3] = lo.w;
                    st[w].l[4] = hi.x; st[w].l[5] = hi.y; st[w].l[6] = hi.z; st[w].l[7] = hi.w;
                }
            }
            __syncthreads();
        }
        Fr out[5];
        fast_perm<5>(&d_fast, st, out, 0);
        uint4 *go = reinterpret_cast<uint4 *>(states + rec0 * 160);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if ((lane >> 5) == h) {
#pragma unroll
                for (int w = 0; w < 5; w++) {
                    uint4 *p = reinterpret_cast<uint4 *>(slab + (lane & 31) * kRec + w * 32);
                    p[0] = make_uint4(out[w].l[0], out[w].l[1], out[w].l[2], out[w].l[3]);
                    p[1] = make_uint4(out[w].l[4], out[w].l[5], out[w].l[6], out[w].l[7]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 5; k++) {
                int c = h * 320 + k * 64 + lane;
                int rec = (c - h * 320) / 10, part = (c - h * 320) - rec * 10;
                uint4 v = *reinterpret_cast<const uint4 *>(slab + rec * kRec + part * 16);
                if (chunk0 + c < total) go[c] = v;
            }
            __syncthreads();
        }
    }
}

__global__ void k_fill(uint64_t *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint64_t z = 0x4861646573323532ull + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        if ((i & 3) == 3) z &= 0x3fffffffffffffffull;
        p[i] = z;
    }
}
// A/B for the north-star wording "constants broadcast from LDS": the same kernel, but every block first copies the
// 17 KB of round records to LDS and the rounds read them from there (wave-uniform addresses: ds_read broadcasts into
// VGPRs) instead of from the scalar cache into SGPRs.
template <int BLOCK, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) k_perm_ldsc(uint8_t *states, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ FastTables Ts;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(&d_fast);
        uint4 *dst = reinterpret_cast<uint4 *>(&Ts);
        for (int i = threadIdx.x; i < (int)(sizeof(FastTables) / 16); i += BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    size_t rec0 = (size_t)blockIdx.x * BLOCK + (threadIdx.x / kWave) * kWave;
    uint8_t *slab = lds + (threadIdx.x / kWave) * lds_wave_bytes(5);
    Fr st[5];
    wave_load_records<5>(states, rec0, n, slab, st);
    Fr out[5];
    fast_perm<5>(&Ts, st, out, 0);
    wave_store_records<5>(states, rec0, n, slab, out);
}

__global__ void k_xor(const uint64_t *p, size_t n, unsigned long long *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t v = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) v += p[i] * (2 * i + 1);
    atomicAdd(out, (unsigned long long)v);
}

typedef void (*kern_t)(uint8_t *, size_t);
static uint64_t g_ref = 0;

static void sweep(const char *name, kern_t k, int block, size_t lds_need, size_t pad, uint8_t *d, size_t n, unsigned long long *d_sum) {
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, (const void *)k));
    size_t lds = lds_need + pad;
    int blocks_per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, (const void *)k, block, lds));
    double waves_per_simd = blocks_per_cu * (block / 64) / 4.0;
    std::vector<float> t;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned grid = (unsigned)((n + block - 1) / block);
    uint64_t sum = 0;
    for (int rep = 0; rep < 8; rep++) {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n * 20 + 255) / 256)), dim3(256), 0, 0, (uint64_t *)d, n * 20);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, d, n);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) t.push_back(ms);
        if (rep == 0) {
            CHECK(hipMemset(d_sum, 0, 8));
            hipLaunchKernelGGL(k_xor, dim3(4096), dim3(256), 0, 0, (const uint64_t *)d, n * 20, d_sum);
            CHECK(hipMemcpy(&sum, d_sum, 8, hipMemcpyDeviceToHost));
            if (g_ref == 0) g_ref = sum;
        }
    }
    std::sort(t.begin(), t.end());
    printf("%-34s block %3d  VGPRs %3d  scratch %3zu  LDS/block %6zu B  blocks/CU %2d  waves/SIMD %4.2f  median %8.3f ms  %7.2f M perm/s  %s\n",
           name, block, fa.numRegs, (size_t)fa.localSizeBytes, lds, blocks_per_cu, waves_per_simd, t[t.size() / 2],
           n / (t[t.size() / 2] * 1e-3) / 1e6, sum == g_ref ? "output ok" : "OUTPUT DIFFERS");
}

int main(int argc, char **argv) {
    if (argc > 1 && strcmp(argv[1], "sizes") == 0) {
        // batch sizes around multiples of what is resident at once: 3 blocks per CU hold 3 072 waves = 196 608 states, 4 hold
        // 4 096 waves = 262 144; a batch slightly above a multiple runs its last blocks alone (round 4)
        uint8_t *d;
        unsigned long long *d_sum;
        CHECK(hipMalloc(&d, ((size_t)1 << 26) * 160));
        CHECK(hipMalloc(&d_sum, 8));
        const size_t full = lds_wave_bytes(5), half = 32 * lds_rec_bytes(5);
        for (size_t n : {(size_t)196608, (size_t)1 << 18, (size_t)3 << 17, (size_t)1 << 19, (size_t)786432, (size_t)1 << 20,
                         (size_t)1 << 22, (size_t)1 << 24, (size_t)1 << 26}) {
            printf("n = %zu states (%zu waves)\n", n, n / 64);
            g_ref = 0;
            sweep("  full slab, 3 blocks/CU (shipped)", k_perm<256, 4, 0>, 256, 4 * full, 0, d, n, d_sum);
            sweep("  half slab, 4 blocks/CU", k_perm<256, 4, 1>, 256, 4 * half, 0, d, n, d_sum);
        }
        return 0;
    }
    const size_t n = (size_t)1 << 24;
    uint8_t *d;
    unsigned long long *d_sum;
    CHECK(hipMalloc(&d, n * 160));
    CHECK(hipMalloc(&d_sum, 8));
    const size_t full = lds_wave_bytes(5), half = 32 * lds_rec_bytes(5);
    printf("k_perm_fast residency sweep, 2^24 permutations (slab per wave: full %zu B, half %zu B)\n", full, half);
    // shipped configuration and fewer resident blocks (LDS pad)
    sweep("256thr lb4 full-slab (shipped)", k_perm<256, 4, 0>, 256, 4 * full, 0, d, n, d_sum);
    sweep("256thr lb4 full-slab pad->2/CU", k_perm<256, 4, 0>, 256, 4 * full, 70 * 1024 - 4 * full, d, n, d_sum);
    sweep("256thr lb4 full-slab pad->1/CU", k_perm<256, 4, 0>, 256, 4 * full, 100 * 1024 - 4 * full, d, n, d_sum);
    sweep("128thr lb5 full-slab", k_perm<128, 5, 0>, 128, 2 * full, 0, d, n, d_sum);
    sweep("64thr lb5 full-slab", k_perm<64, 5, 0>, 64, 1 * full, 0, d, n, d_sum);
    // half slab: LDS no longer limits; launch bounds set the VGPR budget
    sweep("256thr lb4 half-slab", k_perm<256, 4, 1>, 256, 4 * half, 0, d, n, d_sum);
    sweep("256thr lb5 half-slab", k_perm<256, 5, 1>, 256, 4 * half, 0, d, n, d_sum);
    sweep("256thr lb6 half-slab", k_perm<256, 6, 1>, 256, 4 * half, 0, d, n, d_sum);
    sweep("256thr lb4 half-slab pad->4/CU", k_perm<256, 4, 1>, 256, 4 * half, 40 * 1024 - 4 * half, d, n, d_sum);
    sweep("256thr lb4 half-slab pad->3/CU", k_perm<256, 4, 1>, 256, 4 * half, 50 * 1024 - 4 * half, d, n, d_sum);
    sweep("128thr lb5 half-slab", k_perm<128, 5, 1>, 128, 2 * half, 0, d, n, d_sum);
    // round records from LDS instead of SGPRs (median of 7); its 17 KB leave room for 2 blocks per CU, so the SGPR kernel
    // at 2 blocks per CU ("pad->2/CU" above, repeated here) is the like-for-like partner
    printf("-- constants: scalar loads into SGPRs (shipped) vs a per-block LDS copy read by ds_read broadcasts\n");
    sweep("SGPR constants, 3 blocks/CU (shipped)", k_perm<256, 4, 0>, 256, 4 * full, 0, d, n, d_sum);
    sweep("SGPR constants, padded to 2 blocks/CU", k_perm<256, 4, 0>, 256, 4 * full, 70 * 1024 - 4 * full, d, n, d_sum);
    sweep("LDS constants (2 blocks/CU)", k_perm_ldsc<256, 4>, 256, 4 * full, 0, d, n, d_sum);
    sweep("LDS constants, launch bounds 2", k_perm_ldsc<256, 2>, 256, 4 * full, 0, d, n, d_sum);
    return 0;
}
