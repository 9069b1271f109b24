// ubench2.hip -- sustained-load VALU rates with the in-kernel clock (s_memtime / s_memrealtime),
// and the cost of the kernel's own building blocks (Montgomery product, S-box, linear layer)
// free of any memory traffic.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I hades252_amd/csrc -o build_tools/ubench2 tools/ubench2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "hades_constants.inc"
#include "hades_fast.cuh"

using namespace hades;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

struct Stamp { unsigned long long cyc, rt; };

template <int KIND>
__global__ void k_sustained(uint32_t *out, Stamp *st, int iters, uint32_t seed) {
    uint64_t d[16];
    uint32_t a32[16];
    uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;
    for (int i = 0; i < 16; i++) { d[i] = seed + i * 7919u + threadIdx.x; a32[i] = seed * 3 + i + threadIdx.x; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(b), "v"(c) : "vcc");
            if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a32[i]) : "v"(b));
            if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a32[i]) : "v"(b), "v"(c));
            if (KIND == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
            if (KIND == 4) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(d[i]));
            if (KIND == 5) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a32[i]));
            if (KIND == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(b), "s"(seed) : "vcc");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= d[i] ^ a32[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
    if ((threadIdx.x & 63) == 0) {
        Stamp s; s.cyc = t1 - t0; s.rt = r1 - r0;
        st[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;
    }
}

// building blocks: KIND 0 = mont_mul chain, 1 = mont_sqr chain, 2 = sbox chain, 3 = small_mds, 4 = const mul
__device__ const int32_t d_k[16] = {0x12345678 & 0x1fffffff, 0x0abcdef1, 0x1fedcba9, 0x13572468, 0x02468ace, 0x1badf00d, 0x0c0ffee0, 0x1eadbeef & 0x1fffffff, 0x123456};
template <int KIND, int MINW>
__global__ void __launch_bounds__(256, MINW) k_blocks(uint32_t *out, Stamp *stamps, int iters, uint32_t seed) {
    F29 st[5];
    for (int w = 0; w < 5; w++)
        for (int k = 0; k < kNL; k++) st[w].l[k] = (int32_t)((seed * (w * 9 + k + 1) * 2654435761u + threadIdx.x * 40503u) & kMask29);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) st[0] = mont_mul(st[0], st[1]);
        if (KIND == 1) st[0] = mont_sqr(st[0]);
        if (KIND == 2) st[0] = sbox29(st[0]);
        if (KIND == 3) small_mds(st);
        if (KIND == 4) st[0] = mont_mul_const(st[0], d_k);
        if (KIND == 5) { st[0] = sbox29(st[0]); st[1] = sbox29(st[1]); st[2] = sbox29(st[2]); st[3] = sbox29(st[3]); st[4] = sbox29(st[4]); }
#pragma unroll
        for (int w = 0; w < 5; w++)
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(st[w].l[k]);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int w = 0; w < 5; w++)
        for (int k = 0; k < kNL; k++) acc ^= st[w].l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        Stamp s; s.cyc = t1 - t0; s.rt = r1 - r0;
        stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s;
    }
}

typedef void (*kern_t)(uint32_t *, Stamp *, int, uint32_t);

static void run(const char *name, kern_t k, double ops_per_iter, int iters, int waves_per_simd, uint32_t *d_out, Stamp *d_st, int n_cu) {
    int block = 256;
    int grid = n_cu * waves_per_simd;          // 256 threads = 1 wave per SIMD per block
    int n_waves = grid * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d_out, d_st, iters / 8 + 1, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d_out, d_st, iters, 777u);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> s(n_waves);
    CHECK(hipMemcpy(s.data(), d_st, n_waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> cyc(n_waves), clk(n_waves);
    for (int i = 0; i < n_waves; i++) { cyc[i] = (double)s[i].cyc; clk[i] = (double)s[i].cyc / (double)s[i].rt * 100.0; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    double med = cyc[n_waves / 2];
    printf("%-26s W=%d  wall %8.2f ms  clock %7.1f MHz  cycles/op/SIMD %9.2f  (ops/iter %.0f)  => %8.2f G op/s chip\n", name, waves_per_simd,
           ms, clk[n_waves / 2], med / ((double)iters * ops_per_iter * waves_per_simd), ops_per_iter,
           (double)iters * ops_per_iter * n_waves / (ms * 1e-3) / 1e9);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int n_cu = prop.multiProcessorCount;
    printf("device CUs %d clockRate %d kHz\n", n_cu, prop.clockRate);
    uint32_t *d_out; Stamp *d_st;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * sizeof(uint32_t)));
    CHECK(hipMalloc(&d_st, (size_t)n_cu * 8 * 4 * sizeof(Stamp)));
    const int IT = 400000;   // 16 instr x 400k = 6.4M wave-instr per wave: ~10-30 ms per kernel
    for (int w : {1, 2, 4, 8}) run("v_mad_u64_u32 (vv)", k_sustained<0>, 16, IT, w, d_out, d_st, n_cu);
    for (int w : {4, 8}) run("v_mad_u64_u32 (v,s)", k_sustained<6>, 16, IT, w, d_out, d_st, n_cu);
    for (int w : {1, 2, 4, 8}) run("v_add_u32", k_sustained<1>, 16, IT, w, d_out, d_st, n_cu);
    for (int w : {4, 8}) run("v_fma_f32", k_sustained<2>, 16, IT, w, d_out, d_st, n_cu);
    for (int w : {4, 8}) run("v_lshl_add_u64", k_sustained<3>, 16, IT, w, d_out, d_st, n_cu);
    for (int w : {4, 8}) run("v_lshrrev_b64", k_sustained<4>, 16, IT, w, d_out, d_st, n_cu);
    for (int w : {4, 8}) run("v_and_b32 (literal)", k_sustained<5>, 16, IT, w, d_out, d_st, n_cu);
    const int IB = 20000;
    run("mont_mul  minw2", k_blocks<0, 2>, 1, IB, 2, d_out, d_st, n_cu);
    run("mont_mul  minw4", k_blocks<0, 4>, 1, IB, 4, d_out, d_st, n_cu);
    run("mont_mul  minw8", k_blocks<0, 8>, 1, IB, 8, d_out, d_st, n_cu);
    run("mont_sqr  minw4", k_blocks<1, 4>, 1, IB, 4, d_out, d_st, n_cu);
    run("mont_sqr  minw8", k_blocks<1, 8>, 1, IB, 8, d_out, d_st, n_cu);
    run("mul_const minw4", k_blocks<4, 4>, 1, IB, 4, d_out, d_st, n_cu);
    run("sbox      minw4", k_blocks<2, 4>, 1, IB, 4, d_out, d_st, n_cu);
    run("sbox      minw8", k_blocks<2, 8>, 1, IB, 8, d_out, d_st, n_cu);
    run("5 sbox    minw4", k_blocks<5, 4>, 5, IB / 4, 4, d_out, d_st, n_cu);
    run("small_mds minw4", k_blocks<3, 4>, 1, IB, 4, d_out, d_st, n_cu);
    run("small_mds minw8", k_blocks<3, 8>, 1, IB, 8, d_out, d_st, n_cu);
    return 0;
}
