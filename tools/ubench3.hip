// ubench3.hip -- settles the VALU issue ceiling the shipped kernel is priced against (round 2).
//
// What was wrong with ubench2: blocks were assumed to land W per CU, but nothing forced it,
// so per-wave stamps (median wave) and wall time disagreed by up to 60 %.  Here:
//   * every block is 256 threads (one wave per SIMD) and carries a dynamic-LDS pad of floor(160 KiB / W), so a CU
//     can hold AT MOST W blocks; the grid is exactly n_cu * W blocks, so the steady state is W waves per SIMD;
//   * every wave records HW_ID / XCC_ID (which XCD, SE, CU, SIMD it ran on) and its start / end timestamps
//     (s_memrealtime, constant 100 MHz) -- the host reports the residency histogram (waves per SIMD), the spread
//     of start times (did all blocks start together?) and min / median / max wave duration;
//   * rates are reported from WALL time (hipEvents) and from the waves' own durations; the two must agree.
// Also: lone-wave latency of dependent multiply-add chains with 1..8 independent accumulators (what a
// latency-bound launch sees), and the kernel's building blocks.
//
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I hades252_amd/csrc -o build_tools/ubench3 tools/ubench3.hip
//   ./build_tools/ubench3            (all sections)      ./build_tools/ubench3 rates | lone | blocks
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <vector>
#include "hades_constants.inc"
#include "hades_fast.hpp"

using namespace hades;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

struct Stamp { unsigned long long t0, t1, c0, c1; uint32_t hw_id, xcc_id; };

__device__ __forceinline__ void stamp_begin(Stamp &s) {
    s.c0 = __builtin_amdgcn_s_memtime();
    s.t0 = __builtin_amdgcn_s_memrealtime();
}
__device__ __forceinline__ void stamp_end(Stamp &s, Stamp *out, int wave_index) {
    s.c1 = __builtin_amdgcn_s_memtime();
    s.t1 = __builtin_amdgcn_s_memrealtime();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    s.hw_id = hw;
    s.xcc_id = xcc;
    if ((threadIdx.x & 63) == 0) out[wave_index] = s;
}

// ---- 1. sustained issue rates under forced residency ---------------------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(256) k_rate(uint32_t *out, Stamp *st, int iters, uint32_t seed) {
    extern __shared__ uint8_t pad[];
    uint64_t d[16];
    uint32_t a32[16];
    uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;
    for (int i = 0; i < 16; i++) { d[i] = seed + i * 7919u + threadIdx.x; a32[i] = seed * 3 + i + threadIdx.x; }
    if (seed == 0xffffffffu) pad[threadIdx.x] = 1;          // keep the pad alive
    Stamp s;
    stamp_begin(s);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(b), "v"(c) : "vcc");
            if (KIND == 1) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(b), "s"(seed) : "vcc");
            if (KIND == 2) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
            if (KIND == 3) asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(d[i]));
            if (KIND == 4) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a32[i]));
            if (KIND == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a32[i]) : "v"(b));
            if (KIND == 6) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
        }
    }
    stamp_end(s, st, blockIdx.x * 4 + threadIdx.x / 64);
    uint64_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= d[i] ^ a32[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
}

// ---- 2. lone-wave latency: dependent multiply-add chains with ILP independent accumulators ------------------
template <int ILP, int KIND>
__global__ void __launch_bounds__(64) k_lone(uint32_t *out, Stamp *st, int iters, uint32_t seed) {
    uint64_t d[8];
    uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;
    for (int i = 0; i < 8; i++) d[i] = seed + i * 7919u + threadIdx.x;
    Stamp s;
    stamp_begin(s);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16 / ILP; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (KIND == 0) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(b), "v"(c) : "vcc");
                if (KIND == 1) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(*(uint32_t *)&d[i]));
                if (KIND == 2) asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(d[i]));
            }
        }
    }
    stamp_end(s, st, blockIdx.x);
    uint64_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
}

// ---- 3. the kernel's building blocks (no memory traffic) -----------------------------------------------------
__device__ const int32_t d_k[16] = {0x12345678 & 0x1fffffff, 0x0abcdef1, 0x1fedcba9, 0x13572468, 0x02468ace,
                                    0x1badf00d, 0x0c0ffee0, 0x1eadbeef & 0x1fffffff, 0x123456};
template <int KIND, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_blocks(uint32_t *out, Stamp *stamps, int iters, uint32_t seed) {
    extern __shared__ uint8_t pad[];
    if (seed == 0xffffffffu) pad[threadIdx.x] = 1;
    F29 st[5];
    for (int w = 0; w < 5; w++)
        for (int k = 0; k < kNL; k++)
            st[w].l[k] = (int32_t)((seed * (w * 9 + k + 1) * 2654435761u + threadIdx.x * 40503u) & kMask29);
    Stamp s;
    stamp_begin(s);
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) st[0] = mont_mul(st[0], st[1]);
        if (KIND == 1) st[0] = mont_sqr(st[0]);
        if (KIND == 2) st[0] = sbox29(st[0]);
        if (KIND == 3) small_mds(st);
        if (KIND == 4) st[0] = mont_mul_const(st[0], d_k);
        if (KIND == 5) { st[4] = sbox29(st[4]); st[4] = mont_mul_const(st[4], d_k); small_mds(st); }   // a partial round
#pragma unroll
        for (int w = 0; w < 5; w++)
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(st[w].l[k]);
    }
    stamp_end(s, stamps, blockIdx.x * (BLOCK / 64) + threadIdx.x / 64);
    uint32_t acc = 0;
    for (int w = 0; w < 5; w++)
        for (int k = 0; k < kNL; k++) acc ^= st[w].l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---- 4. where do the five waves of a 320-thread block land? ----------------------------------------------------
__global__ void __launch_bounds__(320) k_where(Stamp *st) {
    Stamp s;
    stamp_begin(s);
    stamp_end(s, st, blockIdx.x * 5 + threadIdx.x / 64);
}

typedef void (*kern_t)(uint32_t *, Stamp *, int, uint32_t);

struct Result { double wall_ms, rate_wall_g, rate_wave_g, cyc_per_op_wall, cyc_per_op_wave; };

static int g_ncu = 0;
static size_t g_max_block_lds = 64 * 1024;
static uint32_t *d_out;
static Stamp *d_st;

// Launch `grid` blocks of `block` threads with `lds` bytes of pad; ops = wave-instructions per wave.
static void run(const char *name, kern_t k, int grid, int block, size_t lds, int iters, double ops_per_iter, int W) {
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_max_block_lds);
    (void)hipGetLastError();
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, d_out, d_st, iters / 16 + 1, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, d_out, d_st, iters, 777u);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int wpb = block / 64, n_waves = grid * wpb;
    std::vector<Stamp> s(n_waves);
    CHECK(hipMemcpy(s.data(), d_st, n_waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> dur(n_waves), clk(n_waves);
    unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
    std::map<uint32_t, int> per_simd;
    for (int i = 0; i < n_waves; i++) {
        dur[i] = (double)(s[i].t1 - s[i].t0) * 10.0;                       // ns (100 MHz)
        clk[i] = (double)(s[i].c1 - s[i].c0) / (double)(s[i].t1 - s[i].t0) * 100.0;   // s_memtime ticks per us
        tmin = std::min(tmin, s[i].t0); tmax = std::max(tmax, s[i].t1); smax = std::max(smax, s[i].t0);
        // gfx9 HW_ID: SIMD_ID [5:4], CU_ID [11:8], SH_ID [12], SE_ID [15:13]; XCC_ID [3:0]
        uint32_t key = ((s[i].xcc_id & 0xf) << 16) | (s[i].hw_id & 0xff30);
        per_simd[key]++;
    }
    std::sort(dur.begin(), dur.end()); std::sort(clk.begin(), clk.end());
    std::map<int, int> hist;
    for (auto &kv : per_simd) hist[kv.second]++;
    const double ops = (double)iters * ops_per_iter;                       // wave-instructions per wave
    const double n_simd = g_ncu * 4.0;
    const double wall_rate = ops * n_waves / (ms * 1e-3) / 1e9;            // G wave-instr/s chip-wide
    const double span_ns = (double)(tmax - tmin) * 10.0;
    const double wave_rate = ops * n_waves / (span_ns * 1e-9) / 1e9;       // from the waves' own clocks
    const double med = dur[n_waves / 2];
    // per-SIMD issue interval in ns, using the median wave duration and W waves sharing a SIMD
    printf("%-24s W=%d grid=%5d  wall %8.3f ms  span %8.3f ms  wave min/med/max %8.3f/%8.3f/%8.3f ms  start-spread %7.3f ms\n"
           "    rate(wall) %7.2f G wi/s  rate(span) %7.2f G wi/s  ns/wi/SIMD: wall %6.3f  med-wave %6.3f  (@2.4GHz: %5.2f / %5.2f cyc)  memtime %6.1f ticks/us\n"
           "    SIMDs used %4zu of %4.0f; waves-per-SIMD histogram:",
           name, W, grid, ms, span_ns * 1e-6, dur[0] * 1e-6, med * 1e-6, dur[n_waves - 1] * 1e-6, (double)(smax - tmin) * 1e-5,
           wall_rate, wave_rate, ms * 1e6 / (ops * n_waves / n_simd), med / (ops * W), ms * 1e6 / (ops * n_waves / n_simd) * 2.4,
           med / (ops * W) * 2.4, clk[n_waves / 2], per_simd.size(), n_simd);
    for (auto &kv : hist) printf("  %d waves x %d SIMDs", kv.first, kv.second);
    printf("\n");
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

// LDS pad that lets at most W blocks share a CU's 160 KiB (capped by the per-block limit of the device: for W
// below 160 KiB / limit the pad cannot force the residency -- the histogram printed with each line tells)
static size_t pad_for(int W) {
    if (W >= 8) return 0;
    size_t want = (size_t)(160 * 1024 / W) - 1024;
    return want > g_max_block_lds ? g_max_block_lds : want;
}

int main(int argc, char **argv) {
    const char *what = argc > 1 ? argv[1] : "all";
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_ncu = prop.multiProcessorCount;
    g_max_block_lds = prop.sharedMemPerBlockOptin ? prop.sharedMemPerBlockOptin : prop.sharedMemPerBlock;
    printf("max LDS per block %zu\n", g_max_block_lds);
    printf("device %s  CUs %d  clockRate %d kHz  LDS/CU %zu\n", prop.name, g_ncu, prop.clockRate, prop.maxSharedMemoryPerMultiProcessor);
    CHECK(hipMalloc(&d_out, (size_t)g_ncu * 16 * 256 * sizeof(uint32_t)));
    CHECK(hipMalloc(&d_st, (size_t)g_ncu * 16 * 4 * sizeof(Stamp)));
    const bool all = !strcmp(what, "all");
    if (all || !strcmp(what, "rates")) {
        printf("== 1. sustained issue rates, residency forced by an LDS pad (W blocks of 4 waves per CU) ==\n");
        const int IT = 200000;   // 16 x 200k = 3.2 M wave-instructions per wave
        for (int w : {1, 2, 3, 4, 6, 8}) run("v_mad_u64_u32 v,v", k_rate<0>, g_ncu * w, 256, pad_for(w), IT, 16, w);
        for (int w : {2, 3, 4, 8}) run("v_mad_i64_i32 v,s", k_rate<1>, g_ncu * w, 256, pad_for(w), IT, 16, w);
        for (int w : {3, 4, 8}) run("v_lshl_add_u64", k_rate<2>, g_ncu * w, 256, pad_for(w), IT, 16, w);
        for (int w : {3, 4, 8}) run("v_ashrrev_i64", k_rate<3>, g_ncu * w, 256, pad_for(w), IT, 16, w);
        for (int w : {1, 2, 3, 4, 8}) run("v_and_b32 literal", k_rate<4>, g_ncu * w, 256, pad_for(w), IT, 16, w);
        for (int w : {1, 2, 4, 8}) run("v_add_u32", k_rate<5>, g_ncu * w, 256, pad_for(w), IT, 16, w);
        for (int w : {1, 2, 4, 8}) run("v_fma_f64", k_rate<6>, g_ncu * w, 256, pad_for(w), IT, 16, w);
    }
    if (all || !strcmp(what, "lone")) {
        printf("== 2. lone wave (1 wave on the whole chip): dependent chains with ILP accumulators ==\n");
        const int IT = 100000;
        run("mad_i64_i32 ILP1", k_lone<1, 0>, 1, 64, 0, IT, 16, 1);
        run("mad_i64_i32 ILP2", k_lone<2, 0>, 1, 64, 0, IT, 16, 1);
        run("mad_i64_i32 ILP4", k_lone<4, 0>, 1, 64, 0, IT, 16, 1);
        run("mad_i64_i32 ILP8", k_lone<8, 0>, 1, 64, 0, IT, 16, 1);
        run("and_b32     ILP1", k_lone<1, 1>, 1, 64, 0, IT, 16, 1);
        run("and_b32     ILP4", k_lone<4, 1>, 1, 64, 0, IT, 16, 1);
        run("ashr_i64    ILP1", k_lone<1, 2>, 1, 64, 0, IT, 16, 1);
        run("ashr_i64    ILP4", k_lone<4, 2>, 1, 64, 0, IT, 16, 1);
        const int IB = 20000;
        run("lone mont_mul", k_blocks<0, 64>, 1, 64, 0, IB, 1, 1);
        run("lone mont_sqr", k_blocks<1, 64>, 1, 64, 0, IB, 1, 1);
        run("lone sbox", k_blocks<2, 64>, 1, 64, 0, IB, 1, 1);
        run("lone small_mds", k_blocks<3, 64>, 1, 64, 0, IB, 1, 1);
        run("lone partial round", k_blocks<5, 64>, 1, 64, 0, IB, 1, 1);
    }
    if (all || !strcmp(what, "coop")) {
        printf("== 4. SIMD of each wave of a 320-thread block (HW_ID bits 5:4), 8 blocks ==\n");
        hipLaunchKernelGGL(k_where, dim3(8), dim3(320), 0, 0, d_st);
        CHECK(hipDeviceSynchronize());
        std::vector<Stamp> s(40);
        CHECK(hipMemcpy(s.data(), d_st, 40 * sizeof(Stamp), hipMemcpyDeviceToHost));
        for (int b = 0; b < 8; b++) {
            printf("block %d: xcc %u cu %2u  wave->simd:", b, s[b * 5].xcc_id & 0xf, (s[b * 5].hw_id >> 8) & 0xf);
            for (int w = 0; w < 5; w++) printf(" %u", (s[b * 5 + w].hw_id >> 4) & 3);
            printf("\n");
        }
    }
    if (all || !strcmp(what, "blocks")) {
        printf("== 3. building blocks under forced residency (ops = block executions per wave) ==\n");
        const int IB = 20000;
        for (int w : {2, 3, 4, 5}) run("mont_mul", k_blocks<0, 256>, g_ncu * w, 256, pad_for(w), IB, 1, w);
        for (int w : {3, 4}) run("mont_sqr", k_blocks<1, 256>, g_ncu * w, 256, pad_for(w), IB, 1, w);
        for (int w : {3, 4}) run("mul_const", k_blocks<4, 256>, g_ncu * w, 256, pad_for(w), IB, 1, w);
        for (int w : {3, 4}) run("sbox", k_blocks<2, 256>, g_ncu * w, 256, pad_for(w), IB, 1, w);
        for (int w : {3, 4}) run("small_mds", k_blocks<3, 256>, g_ncu * w, 256, pad_for(w), IB, 1, w);
        for (int w : {2, 3, 4}) run("partial round", k_blocks<5, 256>, g_ncu * w, 256, pad_for(w), IB / 2, 1, w);
    }
    return 0;
}
