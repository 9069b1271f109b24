// ubench.hip -- VALU issue-rate microbenchmark for the integer/FP64 primitives a big-integer
// Montgomery product can be built from on gfx950.  Replaces the "expected quarter rate"
// guess of SURVEY.md section 8(d) with measured cycles per wave-instruction per SIMD.
//
//   hipcc -O3 --offload-arch=gfx950 -o ubench tools/ubench.hip && ./ubench
//
// Method: every kernel runs ITER iterations of 16 independent copies of one instruction (or a
// dependent chain where stated) in W waves per SIMD on every CU; one wave per block stamps
// s_memtime around the loop.  cycles/instr/SIMD = median(delta) / (ITER * 16 * W).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int ITER = 2048;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// 32-bit destination, two 32-bit sources: D[i] = op(D[i], b)
#define KERNEL_32(NAME, ASM)                                                                   \
    __global__ void NAME(uint32_t *out, unsigned long long *cyc, uint32_t seed) {              \
        uint32_t d[16];                                                                        \
        uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;                 \
        for (int i = 0; i < 16; i++) d[i] = seed + i * 7919u + threadIdx.x;                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                  \
        for (int it = 0; it < ITER; it++) {                                                    \
            _Pragma("unroll") for (int i = 0; i < 16; i++)                                     \
                asm volatile(ASM : "+v"(d[i]) : "v"(b), "v"(c));                               \
        }                                                                                      \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                  \
        uint32_t acc = 0;                                                                      \
        for (int i = 0; i < 16; i++) acc ^= d[i];                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                      \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    }

// 64-bit destination: D[i] (64) = op(..)
#define KERNEL_64(NAME, ASM)                                                                   \
    __global__ void NAME(uint32_t *out, unsigned long long *cyc, uint32_t seed) {              \
        uint64_t d[16];                                                                        \
        uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;                 \
        uint64_t e = ((uint64_t)seed << 32) | threadIdx.x;                                     \
        for (int i = 0; i < 16; i++) d[i] = seed + i * 7919u + threadIdx.x;                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                  \
        for (int it = 0; it < ITER; it++) {                                                    \
            _Pragma("unroll") for (int i = 0; i < 16; i++)                                     \
                asm volatile(ASM : "+v"(d[i]) : "v"(b), "v"(c), "v"(e));                       \
        }                                                                                      \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                  \
        uint64_t acc = 0;                                                                      \
        for (int i = 0; i < 16; i++) acc ^= d[i];                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));            \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    }

// f64 destination
#define KERNEL_F64(NAME, ASM)                                                                  \
    __global__ void NAME(uint32_t *out, unsigned long long *cyc, uint32_t seed) {              \
        double d[16];                                                                          \
        double b = 1.0 + seed * 1e-9 + threadIdx.x * 1e-12, c = 0.5 + seed * 1e-10;            \
        for (int i = 0; i < 16; i++) d[i] = 1.0 + i * 1e-3 + threadIdx.x * 1e-6;               \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                  \
        for (int it = 0; it < ITER; it++) {                                                    \
            _Pragma("unroll") for (int i = 0; i < 16; i++)                                     \
                asm volatile(ASM : "+v"(d[i]) : "v"(b), "v"(c));                               \
        }                                                                                      \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                  \
        double acc = 0;                                                                        \
        for (int i = 0; i < 16; i++) acc += d[i];                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(acc);      \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    }

KERNEL_32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL_32(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL_32(k_xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL_32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL_32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 29")
KERNEL_32(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 29")
KERNEL_32(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL_32(k_addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL_32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL_32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL_32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL_32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL_32(k_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL_32(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL_32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL_32(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL_32(k_mad_u16, "v_mad_u16 %0, %0, %1, %2")
KERNEL_32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL_32(k_mov_b32, "v_mov_b32 %0, %1")

// D = S0*S1 + S2(64): accumulate chain on the destination (dependent only on itself)
KERNEL_64(k_mad_u64_u32_acc, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
// independent of the destination: D = b*c + e
KERNEL_64(k_mad_u64_u32_ind, "v_mad_u64_u32 %0, vcc, %1, %2, %3")
KERNEL_64(k_mad_i64_i32_acc, "v_mad_i64_i32 %0, vcc, %1, %2, %0")
KERNEL_64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %3")
KERNEL_64(k_lshrrev_b64, "v_lshrrev_b64 %0, 29, %0")
KERNEL_64(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
KERNEL_64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %3, %3")
KERNEL_64(k_mov_b64, "v_mov_b64 %0, %3")
KERNEL_64(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %1")

KERNEL_F64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL_F64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_F64(k_add_f64, "v_add_f64 %0, %0, %1")

// dependent chains (latency): one accumulator, 16 serial ops per iteration
__global__ void k_mad_u64_u32_dep(uint32_t *out, unsigned long long *cyc, uint32_t seed) {
    uint64_t d = seed + threadIdx.x;
    uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d ^ (d >> 32));
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void k_fma_f64_dep(uint32_t *out, unsigned long long *cyc, uint32_t seed) {
    double d = 1.0 + threadIdx.x * 1e-6;
    double b = 1.0 + seed * 1e-9, c = 0.5 + seed * 1e-10;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(d);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// mixed: 1 mad_u64 : 2 add (does integer-multiply issue overlap with plain VALU?)
__global__ void k_mix_mad_add(uint32_t *out, unsigned long long *cyc, uint32_t seed) {
    uint64_t d[8];
    uint32_t a[16];
    uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;
    for (int i = 0; i < 8; i++) d[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 16; i++) a[i] = seed * 3 + i + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(b), "v"(c));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[2 * i]) : "v"(b));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[2 * i + 1]) : "v"(c));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint64_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= d[i];
    for (int i = 0; i < 16; i++) acc ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// mixed: 1 fma_f64 : 1 mad_u64 (do the two multipliers run in parallel?)
__global__ void k_mix_mad_dfma(uint32_t *out, unsigned long long *cyc, uint32_t seed) {
    uint64_t d[8];
    double f[8];
    uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;
    double fb = 1.0 + seed * 1e-9, fc = 0.5;
    for (int i = 0; i < 8; i++) d[i] = seed + i + threadIdx.x, f[i] = 1.0 + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(b), "v"(c));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fb), "v"(fc));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint64_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= d[i] ^ (uint64_t)__double_as_longlong(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(acc ^ (acc >> 32));
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

typedef void (*kern_t)(uint32_t *, unsigned long long *, uint32_t);

static void run(const char *name, kern_t k, int instr_per_iter, int waves_per_simd, uint32_t *d_out,
                unsigned long long *d_cyc, int n_cu) {
    int block = 64 * 4 * waves_per_simd;      // one block per CU, waves spread over its 4 SIMDs
    if (block > 1024) block = 1024;
    int blocks_per_cu = (64 * 4 * waves_per_simd) / block;
    int grid = n_cu * blocks_per_cu;
    int n_waves = grid * block / 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d_out, d_cyc, 12345u);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d_out, d_cyc, 777u);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(n_waves);
    CHECK(hipMemcpy(cyc.data(), d_cyc, n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    double med = (double)cyc[n_waves / 2];
    double per = med / ((double)ITER * instr_per_iter * waves_per_simd);
    double total_instr = (double)ITER * instr_per_iter * n_waves;
    printf("%-22s W=%d  cyc/instr/SIMD %7.3f   (memtime ticks; median wave %10.0f)   wall %8.3f ms  => %7.2f G wave-instr/s\n",
           name, waves_per_simd, per, med, ms, total_instr / (ms * 1e-3) / 1e9);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int n_cu = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %d kHz\n", prop.name, n_cu, prop.clockRate);
    uint32_t *d_out;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 1024 * sizeof(uint32_t)));
    CHECK(hipMalloc(&d_cyc, (size_t)n_cu * 8 * 16 * sizeof(unsigned long long)));
#define RUN(K, IPI)                               \
    for (int w : {1, 2, 4}) run(#K, K, IPI, w, d_out, d_cyc, n_cu);
    RUN(k_add_u32, 16)
    RUN(k_mov_b32, 16)
    RUN(k_add3_u32, 16)
    RUN(k_and_b32, 16)
    RUN(k_xor_b32, 16)
    RUN(k_lshl_add_u32, 16)
    RUN(k_alignbit, 16)
    RUN(k_bfe_u32, 16)
    RUN(k_add_co, 16)
    RUN(k_addc_co, 16)
    RUN(k_cndmask, 16)
    RUN(k_mul_lo_u32, 16)
    RUN(k_mul_hi_u32, 16)
    RUN(k_mul_u32_u24, 16)
    RUN(k_mul_hi_u32_u24, 16)
    RUN(k_mad_u32_u24, 16)
    RUN(k_mad_i32_i24, 16)
    RUN(k_dot4_u32_u8, 16)
    RUN(k_pk_mul_lo_u16, 16)
    RUN(k_pk_mad_u16, 16)
    RUN(k_mad_u16, 16)
    RUN(k_fma_f32, 16)
    RUN(k_cvt_f32_u32, 16)
    RUN(k_mad_u64_u32_acc, 16)
    RUN(k_mad_u64_u32_ind, 16)
    RUN(k_mad_i64_i32_acc, 16)
    RUN(k_lshl_add_u64, 16)
    RUN(k_lshrrev_b64, 16)
    RUN(k_lshlrev_b64, 16)
    RUN(k_pk_fma_f32, 16)
    RUN(k_mov_b64, 16)
    RUN(k_cvt_f64_u32, 16)
    RUN(k_fma_f64, 16)
    RUN(k_mul_f64, 16)
    RUN(k_add_f64, 16)
    RUN(k_mad_u64_u32_dep, 16)
    RUN(k_fma_f64_dep, 16)
    RUN(k_mix_mad_add, 24)
    RUN(k_mix_mad_dfma, 16)
    return 0;
}
