// dfma_proto.hip -- the FP64-FMA big-integer product, counted (VERDICT r1 next #6 i).
//
// Idea (Emmart, Zheng, Weems: "Faster modular exponentiation using double precision floating point
// arithmetic on the GPU"): with 52-bit limbs a limb product a*b < 2^104 is split exactly by two FMAs in
// round-toward-zero mode,
//     hi = fma(a, b, 2^104)                 = 2^104 + floor(ab / 2^52) * 2^52     (mantissa = high half)
//     lo = fma(a, b, (2^104 + 2^52) - hi)   = 2^52 + (ab mod 2^52)                (mantissa = low half)
// and the halves are accumulated as INTEGERS (the bit patterns add; the exponent constants are removed once
// per column).  A 256-bit operand is 5 limbs, so a full product is 25 limb products instead of 81.
//
// What it costs on gfx950: per limb product 2 v_fma_f64 + 1 v_add_f64 + 2 v_lshl_add_u64 = 5 VALU
// instructions, every one of them in the same issue class as v_mad_i64_i32 (tools/ubench3.hip: v_fma_f64
// 559 G wave-instr/s, v_mad_i64_i32 547 G, v_lshl_add_u64 516 G at 8 waves/SIMD).  So the PRODUCT part alone is
// 25 x 5 = 125 instructions + column fix-ups against 81 multiply-adds in mont_fips; the Montgomery reduction
// (5 digits x 4 limb products + 5 quotient digits, each again a split product) adds >= 5 x (4 x 5 + 4) = 120
// against 72 + 34.  ~250 vs 187 instructions per Montgomery product: 1.3x MORE issue cycles, not fewer --
// because CDNA4's 32x32+64 integer multiply-add already runs at the FP64 FMA rate (on the NVIDIA parts the
// technique was designed for, DFMA was 4-16x faster than IMAD).  Not shipped.
//
// This file is the prototype of the product part: it is checked against unsigned __int128 arithmetic on the
// host, and its instruction count is read off the ISA:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S -o /tmp/dfma.s tools/dfma_proto.hip
//   awk '/^_Z10k_dfma_mulPKmS0_Pm:/,/s_endpgm/' /tmp/dfma.s | grep -cE "^\s+v_(fma_f64|add_f64|lshl_add_u64)"
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o build_tools/dfma_proto tools/dfma_proto.hip && ./build_tools/dfma_proto
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int NL = 5;                  // 5 x 52 = 260 bits

__device__ __forceinline__ double fma_asm(double a, double b, double c) {
    double r;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ double sub_asm(double a, double b) {
    double r;
    asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// a, b: 5 limbs of 52 bits as doubles (exact integers).  cols[0..9]: the 10 columns of the 520-bit product,
// column k = sum of the low halves of products i + j = k and the high halves of products i + j = k - 1.
__device__ __forceinline__ void dfma_mul(const double (&a)[NL], const double (&b)[NL], uint64_t (&cols)[2 * NL]) {
    const double c1 = 0x1p104, c2 = 0x1p104 + 0x1p52;
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) cols[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) {
            double hi = fma_asm(a[i], b[j], c1);
            double lo = fma_asm(a[i], b[j], sub_asm(c2, hi));
            cols[i + j + 1] += (uint64_t)__double_as_longlong(hi);      // v_lshl_add_u64
            cols[i + j] += (uint64_t)__double_as_longlong(lo);
        }
    // remove the exponent patterns: column k received n_lo(k) copies of bits(2^52) and n_hi(k) of bits(2^104)
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) {
        int n_lo = 0, n_hi = 0;
        for (int i = 0; i < NL; i++)
            for (int j = 0; j < NL; j++) {
                n_lo += (i + j == k);
                n_hi += (i + j + 1 == k);
            }
        cols[k] -= (uint64_t)n_lo * 0x4330000000000000ull + (uint64_t)n_hi * 0x4670000000000000ull;
    }
}

// in: 4 x u64 per operand (256-bit), out: 10 columns (each < 2^56: 52-bit halves, up to 9 per column)
__global__ void k_dfma_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) {
    // FP64 rounding mode = toward zero (MODE[3:2] = 3)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x[4], y[4];
    for (int k = 0; k < 4; k++) { x[k] = a[4 * t + k]; y[k] = b[4 * t + k]; }
    double fa[NL], fb[NL];
    const uint64_t M = (1ull << 52) - 1;
    for (int k = 0; k < NL; k++) {
        int bit = 52 * k, w = bit >> 6, sh = bit & 63;
        uint64_t va = x[w] >> sh, vb = y[w] >> sh;
        if (sh > 12 && w + 1 < 4) { va |= x[w + 1] << (64 - sh); vb |= y[w + 1] << (64 - sh); }
        fa[k] = (double)(va & M);
        fb[k] = (double)(vb & M);
    }
    uint64_t cols[2 * NL];
    dfma_mul(fa, fb, cols);
    for (int k = 0; k < 2 * NL; k++) out[2 * NL * t + k] = cols[k];
}

int main() {
    const int n = 4096;
    uint64_t *ha = (uint64_t *)malloc(n * 32), *hb = (uint64_t *)malloc(n * 32), *ho = (uint64_t *)malloc(n * 80);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 4 * n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; ha[i] = s;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; hb[i] = s;
    }
    for (int i = 0; i < 8; i++) ha[i] = hb[i] = ~0ull;        // all-ones operands
    uint64_t *da, *db, *dout;
    CHECK(hipMalloc(&da, n * 32)); CHECK(hipMalloc(&db, n * 32)); CHECK(hipMalloc(&dout, n * 80));
    CHECK(hipMemcpy(da, ha, n * 32, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, hb, n * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dfma_mul, dim3(n / 64), dim3(64), 0, 0, da, db, dout);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ho, dout, n * 80, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < n; t++) {
        // reference: 512-bit product by schoolbook on 64-bit words, then compare with sum cols[k] 2^(52k)
        unsigned __int128 acc[9] = {0};
        uint64_t prod[8] = {0};
        for (int i = 0; i < 4; i++) {
            unsigned __int128 carry = 0;
            for (int j = 0; j < 4; j++) {
                unsigned __int128 v = (unsigned __int128)ha[4 * t + i] * hb[4 * t + j] + prod[i + j] + carry;
                prod[i + j] = (uint64_t)v;
                carry = v >> 64;
            }
            prod[i + 4] = (uint64_t)carry;
        }
        (void)acc;
        // recombine the columns into 8 words
        uint64_t got[9] = {0};
        unsigned __int128 run = 0;
        // big shift-add: add cols[k] << 52k
        uint64_t words[10] = {0};
        for (int k = 0; k < 10; k++) {
            uint64_t c = ho[10 * t + k];
            int bit = 52 * k, w = bit >> 6, sh = bit & 63;
            unsigned __int128 v = (unsigned __int128)c << sh;
            unsigned __int128 cy = 0;
            for (int q = w; q < 10 && (v || cy); q++) {
                unsigned __int128 sum = (unsigned __int128)words[q] + (uint64_t)v + cy;
                words[q] = (uint64_t)sum;
                cy = sum >> 64;
                v >>= 64;
            }
        }
        (void)got; (void)run;
        for (int q = 0; q < 8; q++) bad += words[q] != prod[q];
        bad += words[8] != 0 || words[9] != 0;
    }
    printf("dfma 5x52-bit product vs 64-bit schoolbook on %d random operand pairs: %s\n", n, bad ? "MISMATCH" : "all equal");
    return bad != 0;
}
