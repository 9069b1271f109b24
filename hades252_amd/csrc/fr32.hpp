// fr32.hpp -- BLS12-381 scalar field Fr on gfx950, saturated 8 x u32 limbs, Montgomery R = 2^256.
//
// This is the memory-format arithmetic: values are exactly the reference's in-memory
// `BlsScalar` (4 x u64 LE limbs of value*R mod p, fully reduced), viewed as 8 x u32.
// It replaces the calls into the external crate dusk-bls12_381 made at
// reference src/strategies/scalar.rs:28 (`+=`), :33 (`square`, `*`), :44 (`*`, `+=`).
//
// p = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
// p == 1 (mod 2^32)  =>  -p^{-1} mod 2^32 = 0xffffffff, so the Montgomery quotient digit is
// simply -t0, and p's two low limbs are {1, 0xffffffff}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hades {

struct Fr {
    uint32_t l[8];
};

__device__ static constexpr uint32_t FR_P[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                                0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};

// Multi-word borrow / carry chains are written with clang's subtract- / add-with-carry builtins: they compile to the chain
// the hardware has (v_sub_co_u32 + 7 v_subb_co_u32).  The portable formulation -- a 64-bit difference per word with the
// borrow taken from bit 32 -- compiles on gfx950 to 64-bit adds with moves and sign extensions in between (16 v_lshl_add_u64
// + 16 v_mov + 7 v_ashrrev for eight words: round 6, measured on the kernels that leave through it hundreds of times per
// state and on the HBM-bound wire kernels, profiles/r6/carry_chain_ab.txt).

// r = a - p if a >= p (a < 2p < 2^256 + p; `top` is the 257th bit)
__device__ __forceinline__ Fr fr_cond_sub_p(const Fr &a, uint32_t top = 0) {
    Fr d;
    unsigned borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        unsigned bo;
        d.l[i] = __builtin_subc(a.l[i], FR_P[i], borrow, &bo);
        borrow = bo;
    }
    // keep d when no borrow happened, or when the borrow is paid by the top bit
    bool use_d = (borrow == 0) || (top != 0);
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = use_d ? d.l[i] : a.l[i];
    return r;
}

__device__ __forceinline__ Fr fr_add(const Fr &a, const Fr &b) {
    Fr s;
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        unsigned co;
        s.l[i] = __builtin_addc(a.l[i], b.l[i], c, &co);
        c = co;
    }
    return fr_cond_sub_p(s, (uint32_t)c);
}

// Montgomery product a*b/R mod p, CIOS, inputs and output fully reduced.
__device__ __forceinline__ Fr fr_mul(const Fr &a, const Fr &b) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t x = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        uint64_t x = (uint64_t)t[8] + c;
        t[8] = (uint32_t)x;
        t[9] = (uint32_t)(x >> 32);
        uint32_t m = 0u - t[0];              // t0 * (-p^-1) mod 2^32
        c = ((uint64_t)m * FR_P[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            uint64_t y = (uint64_t)m * FR_P[j] + t[j] + c;
            t[j - 1] = (uint32_t)y;
            c = y >> 32;
        }
        x = (uint64_t)t[8] + c;
        t[7] = (uint32_t)x;
        t[8] = t[9] + (uint32_t)(x >> 32);
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return fr_cond_sub_p(r, t[8]);
}

__device__ __forceinline__ Fr fr_sqr(const Fr &a) { return fr_mul(a, a); }

// out-of-line instance: keeps the literal kernels' code inside the instruction cache
__device__ __noinline__ Fr fr_mul_call(Fr a, Fr b) { return fr_mul(a, b); }

__device__ __forceinline__ bool fr_is_canonical(const Fr &a) {   // a < p ?
    unsigned borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        unsigned bo;
        (void)__builtin_subc(a.l[i], FR_P[i], borrow, &bo);
        borrow = bo;
    }
    return borrow != 0;
}

}  // namespace hades
