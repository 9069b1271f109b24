// hades_fast.cuh -- the scale-tracked Hades252 permutation kernel (the shipped hot path).
//
// Same field elements as the reference's ScalarStrategy::perm (src/strategies.rs:140-157,
// src/strategies/scalar.rs:23-49) -- hence the same bits after the final full reduction --
// with ~10x fewer VALU instructions than the literal round structure:
//
//  1. Unsaturated radix 2^29, 9 limbs per element.  Measured on gfx950 (tools/ubench.hip):
//     v_mad_u64_u32 issues at the same ~3.4 cycles/wave as v_addc_co_u32, so the cost of a
//     big-integer product is its INSTRUCTION COUNT.  29-bit limbs leave 6 bits of headroom in a
//     64-bit column, so a column of 9 products + 9 reduction terms needs no carry handling:
//     every limb product is exactly one v_mad_u64_u32 accumulating in place.
//  2. MDS with small integers.  The reference matrix is M[i][j] = 2^256/(i+j+5) mod p
//     (loader semantics of src/mds_matrix.rs:18-40) = lam * C with C[i][j] = 360360/(i+j+5)
//     < 2^17.  The kernel multiplies by C (9 mads per word instead of a 81+72 mad Montgomery
//     product) and never applies lam: it is tracked as a known scale factor of the state.
//  3. Scale tracking.  Neither lam, nor the 1/Rp of each Montgomery product (Rp = 2^261), nor the
//     2^-29 of the one-limb reduction after each linear layer is ever applied; the running scale
//     s_r is folded into the round constants on the host (hades252_amd/_derive.py).  In partial
//     rounds one extra constant product K_r brings the S-boxed word back to the common scale.
//     One product with FINAL_F at the end returns value * 2^256, which is then fully reduced --
//     the unique in-memory BlsScalar.
//  4. Partial-round ARK on words 0..3 is pushed through the linear layer (D_r seeds the
//     accumulators).
//
// Register budget: state 5 x 9 VGPRs; one product in flight needs <= 36 (columns) + 18.
// Constants are wave-uniform: scalar loads (SMEM) into SGPRs, consumed directly as
// v_mad_u64_u32 operands.
#pragma once
#include "fr32.cuh"
#include "staging.cuh"

namespace hades {

constexpr int kLB = 29;                       // limb bits
constexpr int kNL = 9;                        // limbs
constexpr uint32_t kMask29 = (1u << kLB) - 1;

struct F29 {
    uint32_t l[kNL];
};

__device__ static constexpr uint32_t P29[kNL] = HADES_P29;
__device__ static constexpr uint32_t MDS_SMALL[5][5] = HADES_FAST_MDS_SMALL;

struct FastTables {
    uint32_t full[8][48];     // [round][word*9 + limb], rounds 0..3 then 63..66
    uint32_t part[59][64];    // {A4[9], K[9], D[5][9], pad}
    uint32_t final_f[kNL + 7];
};

// ---- 8 x 32 <-> 9 x 29 ---------------------------------------------------------------------
__device__ __forceinline__ F29 to_f29(const Fr &a) {
    F29 r;
#pragma unroll
    for (int k = 0; k < kNL; k++) {
        int bit = kLB * k, w = bit >> 5, sh = bit & 31;
        uint64_t two = a.l[w];
        if (w + 1 < 8) two |= (uint64_t)a.l[w + 1] << 32;
        r.l[k] = (uint32_t)(two >> sh) & kMask29;
    }
    return r;
}

// limbs must be normalized (< 2^29) and the value < 2^256
__device__ __forceinline__ Fr from_f29(const F29 &a) {
    Fr r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        // word w = bits [32w, 32w+32)
        int k = (32 * w) / kLB, sh = 32 * w - kLB * k;       // starts inside limb k at bit sh
        uint64_t acc = (uint64_t)a.l[k] >> sh;
        int have = kLB - sh;
        if (k + 1 < kNL) acc |= (uint64_t)a.l[k + 1] << have;
        have += kLB;
        if (have < 32 && k + 2 < kNL) acc |= (uint64_t)a.l[k + 2] << have;
        r.l[w] = (uint32_t)acc;
    }
    return r;
}

// ---- Montgomery product, Rp = 2^261 ----------------------------------------------------------
// Inputs: limbs < 2^30 (lazy), values < 2^258.  Output: limbs < 2^29, value < 2^256.
// Column bound: 9 * 2^60 + 9 * 2^58 + carry < 2^64.
__device__ __forceinline__ F29 mont_reduce(uint64_t (&t)[2 * kNL]) {
#pragma unroll
    for (int k = 0; k < kNL; k++) {
        uint32_t m = (0u - (uint32_t)t[k]) & kMask29;        // -t_k * p^-1 mod 2^29, p == 1 mod 2^29
        t[k] += m;                                           // m * p_0, p_0 = 1
#pragma unroll
        for (int j = 1; j < kNL; j++) t[k + j] += (uint64_t)m * P29[j];
        t[k + 1] += t[k] >> kLB;                             // exact: low 29 bits are zero
    }
    F29 r;
#pragma unroll
    for (int k = 0; k < kNL - 1; k++) {
        r.l[k] = (uint32_t)t[kNL + k] & kMask29;
        t[kNL + k + 1] += t[kNL + k] >> kLB;
    }
    r.l[kNL - 1] = (uint32_t)t[2 * kNL - 1];
    return r;
}

__device__ __forceinline__ F29 mont_mul(const F29 &a, const F29 &b) {
    uint64_t t[2 * kNL];
#pragma unroll
    for (int k = 0; k < 2 * kNL; k++) t[k] = 0;
#pragma unroll
    for (int i = 0; i < kNL; i++)
#pragma unroll
        for (int j = 0; j < kNL; j++) t[i + j] += (uint64_t)a.l[i] * b.l[j];
    return mont_reduce(t);
}

__device__ __forceinline__ F29 mont_sqr(const F29 &a) {
    uint64_t t[2 * kNL];
#pragma unroll
    for (int k = 0; k < 2 * kNL; k++) t[k] = 0;
    uint32_t d[kNL];
#pragma unroll
    for (int i = 0; i < kNL; i++) d[i] = a.l[i] << 1;       // < 2^31
#pragma unroll
    for (int i = 0; i < kNL; i++) {
        t[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < kNL; j++) t[i + j] += (uint64_t)a.l[i] * d[j];
    }
    return mont_reduce(t);
}

// v^5 / Rp^4
__device__ __forceinline__ F29 sbox29(const F29 &v) {
    F29 v2 = mont_sqr(v);
    F29 v4 = mont_sqr(v2);
    return mont_mul(v4, v);
}

// One-limb Montgomery step + carry normalisation of 9 accumulator columns (each < 2^59):
// returns (T + m p) / 2^29 with limbs < 2^29.  T < 2^275 => result < 2^256.
__device__ __forceinline__ F29 redc1_normalize(uint64_t (&t)[kNL]) {
    uint32_t m = (0u - (uint32_t)t[0]) & kMask29;
    uint64_t carry = (t[0] + m) >> kLB;
    F29 r;
#pragma unroll
    for (int j = 1; j < kNL; j++) {
        uint64_t v = t[j] + (uint64_t)m * P29[j] + carry;
        r.l[j - 1] = (uint32_t)v & kMask29;
        carry = v >> kLB;
    }
    r.l[kNL - 1] = (uint32_t)carry;
    return r;
}

__device__ __forceinline__ F29 load_f29(const uint32_t *p) {
    F29 r;
#pragma unroll
    for (int k = 0; k < kNL; k++) r.l[k] = p[k];
    return r;
}

__device__ __forceinline__ void add_lazy(F29 &x, const F29 &c) {
#pragma unroll
    for (int k = 0; k < kNL; k++) x.l[k] += c.l[k];
}

// Y = C * X (+ seed), then REDC1 + normalise every word.
template <bool SEED>
__device__ __forceinline__ void small_mds(F29 (&st)[5], const uint32_t *seed /* [5][9] */) {
    F29 out[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        uint64_t t[kNL];
#pragma unroll
        for (int k = 0; k < kNL; k++) {
            t[k] = SEED ? (uint64_t)seed[i * kNL + k] : 0;
#pragma unroll
            for (int j = 0; j < 5; j++) t[k] += (uint64_t)st[j].l[k] * MDS_SMALL[i][j];
        }
        out[i] = redc1_normalize(t);
    }
#pragma unroll
    for (int i = 0; i < 5; i++) st[i] = out[i];
}

template <int W>
__device__ __forceinline__ void full_round_word(const uint32_t *rec, F29 (&st)[5]) {
    add_lazy(st[W], load_f29(rec + W * kNL));
    st[W] = sbox29(st[W]);
}

__device__ __forceinline__ void fast_full_round(const FastTables *T, int idx, F29 (&st)[5]) {
    const uint32_t *rec = T->full[idx];
    full_round_word<0>(rec, st);
    full_round_word<1>(rec, st);
    full_round_word<2>(rec, st);
    full_round_word<3>(rec, st);
    full_round_word<4>(rec, st);
    small_mds<false>(st, nullptr);
}

__device__ __forceinline__ void fast_partial_round(const FastTables *T, int idx, F29 (&st)[5]) {
    const uint32_t *rec = T->part[idx];
    F29 w = st[4];
    add_lazy(w, load_f29(rec));
    w = sbox29(w);
    st[4] = mont_mul(w, load_f29(rec + kNL));
    small_mds<true>(st, rec + 2 * kNL);
}

// in: 5 BlsScalars (Montgomery 2^256 form, fully reduced); out: same format, fully reduced.
template <int NOUT>
__device__ __forceinline__ void fast_perm(const FastTables *T, const Fr (&in)[5], Fr (&out)[NOUT], int out_first) {
    F29 st[5];
#pragma unroll
    for (int w = 0; w < 5; w++) st[w] = to_f29(in[w]);
    // one loop, two bodies: each round body exists once in the instruction stream (the
    // branch is wave-uniform), so the whole kernel stays inside the instruction cache
#pragma unroll 1
    for (int r = 0; r < 67; r++) {
        if (r < 4 || r >= 63)
            fast_full_round(T, r < 4 ? r : r - 59, st);
        else
            fast_partial_round(T, r - 4, st);
    }
    F29 f = load_f29(T->final_f);
    if constexpr (NOUT == 5) {
#pragma unroll
        for (int w = 0; w < 5; w++) out[w] = fr_cond_sub_p(from_f29(mont_mul(st[w], f)));
    } else {
        F29 sel = st[0];
#pragma unroll
        for (int w = 1; w < 5; w++)
            if (out_first == w) sel = st[w];
        out[0] = fr_cond_sub_p(from_f29(mont_mul(sel, f)));
    }
}

}  // namespace hades
