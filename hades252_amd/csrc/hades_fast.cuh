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
//  4. Every product and every linear-layer row runs on ONE 64-bit accumulator (finely integrated
//     product scanning), so a product in flight needs its two operands, nine quotient digits and
//     two accumulator registers instead of an 18-column array: the kernel fits 4+ waves per SIMD,
//     which the VALU needs to reach its ~3.4-cycle issue rate for v_mad_u64_u32.
//
// Register budget: state 5 x 9 VGPRs + ~30 for the product in flight.
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
    uint32_t round[67][64];   // per round {A[5][9], K[9], pad}: scaled ARK constants, rescale factor
    uint32_t final_f[kNL + 7];
};

// One limb product accumulated in place; hipcc selects a single v_mad_u64_u32 for this shape
// as long as both factors are provably 32-bit (see limb_fence).
// The trailing input-only asm gives every partial sum a second use, which stops LLVM's
// reassociation from rebuilding the column as (p1 + p2 + ...) + carry -- that form needs a fresh
// chain from zero and an extra 64-bit add per column.  It emits no instruction.
__device__ __forceinline__ void pin(const uint64_t &acc) { asm volatile("" ::"v"((uint32_t)acc)); }
__device__ __forceinline__ void mac(uint64_t &acc, uint32_t a, uint32_t b) {
    acc += (uint64_t)a * b;
    pin(acc);
}
__device__ __forceinline__ void mac_s(uint64_t &acc, uint32_t a, uint32_t b_uniform) {
    acc += (uint64_t)a * b_uniform;
    pin(acc);
}

// Zero-instruction fence on one limb: makes the value an opaque 32-bit VGPR.  Without it hipcc
// carries limbs across the round loop's back-edge as 64-bit values (zext(trunc(acc) & mask) is
// folded to a 64-bit AND, the PHI loses the known-zero high half) and every limb product of the
// next round turns into a 64 x 32 multiply: two mads and two moves.
__device__ __forceinline__ void limb_fence(uint32_t &x) { asm volatile("" : "+v"(x)); }

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
// Finely integrated product scanning: ONE 64-bit accumulator walks the 18 columns; column k
// receives its limb products, the reduction terms m_i * p_{k-i} of the quotient digits already
// known, then (k < 9) yields the next digit m_k = -acc mod 2^29 (p == 1 mod 2^29, so
// -p^-1 == -1) or (k >= 9) a result limb.  Every limb product is one v_mad_u64_u32 on the
// accumulator; a column costs two extra VALU ops (digit / limb, shift).  Live registers: the two
// operands, nine digits and the accumulator -- no column array.
// Inputs: limbs < 2^30 (lazy), values < 2^258.  Output: limbs < 2^29, value < 2^256.
// Column bound: 9 * 2^60 + 8 * 2^58 + carry < 2^64.
template <bool SQR>
__device__ __forceinline__ F29 mont_fips(const F29 &a, const F29 &b) {
    uint32_t m[kNL];
    uint32_t d[kNL];                       // 2 * a (squaring only)
    if constexpr (SQR) {
#pragma unroll
        for (int i = 0; i < kNL; i++) d[i] = a.l[i] << 1;
    }
    F29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * kNL - 1; k++) {
        const int lo = k < kNL ? 0 : k - kNL + 1, hi = k < kNL ? k : kNL - 1;
        if constexpr (SQR) {
#pragma unroll
            for (int i = lo; i <= hi; i++) {
                int j = k - i;
                if (i < j) mac(acc, a.l[i], d[j]);
                if (i == j) mac(acc, a.l[i], a.l[i]);
            }
        } else {
#pragma unroll
            for (int i = lo; i <= hi; i++) mac(acc, a.l[i], b.l[k - i]);
        }
#pragma unroll
        for (int i = lo; i <= hi; i++)
            if (k - i >= 1) mac_s(acc, m[i], P29[k - i]);
        if (k < kNL) {
            m[k] = (0u - (uint32_t)acc) & kMask29;
            acc = (acc + kMask29) >> kLB;           // == (acc + m_k * p_0) >> 29, exact
        } else {
            r.l[k - kNL] = (uint32_t)acc & kMask29;
            acc >>= kLB;
        }
    }
    r.l[kNL - 1] = (uint32_t)acc;
    return r;
}

__device__ __forceinline__ F29 mont_mul(const F29 &a, const F29 &b) { return mont_fips<false>(a, b); }

// product with a wave-uniform constant (limbs in SGPRs)
__device__ __forceinline__ F29 mont_mul_const(const F29 &a, const uint32_t *c) {
    uint32_t m[kNL];
    F29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * kNL - 1; k++) {
        const int lo = k < kNL ? 0 : k - kNL + 1, hi = k < kNL ? k : kNL - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) mac_s(acc, a.l[i], c[k - i]);
#pragma unroll
        for (int i = lo; i <= hi; i++)
            if (k - i >= 1) mac_s(acc, m[i], P29[k - i]);
        if (k < kNL) {
            m[k] = (0u - (uint32_t)acc) & kMask29;
            acc = (acc + kMask29) >> kLB;
        } else {
            r.l[k - kNL] = (uint32_t)acc & kMask29;
            acc >>= kLB;
        }
    }
    r.l[kNL - 1] = (uint32_t)acc;
    return r;
}
__device__ __forceinline__ F29 mont_sqr(const F29 &a) { return mont_fips<true>(a, a); }

// v^5 / Rp^4
__device__ __forceinline__ F29 sbox29(const F29 &v) {
    F29 v2 = mont_sqr(v);
    F29 v4 = mont_sqr(v2);
    return mont_mul(v4, v);
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

// Y = C * X followed by a one-limb Montgomery step and carry normalisation of every row:
// st[i] <- (Y_i + m_i p) / 2^29 with limbs < 2^29.  Limb-major: five accumulators (one per output
// row) walk the limbs together, so input limb k of all five words dies at step k and output limb
// k-1 takes its place -- the layer needs ~15 registers beyond the state itself (row-major needs a
// second copy of the state).
// Input limbs < 2^30, C < 2^17: columns < 2^50 + 2^58; Y_i < 2^275 => result < 2^256.
__device__ __forceinline__ void small_mds(F29 (&st)[5]) {
    uint64_t acc[5];
    uint32_t m[5];
    uint32_t x[5];
#pragma unroll
    for (int j = 0; j < 5; j++) x[j] = st[j].l[0];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        acc[i] = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) mac_s(acc[i], x[j], MDS_SMALL[i][j]);
        m[i] = (0u - (uint32_t)acc[i]) & kMask29;
        acc[i] = (acc[i] + kMask29) >> kLB;
    }
#pragma unroll
    for (int k = 1; k < kNL; k++) {
#pragma unroll
        for (int j = 0; j < 5; j++) x[j] = st[j].l[k];
#pragma unroll
        for (int i = 0; i < 5; i++) {
#pragma unroll
            for (int j = 0; j < 5; j++) mac_s(acc[i], x[j], MDS_SMALL[i][j]);
            mac_s(acc[i], m[i], P29[k]);
            st[i].l[k - 1] = (uint32_t)acc[i] & kMask29;
            acc[i] >>= kLB;
        }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) st[i].l[kNL - 1] = (uint32_t)acc[i];
}

// One round.  ARK touches all five words in both round kinds (reference src/strategies.rs:86,
// :111); full rounds S-box every word, partial rounds the last word only, which then takes the
// rescale product.  `full` is wave-uniform, so the branches are scalar.
__device__ __forceinline__ void fast_round(const uint32_t *rec, bool full, F29 (&st)[5]) {
#pragma unroll
    for (int w = 0; w < 5; w++) add_lazy(st[w], load_f29(rec + w * kNL));
    if (full) {
        st[0] = sbox29(st[0]);
        st[1] = sbox29(st[1]);
        st[2] = sbox29(st[2]);
        st[3] = sbox29(st[3]);
    }
    st[4] = sbox29(st[4]);
    if (!full) st[4] = mont_mul_const(st[4], rec + 5 * kNL);
    small_mds(st);
#pragma unroll
    for (int w = 0; w < 5; w++)
#pragma unroll
        for (int k = 0; k < kNL; k++) limb_fence(st[w].l[k]);
}

// in: 5 BlsScalars (Montgomery 2^256 form, fully reduced); out: same format, fully reduced.
template <int NOUT>
__device__ __forceinline__ void fast_perm(const FastTables *T, const Fr (&in)[5], Fr (&out)[NOUT], int out_first) {
    F29 st[5];
#pragma unroll
    for (int w = 0; w < 5; w++) st[w] = to_f29(in[w]);
    // one loop, one body: every piece of round code exists once in the instruction stream, so the
    // whole kernel stays inside the instruction cache
#pragma unroll 1
    for (int r = 0; r < 67; r++) fast_round(T->round[r], r < 4 || r >= 63, st);
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
