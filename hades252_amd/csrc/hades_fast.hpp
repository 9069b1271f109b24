// hades_fast.hpp -- the scale-tracked Hades252 permutation kernel (the shipped hot path).
//
// Same field elements as the reference's ScalarStrategy::perm (src/strategies.rs:140-157,
// src/strategies/scalar.rs:23-49) -- hence the same bits after the final full reduction --
// with 13x fewer VALU instructions than the literal round structure:
//
//  1. Unsaturated radix 2^29, 9 signed limbs per element.  Measured on gfx950 (tools/ubench*.hip):
//     a 32x32+64 multiply-add (v_mad_i64_i32 / v_mad_u64_u32) issues at the same rate as a 64-bit
//     add or shift (~4 cycles per wave per SIMD), so the cost of a big-integer product is its
//     INSTRUCTION COUNT.  29-bit limbs leave headroom in a signed 64-bit column for 9 products + 8
//     reduction terms + carry, so every limb product is exactly one multiply-add in place.
//  2. Signed-digit Montgomery reduction interleaved with the product on ONE accumulator: p == 1
//     (mod 2^29), so the quotient digit is just the low 29 bits of the accumulator.
//  3. MDS with small integers.  The reference matrix is M[i][j] = 2^256/(i+j+5) mod p
//     (loader semantics of src/mds_matrix.rs:18-40) = lam * C with C[i][j] = 360360/(i+j+5)
//     < 2^17.  The kernel multiplies by C (9 multiply-adds per word instead of a 153 multiply-add
//     Montgomery product) and never applies lam: it is tracked as a known scale of the state.
//  4. Scale tracking.  Neither lam, nor the 1/Rp of each Montgomery product (Rp = 2^261), nor the
//     2^-29 of the one-limb reduction after each linear layer is ever applied; the running scale
//     s_r is folded into the round constants on the host (hades252_amd/_derive.py).  In partial
//     rounds one extra constant product K_r brings the S-boxed word back to the common scale.
//     One product with FINAL_F at the end returns value * 2^256, which is then fully reduced --
//     the unique in-memory BlsScalar.
//  5. Partial-round constants of words 0..3 are pushed through the linear layers on the host.
//
// Registers: state 5 x 9 VGPRs + ~35 for the product in flight (operands, nine quotient digits,
// one accumulator -- no column array).  Constants are wave-uniform: scalar loads (SMEM) into
// SGPRs, consumed directly as multiply-add operands.
#pragma once
#include "fr32.hpp"
#include "staging.hpp"

namespace hades {

constexpr int kLB = 29;                       // limb bits
constexpr int kNL = 9;                        // limbs
constexpr uint32_t kMask29 = (1u << kLB) - 1;

// Signed-limb element: value = sum l[k] * 2^(29k).  Normalised form: l[0..7] in [0, 2^29), l[8]
// signed (it carries the sign: values live in (-2^256, 2^256)).  Lazy form (after ARK): limbs in
// [-2^28, 2^29 + 2^28).
struct F29 {
    int32_t l[kNL];
};

__device__ static constexpr int32_t NEGP29[kNL] = HADES_NEG_P29;          // -p, limb by limb
__device__ static constexpr int32_t TWOP29[kNL] = HADES_TWO_P29;          // 2p, normalised limbs
__device__ static constexpr int32_t P29[kNL] = HADES_P29;                 // p, normalised limbs
__device__ static constexpr int32_t MDS_SMALL[5][5] = HADES_FAST_MDS_SMALL;

constexpr int kLinRow = 96;   // a linear-map table: 81 entries + pad (see mont_lin)
struct FastTables {
    int32_t round[67][64];    // per round {A[5][9] (balanced limbs), K[9], pad}
    int32_t final_f[kNL + 7];
    int32_t lin[67][kLinRow]; // per round: the linear-map table of K_r (partial rounds; zeros in full rounds)
    int32_t final_lin[kLinRow];   // ... and of FINAL_F
};

// One limb product accumulated in place; hipcc selects a single v_mad_i64_i32 for this shape as
// long as both factors are provably 32-bit (see limb_fence).
// The trailing input-only asm gives every partial sum a second use, which stops LLVM's
// reassociation from rebuilding the column as (p1 + p2 + ...) + carry -- that form needs a fresh
// chain from zero and an extra 64-bit add per column.  It emits no instruction.
__device__ __forceinline__ void pin(const int64_t &acc) { asm volatile("" ::"v"((uint32_t)acc)); }
__device__ __forceinline__ void mac(int64_t &acc, int32_t a, int32_t b) {
    acc += (int64_t)a * b;
    pin(acc);
}

// Zero-instruction fence on one limb: makes the value an opaque 32-bit VGPR.  Without it hipcc
// carries limbs across the round loop's back-edge as 64-bit values and every limb product of the
// next round turns into a 64 x 32 multiply: two mads and two moves.
__device__ __forceinline__ void limb_fence(int32_t &x) { asm volatile("" : "+v"(x)); }

// ---- 8 x 32 <-> 9 x 29 ---------------------------------------------------------------------
__device__ __forceinline__ F29 to_f29(const Fr &a) {
    F29 r;
#pragma unroll
    for (int k = 0; k < kNL; k++) {
        int bit = kLB * k, w = bit >> 5, sh = bit & 31;
        uint64_t two = a.l[w];
        if (w + 1 < 8) two |= (uint64_t)a.l[w + 1] << 32;
        r.l[k] = (int32_t)((uint32_t)(two >> sh) & kMask29);
    }
    return r;
}

// limbs must be normalised and non-negative (all < 2^29), value < 2^256
__device__ __forceinline__ Fr from_f29(const F29 &a) {
    Fr r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        int k = (32 * w) / kLB, sh = 32 * w - kLB * k;       // word w starts inside limb k at bit sh
        uint64_t acc = (uint64_t)(uint32_t)a.l[k] >> sh;
        int have = kLB - sh;
        if (k + 1 < kNL) acc |= (uint64_t)(uint32_t)a.l[k + 1] << have;
        have += kLB;
        if (have < 32 && k + 2 < kNL) acc |= (uint64_t)(uint32_t)a.l[k + 2] << have;
        r.l[w] = (uint32_t)acc;
    }
    return r;
}

// ---- Montgomery product, Rp = 2^261, signed digits ---------------------------------------------
// Finely integrated product scanning: ONE signed 64-bit accumulator walks the 17 columns; column k
// receives its limb products and the reduction terms -m_i * p_{k-i} of the quotient digits already
// known, then yields (k < 9) the next digit or (k >= 9) a result limb -- in both cases simply the
// low 29 bits of the accumulator -- and is shifted down arithmetically.
//   p == 1 (mod 2^29), so subtracting m_k * p with m_k = acc mod 2^29 clears the low limb: the
//   quotient digit costs one AND, there is no "+ m * p_0" step at all, and acc >> 29 is exact.
// Every limb product is one v_mad_i64_i32 on the accumulator.  Live registers: the two operands,
// nine digits and the accumulator -- no column array.
// Result = (a*b - M p) / Rp with M in [0, Rp): it lies in (a*b/Rp - p, a*b/Rp].
// Inputs: |limb| < 1.5 * 2^29, |value| < 2^257.  Output: normalised, value in (-p - 2^253, 2^253).
// Column bound: 9 * 2.25 * 2^58 + 8 * 2^58 + carry < 2^63.
template <bool SQR, bool CONST_B>
__device__ __forceinline__ F29 mont_fips(const F29 &a, const int32_t *b) {
    int32_t m[kNL];
    int32_t d[kNL];                       // 2 * a (squaring only)
    if constexpr (SQR) {
#pragma unroll
        for (int i = 0; i < kNL; i++) d[i] = a.l[i] * 2;
    }
    F29 r;
    int64_t acc = 0;
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
            for (int i = lo; i <= hi; i++) mac(acc, a.l[i], b[k - i]);
        }
#pragma unroll
        for (int i = lo; i <= hi; i++)
            if (k - i >= 1) mac(acc, m[i], NEGP29[k - i]);
        const int32_t low = (int32_t)((uint32_t)acc & kMask29);
        if (k < kNL)
            m[k] = low;
        else
            r.l[k - kNL] = low;
        acc >>= kLB;                          // arithmetic: exact for k < 9, floor for k >= 9
    }
    r.l[kNL - 1] = (int32_t)acc;
    return r;
}

// a * c / Rp for a one-limb constant 0 <= c < 2^29 (e.g. Rp / 2^256 = 32, the factor of to_bytes): the product part is
// one multiply-add per column instead of up to nine -- 81 multiply-adds in all instead of 153.  Same bounds and result
// range as mont_fips.
__device__ __forceinline__ F29 mont_mul_small(const F29 &a, int32_t c) {
    int32_t m[kNL];
    F29 r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * kNL - 1; k++) {
        const int lo = k < kNL ? 0 : k - kNL + 1, hi = k < kNL ? k : kNL - 1;
        if (k < kNL) mac(acc, a.l[k], c);
#pragma unroll
        for (int i = lo; i <= hi; i++)
            if (k - i >= 1) mac(acc, m[i], NEGP29[k - i]);
        const int32_t low = (int32_t)((uint32_t)acc & kMask29);
        if (k < kNL)
            m[k] = low;
        else
            r.l[k - kNL] = low;
        acc >>= kLB;
        // Ties the volatile statements to the data flow once per column: in a straight-line caller (k_wire) nothing else
        // does, and hipcc then runs the whole arithmetic first and the empty `pin` statements afterwards, which keeps
        // EVERY partial sum alive until then (197 VGPRs, or 536 bytes of scratch under launch bounds).  A read-write
        // statement cannot leave the chain, and the pins cannot cross it.
        asm volatile("" : "+v"(acc));
    }
    r.l[kNL - 1] = (int32_t)acc;
    return r;
}

// a * F / Rp for a wave-uniform CONSTANT F, as a linear map instead of a product: 97 multiply-adds instead of 153.
// A product with a constant is linear in the limbs of a:  a F 2^x = sum_k a_k (F 2^(29 k + x) mod p)  (mod p), so the
// host tabulates E_k = F 2^(29 (k - 7)) mod p (nine constants of nine non-negative 29-bit limbs; e[9 j + k] = limb j
// of E_k, column-major: column j's nine multipliers are contiguous for the scalar loads) and the kernel computes
//     W = sum_k a_k E_k          81 multiply-adds over NINE columns (a product has seventeen)
// followed by only TWO digit steps of the signed-digit reduction of mont_fips (W - (m_0 + 2^29 m_1) p) / 2^58 -- 16
// multiply-adds instead of 72 -- because W is already below 2^33 p: the constants were reduced on the host.
// The result is congruent to W / 2^58 = a F / 2^261, exactly what mont_fips(a, F) returns, and lies in
// (W / 2^58 - p (1 + 2^-29), W / 2^58].
// Input: normalised (limbs 0..7 in [0, 2^29), |top limb| < 2^25: the output of mont_fips / small_mds / to_f29).
// Output: normalised, value in (-p - 2^227, 2^230).  Column bound: 9 * 2^58 + 2 * 2^58 + carry < 2^62.
// hades252_amd/_derive.py::lin_table builds the table; tests/test_fast_model.py::mont_lin replays this limb for limb.
template <int STRIDE = kNL>   // dwords between the columns of the table (9: packed; 12: 16-byte aligned columns in LDS)
__device__ __forceinline__ F29 mont_lin(const F29 &a, const int32_t *e) {
    int32_t m0 = 0, m1 = 0;
    F29 r;
    int64_t acc = 0;
    // The 81 multipliers arrive by scalar loads, one column (nine contiguous dwords) ahead of the column being summed:
    // source order + a scheduling barrier per column pin that, so at most two columns of SGPRs are live (left to itself
    // the scheduler hoists what fits and then fetches the rest dword by dword, each behind its own wait).
    int32_t cur[kNL], nxt[kNL];
#pragma unroll
    for (int k = 0; k < kNL; k++) cur[k] = e[k];
#pragma unroll
    for (int j = 0; j < kNL; j++) {
        if (j + 1 < kNL) {
#pragma unroll
            for (int k = 0; k < kNL; k++) nxt[k] = e[STRIDE * (j + 1) + k];
        }
#pragma unroll
        for (int k = 0; k < kNL; k++) mac(acc, a.l[k], cur[k]);
        if (j >= 1) mac(acc, m0, NEGP29[j]);
        if (j >= 2) mac(acc, m1, NEGP29[j - 1]);
        const int32_t low = (int32_t)((uint32_t)acc & kMask29);
        if (j == 0)
            m0 = low;
        else if (j == 1)
            m1 = low;
        else
            r.l[j - 2] = low;
        acc >>= kLB;                          // exact for j < 2 (p == 1 mod 2^29), floor afterwards
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < kNL; k++) cur[k] = nxt[k];
    }
    mac(acc, m1, NEGP29[kNL - 1]);            // column 9: what is left of m_1 p
    r.l[kNL - 2] = (int32_t)((uint32_t)acc & kMask29);
    acc >>= kLB;
    r.l[kNL - 1] = (int32_t)acc;
    return r;
}

// The same with ONE digit step, for a result that only feeds the linear layer (K_r in a partial round): the table holds
// E_k = F 2^(29 (k - 8)) mod p, the result is (W - m_0 p) / 2^29, congruent to a F / 2^261 again, in (-1.04 p, 8.6 p) --
// nine limbs still (top limb < 2^26), and small_mds divides by 2^29 itself, so its output stays below 2^256.  89
// multiply-adds.  NOT for a value that is squared next (mont_fips wants |value| < 2^257).
__device__ __forceinline__ F29 mont_lin1(const F29 &a, const int32_t *e) {
    int32_t m0 = 0;
    F29 r;
    int64_t acc = 0;
    int32_t cur[kNL], nxt[kNL];
#pragma unroll
    for (int k = 0; k < kNL; k++) cur[k] = e[k];
#pragma unroll
    for (int j = 0; j < kNL; j++) {
        if (j + 1 < kNL) {
#pragma unroll
            for (int k = 0; k < kNL; k++) nxt[k] = e[kNL * (j + 1) + k];
        }
#pragma unroll
        for (int k = 0; k < kNL; k++) mac(acc, a.l[k], cur[k]);
        if (j >= 1) mac(acc, m0, NEGP29[j]);
        const int32_t low = (int32_t)((uint32_t)acc & kMask29);
        if (j == 0)
            m0 = low;
        else
            r.l[j - 1] = low;
        acc >>= kLB;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < kNL; k++) cur[k] = nxt[k];
    }
    r.l[kNL - 1] = (int32_t)acc;
    return r;
}

__device__ __forceinline__ F29 mont_mul(const F29 &a, const F29 &b) { return mont_fips<false, false>(a, b.l); }
__device__ __forceinline__ F29 mont_sqr(const F29 &a) { return mont_fips<true, false>(a, a.l); }
// product with a wave-uniform constant (limbs in SGPRs)
__device__ __forceinline__ F29 mont_mul_const(const F29 &a, const int32_t *c) { return mont_fips<false, true>(a, c); }

// v^5 / Rp^4
__device__ __forceinline__ F29 sbox29(const F29 &v) {
    F29 v2 = mont_sqr(v);
    F29 v4 = mont_sqr(v2);
    return mont_mul(v4, v);
}

__device__ __forceinline__ void add_lazy(F29 &x, const int32_t *c) {
#pragma unroll
    for (int k = 0; k < kNL; k++) x.l[k] += c[k];
}

// Y = C * X followed by a one-limb Montgomery step and carry normalisation of every row:
// st[i] <- (Y_i - m_i p) / 2^29, normalised.  Limb-major: five accumulators (one per output row)
// walk the limbs together, so input limb k of all five words dies at step k and output limb k-1
// takes its place -- the layer needs ~15 registers beyond the state itself (row-major needs a
// second copy of the state).
// |input limb| < 1.5 * 2^29, C < 2^17: |column| < 2^50 + 2^58; |Y_i| < 2^275 => |result| < 2^256.
__device__ __forceinline__ void small_mds(F29 (&st)[5]) {
    int64_t acc[5];
    int32_t m[5];
    int32_t x[5];
#pragma unroll
    for (int j = 0; j < 5; j++) x[j] = st[j].l[0];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        acc[i] = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) mac(acc[i], x[j], MDS_SMALL[i][j]);
        m[i] = (int32_t)((uint32_t)acc[i] & kMask29);
        acc[i] >>= kLB;
    }
#pragma unroll
    for (int k = 1; k < kNL; k++) {
#pragma unroll
        for (int j = 0; j < 5; j++) x[j] = st[j].l[k];
#pragma unroll
        for (int i = 0; i < 5; i++) {
#pragma unroll
            for (int j = 0; j < 5; j++) mac(acc[i], x[j], MDS_SMALL[i][j]);
            mac(acc[i], m[i], NEGP29[k]);
            st[i].l[k - 1] = (int32_t)((uint32_t)acc[i] & kMask29);
            acc[i] >>= kLB;
        }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) st[i].l[kNL - 1] = (int32_t)acc[i];
}

// One round.  The reference adds round keys to all five words in both round kinds
// (src/strategies.rs:86, :111); full rounds S-box every word, partial rounds the last word only,
// which then takes the rescale product.  `full` is wave-uniform, so the branches are scalar.
__device__ __forceinline__ void fast_round(const int32_t *rec, const int32_t *lin, bool full, F29 (&st)[5]) {
    // partial rounds: the constants of words 0..3 were pushed through the linear layers on the
    // host (hades252_amd/_derive.py), only word 4 receives one
    if (full) {
#pragma unroll
        for (int w = 0; w < 4; w++) add_lazy(st[w], rec + w * kNL);
    }
    add_lazy(st[4], rec + 4 * kNL);
    if (!full) {
        // Touch the six 64-byte lines of this round's linear-map table NOW: the column loads of mont_lin, an S-box later,
        // then hit the scalar cache.  A batch too small to put a second wave on a SIMD cannot hide a scalar-cache miss per
        // column behind another wave (measured on 4 096 states: 218 us per launch without this, 178 with the product form).
        const int32_t warm = lin[0] | lin[16] | lin[32] | lin[48] | lin[64] | lin[80];
        asm volatile("" ::"s"(warm));
    }
    if (full) {
        st[0] = sbox29(st[0]);
        st[1] = sbox29(st[1]);
        st[2] = sbox29(st[2]);
        st[3] = sbox29(st[3]);
    }
    st[4] = sbox29(st[4]);
    if (!full) st[4] = mont_lin1(st[4], lin);           // back to the common scale: x K_r / Rp as a linear map
    small_mds(st);
#pragma unroll
    for (int w = 0; w < 5; w++)
#pragma unroll
        for (int k = 0; k < kNL; k++) limb_fence(st[w].l[k]);
}

// x = mont_lin(state word, FINAL_F) or mont_mul_const(., .): normalised, value in (-p - 2^250, 2^250]
// -> the fully reduced BlsScalar of x mod p
__device__ __forceinline__ Fr finalize(const F29 &x) {
    // x + 2p lies in (p - 2^250, 2p + 2^250), below 2^256: carry-normalise (all limbs end
    // non-negative), then two conditional subtractions of p in the saturated 8 x 32 form
    F29 y;
    int32_t carry = 0;
#pragma unroll
    for (int k = 0; k < kNL - 1; k++) {
        int32_t v = x.l[k] + TWOP29[k] + carry;
        y.l[k] = (int32_t)((uint32_t)v & kMask29);
        carry = v >> kLB;
    }
    y.l[kNL - 1] = x.l[kNL - 1] + TWOP29[kNL - 1] + carry;
    return fr_cond_sub_p(fr_cond_sub_p(from_f29(y)));
}

// The same for x known to lie in (-p, p) -- mont_mul_small of a reduced input: (-p, 0]; mont_lin of non-negative limbs:
// (-p, 2^-25 p) -- where x + p is in (0, 2p): ONE conditional subtraction.
__device__ __forceinline__ Fr finalize1(const F29 &x) {
    F29 y;
    int32_t carry = 0;
#pragma unroll
    for (int k = 0; k < kNL - 1; k++) {
        int32_t v = x.l[k] + P29[k] + carry;
        y.l[k] = (int32_t)((uint32_t)v & kMask29);
        carry = v >> kLB;
    }
    y.l[kNL - 1] = x.l[kNL - 1] + P29[kNL - 1] + carry;
    return fr_cond_sub_p(from_f29(y));
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
    for (int r = 0; r < 67; r++) fast_round(T->round[r], T->lin[r], r < 4 || r >= 63, st);
    if constexpr (NOUT == 5) {
#pragma unroll
        for (int w = 0; w < 5; w++) out[w] = finalize(mont_lin(st[w], T->final_lin));
    } else {
        F29 sel = st[0];
#pragma unroll
        for (int w = 1; w < 5; w++)
            if (out_first == w) sel = st[w];
        out[0] = finalize(mont_lin(sel, T->final_lin));
    }
}

}  // namespace hades
