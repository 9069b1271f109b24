// hades_lanes.hpp -- the lane-split schedule: ONE field element spread over the lanes of a 16-lane DPP row,
// one permutation per wave.
//
// The throughput kernel (hades_fast.hpp) and the five-waves kernel (hades_coop.hpp) keep a whole element in one lane: a
// Montgomery product is then ~190 DEPENDENT instructions of that lane's wave, and a lone wave issues one instruction
// per ~4 cycles whatever it is -- that chain is what one permutation's latency is made of (the reference's real call
// shape is ONE permutation, README.md:60-61).  Here lane k of a row holds limb k (radix 2^29, nine limbs, lanes 9..15
// zero) and the product is done by the row:
//
//   T = a b       lane k sums column k:  acc_k = sum_i a_i b_{k-i}          9 multiply-adds  (a_i: row broadcast,
//                                                                               b_{k-i}: row shift right by i)
//   M = T p' mod 2^261   (p' = -1/p mod 2^261: nine limbs in SGPRs)            9 multiply-adds  (T shifted right by i)
//   R = (T + M p) / 2^261                                                       9 multiply-adds  (M shifted right by i)
//
// i.e. the digit-serial quotient of mont_fips (m_k needs column k complete) is replaced by the two-product form,
// whose columns are independent.  Everything is UNSIGNED (limbs, columns, values all >= 0: the round constants are
// stored with plain limbs here, not balanced ones), so a column may use all 64 bits: 9 (2^30 + 1)^2 < 2^64.
// Between the stages the 64-bit column sums go back to limbs in ONE lane-parallel step of 32-bit operations: a column
// is cut into bits 0..28 / 29..57 / 58..63, the middle piece moves one lane up and the top piece two -- limbs come out
// LAZY (below 2^30 + 64), which is all a multiply-add operand needs; nothing is ever propagated serially across the
// row.  A product's result keeps those lazy limbs: 9 (2^30 + 66)^2 still fits 64 bits, so x^2 and x^4 go straight into
// the next product; only the linear layer (whose output meets a round key before it is squared) normalises to 2^29 + 2.
// 17 columns (0..16) on 16 lanes: column 16 (and 17, times 2^29) lives in lane 0 of a 64-bit "top" accumulator, zero
// in every other lane by construction (it is only ever fed by row_shl:8 / row_shl:15 moves, which zero-fill).
// Exact division by 2^261: after the carry step the nine low limbs represent 0, 2^261 or 2 * 2^261 exactly (their value
// is congruent to 0 and below 2^262 (1 + 2^-24)); limb 8 alone tells which (it is c 2^29 minus at most 3), and c is
// carried into limb 9.
//
// ~110 instructions per product instead of ~190, same field element: R = a b / 2^261 mod p, 0 <= R < ab/2^261 + 2.01 p.
// tests/test_fast_model.py::lanes_* replays this file limb for limb in Python with the 64-bit column bounds asserted.
#pragma once
#include "hades_fast.hpp"

namespace hades {

// DPP controls (gfx9 encoding; row = 16 lanes)
constexpr int kRowShl = 0x100, kRowShr = 0x110, kRowBcast = 0x150;   // + n; 0x150 + n = row_newbcast:n (gfx90a+)

// lane k <- lane k - N of the same row (zero for k < N)
template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t v) {
    if constexpr (N == 0) return v;
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kRowShr + N, 0xF, 0xF, true);
}
// lane k <- lane k + N of the same row (zero for k + N > 15)
template <int N>
__device__ __forceinline__ uint32_t row_shl(uint32_t v) {
    if constexpr (N == 0) return v;
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kRowShl + N, 0xF, 0xF, true);
}
// every lane of the row <- lane N of the row
template <int N>
__device__ __forceinline__ uint32_t row_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kRowBcast + N, 0xF, 0xF, true);
}

// one unsigned limb product accumulated in place: a single v_mad_u64_u32.  The empty asm gives every partial sum a second
// use, which stops LLVM's reassociation from rebuilding a column as (p1 + p2 + ...) + init with an extra 64-bit add
// (see `pin` in hades_fast.hpp).  It is volatile on purpose: the multiply-adds then stay in SOURCE order, and the
// source order of lane_mont_mul_n is the hand-made schedule (measured: letting the scheduler move them costs 5 %).
__device__ __forceinline__ void umac(uint64_t &acc, uint32_t a, uint32_t b) {
    acc += (uint64_t)a * b;
    asm volatile("" ::"v"((uint32_t)acc));
}

// wave-uniform constants of the reduction: p and p' = -p^-1 mod 2^261, nine 29-bit limbs each (SGPRs)
struct LaneConsts {
    uint32_t p[kNL];
    uint32_t pinv[kNL];
};

// nine operand forms of a variable (shifted right by 0..8 lanes, or lane 0..8 broadcast)
struct LaneForms {
    uint32_t s[kNL];
};
__device__ __forceinline__ LaneForms lane_shifts(uint32_t b) {
    LaneForms r;
    r.s[0] = b;
    r.s[1] = row_shr<1>(b);
    r.s[2] = row_shr<2>(b);
    r.s[3] = row_shr<3>(b);
    r.s[4] = row_shr<4>(b);
    r.s[5] = row_shr<5>(b);
    r.s[6] = row_shr<6>(b);
    r.s[7] = row_shr<7>(b);
    r.s[8] = row_shr<8>(b);
    return r;
}
__device__ __forceinline__ LaneForms lane_bcasts(uint32_t a) {
    LaneForms r;
    r.s[0] = row_bcast<0>(a);
    r.s[1] = row_bcast<1>(a);
    r.s[2] = row_bcast<2>(a);
    r.s[3] = row_bcast<3>(a);
    r.s[4] = row_bcast<4>(a);
    r.s[5] = row_bcast<5>(a);
    r.s[6] = row_bcast<6>(a);
    r.s[7] = row_bcast<7>(a);
    r.s[8] = row_bcast<8>(a);
    return r;
}

// One lane-parallel carry step over columns 0..15 (acc), 32-bit operations only:
//   acc = lo + 2^29 mid + 2^58 tp  (lo, mid < 2^29, tp < 64);   limb_k = lo_k + mid_{k-1} + tp_{k-2}  < 2^30 + 64
// What leaves lane 15 belongs to columns 16 (u_15 = mid_15 + tp_14) and 17 (tp_15): returned per lane in c16 / c17 for
// the caller to collect (only lane 15's values count).
__device__ __forceinline__ uint32_t carry_split(uint64_t acc, uint32_t &c16, uint32_t &c17) {
    const uint32_t lo32 = (uint32_t)acc, hi32 = (uint32_t)(acc >> 32);
    const uint32_t lo = lo32 & kMask29;
    const uint32_t mid = __builtin_amdgcn_alignbit(hi32, lo32, kLB) & kMask29;
    const uint32_t tp = hi32 >> (58 - 32);
    const uint32_t u = mid + row_shr<1>(tp);
    c16 = u;
    c17 = tp;
    return lo + row_shr<1>(u);
}
// the same for columns known to be below 2^61 (the linear layer): limb_k = lo_k + (acc_{k-1} >> 29)  < 2^29 + 2^32
__device__ __forceinline__ uint32_t carry_split2(uint64_t acc) {
    const uint32_t lo32 = (uint32_t)acc, hi32 = (uint32_t)(acc >> 32);
    return (lo32 & kMask29) + row_shr<1>(__builtin_amdgcn_alignbit(hi32, lo32, kLB));
}

// second, light pass on limbs below 2^32: limb_k = (t_k mod 2^29) + (t_{k-1} >> 29); lane 15's carry (column 16) is
// returned per lane in c16.  For t < 2^30 + 64 the result is <= 2^29 + 2.
__device__ __forceinline__ uint32_t carry_light(uint32_t t, uint32_t &c16) {
    const uint32_t h = t >> kLB;
    c16 = h;
    return (t & kMask29) + row_shr<1>(h);
}

// N independent products side by side, R[n] = a[n] b[n] / 2^261 (mod p): the statements of the N chains alternate, so
// one chain's instructions fill the wait states the other's DPP moves and multiply-adds need (a lone chain spends a
// tenth of its issue slots in s_nop).
// a, b: lane k = limb k (k < 9), lanes 9..15 zero, limbs <= 2^30 + 66, values < 2^257.
// as = lane_bcasts(a), bs = lane_shifts(b): kept by the caller when an operand is used again (x^5 = x * x^4 reuses
// the broadcasts of x).  Result limbs <= 2^30 + 66 (limb 8 < 2^26), 0 <= R < ab/2^261 + 2.01 p.
template <int N>
__device__ __forceinline__ void lane_mont_mul_n(const LaneConsts &K, const LaneForms (&as)[N], const uint32_t (&b)[N],
                                                const LaneForms (&bs)[N], uint32_t (&out)[N]) {
    const int lane = threadIdx.x & 15;
    // ---- T = a b: columns 0..15 in acc, column 16 in lane 0 of top
    uint64_t acc[N], top[N];
#pragma unroll
    for (int n = 0; n < N; n++) acc[n] = 0;
#pragma unroll
    for (int i = 0; i < kNL; i++)
#pragma unroll
        for (int n = 0; n < N; n++) umac(acc[n], as[n].s[i], bs[n].s[i]);
    uint32_t t[N], c16a[N], c17[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
        top[n] = (uint64_t)as[n].s[kNL - 1] * row_shl<8>(b[n]);     // lane 0: a_8 b_8; other lanes: 0
        t[n] = carry_split(acc[n], c16a[n], c17[n]);
    }
    // ---- M = T p' mod 2^261: only lanes 0..8 count; what the carries push beyond lane 8 is a multiple of 2^261
    LaneForms ts[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
        ts[n] = lane_shifts(t[n]);
        acc[n] = 0;
    }
#pragma unroll
    for (int i = 0; i < kNL; i++)
#pragma unroll
        for (int n = 0; n < N; n++) umac(acc[n], ts[n].s[i], K.pinv[i]);
    uint32_t m[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
        uint32_t d0, d1;
        m[n] = carry_split(acc[n], d0, d1);
        m[n] = lane < kNL ? m[n] : 0;
    }
    // ---- S = T + M p: the low nine limbs cancel to 0 or 2^261
    LaneForms ms[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
        ms[n] = lane_shifts(m[n]);
        acc[n] = (uint64_t)t[n];
    }
#pragma unroll
    for (int i = 0; i < kNL; i++)
#pragma unroll
        for (int n = 0; n < N; n++) umac(acc[n], ms[n].s[i], K.p[i]);
#pragma unroll
    for (int n = 0; n < N; n++) {
        umac(top[n], row_shl<8>(m[n]), K.p[kNL - 1]);                // column 16: + m_8 p_8 (lane 0)
        uint32_t c16b;
        uint32_t w = carry_split(acc[n], c16b, c17[n]);
        // residual carry of the low part into limb 9: the nine low limbs (each < 2^30 + 64) represent 0, 2^261 or
        // 2 * 2^261 exactly, and limb 8 alone tells which: it is c 2^29 minus at most 3
        const uint32_t z = lane == kNL - 1 ? (w + 3u) >> kLB : 0;
        w += row_shr<1>(z);
        // ---- column 16: everything that left lane 15, collected in lane 0.  Nothing reaches column 17 directly: column
        // 15 holds two products with a top limb (a_7 b_8 + a_8 b_7 < 2^57, t_15 + m_7 p_8 + m_8 p_7 < 2^58), so its top
        // piece (c17) is zero.
        umac(top[n], row_shl<15>(c16a[n] + c16b), 1);
        // ---- R: limbs 0..6 = lanes 9..15 of w, limb 7 = low 29 bits of top, limb 8 = the rest (small)
        const uint32_t tlo = (uint32_t)top[n], thi = (uint32_t)(top[n] >> 32);
        const uint32_t r7 = tlo & kMask29;                           // zero outside lane 0
        const uint32_t r8 = __builtin_amdgcn_alignbit(thi, tlo, kLB);
        out[n] = row_shl<9>(w) + row_shr<7>(r7) + row_shr<8>(r8);
    }
}

__device__ __forceinline__ uint32_t lane_mont_mul(const LaneConsts &K, const LaneForms &as, uint32_t b, const LaneForms &bs) {
    const LaneForms a1[1] = {as}, b1[1] = {bs};
    const uint32_t bb[1] = {b};
    uint32_t r[1];
    lane_mont_mul_n<1>(K, a1, bb, b1, r);
    return r[0];
}
__device__ __forceinline__ uint32_t lane_mont_mul(const LaneConsts &K, uint32_t a, uint32_t b) {
    return lane_mont_mul(K, lane_bcasts(a), b, lane_shifts(b));
}
__device__ __forceinline__ uint32_t lane_mont_sqr(const LaneConsts &K, uint32_t a) { return lane_mont_mul(K, a, a); }

// a F / 2^261 (mod p) for a CONSTANT F, as a linear map by the row (the lane form of mont_lin, hades_fast.hpp): the table
// holds E_k = F 2^(29 (k - 7)) mod p as PER-LANE constants (e[k]: lane j = limb j of E_k, zero for j >= 9), so
//   W = sum_k a_k E_k           lane j sums column j: 9 multiply-adds with a_k broadcast by the row
//   M = W p' mod 2^58           two limbs: 2 multiply-adds (W, and W shifted by one lane)
//   R = (W + M p) / 2^58        2 multiply-adds (m_0, m_1 broadcast; pk = limb k of p per lane, pk1 = the same shifted)
// 13 multiply-adds, ~12 DPP moves and three carry steps instead of the 28 / ~45 / three of lane_mont_mul_n: about half
// a product.  After the last carry step the two low limbs are an exact multiple of 2^58 (w_0 = 0, w_1 = c 2^29), c is
// carried into limb 2 and the row moves down two lanes.  Same operand / result bounds as lane_mont_mul_n.
// tests/test_fast_model.py::lane_lin replays this statement by statement.
__device__ __forceinline__ uint32_t lane_lin(const LaneConsts &K, uint32_t a, const uint32_t (&e)[kNL], uint32_t pk, uint32_t pk1) {
    const int lane = threadIdx.x & 15;
    uint64_t acc = 0;
    umac(acc, row_bcast<0>(a), e[0]);
    umac(acc, row_bcast<1>(a), e[1]);
    umac(acc, row_bcast<2>(a), e[2]);
    umac(acc, row_bcast<3>(a), e[3]);
    umac(acc, row_bcast<4>(a), e[4]);
    umac(acc, row_bcast<5>(a), e[5]);
    umac(acc, row_bcast<6>(a), e[6]);
    umac(acc, row_bcast<7>(a), e[7]);
    umac(acc, row_bcast<8>(a), e[8]);
    uint32_t c16, c17;
    const uint32_t t = carry_split(acc, c16, c17);
    uint64_t am = 0;
    umac(am, t, K.pinv[0]);
    umac(am, row_shr<1>(t), K.pinv[1]);
    const uint32_t m = carry_split(am, c16, c17);                    // lanes 0 and 1 count
    acc = (uint64_t)t;
    umac(acc, row_bcast<0>(m), pk);
    umac(acc, row_bcast<1>(m), pk1);
    uint32_t w = carry_split(acc, c16, c17);
    const uint32_t z = lane == 1 ? w >> kLB : 0;
    w += row_shr<1>(z);
    return row_shl<2>(w);
}

// v^5 / Rp^4 for N independent values side by side
template <int N>
__device__ __forceinline__ void lane_sbox_n(const LaneConsts &K, uint32_t (&v)[N]) {
    LaneForms vb[N], fs[N], fb[N];
    uint32_t v2[N], v4[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
        vb[n] = lane_bcasts(v[n]);
        fs[n] = lane_shifts(v[n]);
    }
    lane_mont_mul_n<N>(K, vb, v, fs, v2);
#pragma unroll
    for (int n = 0; n < N; n++) {
        fb[n] = lane_bcasts(v2[n]);
        fs[n] = lane_shifts(v2[n]);
    }
    lane_mont_mul_n<N>(K, fb, v2, fs, v4);
#pragma unroll
    for (int n = 0; n < N; n++) fs[n] = lane_shifts(v4[n]);
    lane_mont_mul_n<N>(K, vb, v4, fs, v);                            // x * x^4: the broadcasts of x are reused
}

// v^5 / Rp^4 (same scale as sbox29)
__device__ __forceinline__ uint32_t lane_sbox(const LaneConsts &K, uint32_t v) {
    uint32_t x[1] = {v};
    lane_sbox_n<1>(K, x);
    return x[0];
}

// every row of the wave <- row ROW (0 or 1) of v: two half-exchanges, no LDS round trip.
//   v_permlane16_swap d, s:  rows 1, 3 of d <-> rows 0, 2 of s;   v_permlane32_swap d, s:  rows 2, 3 of d <-> rows 0, 1 of s
template <int ROW>
__device__ __forceinline__ uint32_t wave_bcast_row(uint32_t v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);    // {(v0, v0, v2, v2), (v1, v1, v3, v3)}
    const uint32_t h = ROW == 0 ? r[0] : r[1];
    const auto q = __builtin_amdgcn_permlane32_swap(h, h, false, false);
    return q[0];
}

// One output row of the small-integer linear layer followed by the one-limb Montgomery step (small_mds_row of
// hades_coop.hpp in lane form):  y = (sum_j c_j x_j + m p) / 2^29,  m = -Y_0 mod 2^29.
// x[j]: limb k of word j in lane k; c[j] < 2^17; pk = limb k of p in lane k (zero in lanes 9..15).
// Columns < 5 * 2^17 * (2^30 + 66) + 2^58 < 2^59; result limbs <= 2^29 + 2, top limb < 2^24.
__device__ __forceinline__ uint32_t lane_mds_row(const uint32_t (&c)[5], const uint32_t (&x)[5], uint32_t pk) {
    uint64_t y = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) umac(y, x[j], c[j]);
    const uint32_t m = row_bcast<0>((0u - (uint32_t)y) & kMask29);
    umac(y, m, pk);
    uint32_t unused;
    const uint32_t w = carry_light(carry_split2(y), unused);         // column 0 is an exact multiple of 2^29: w_0 = 0
    return row_shl<1>(w);                                            // / 2^29
}

// ---- one permutation per wave ---------------------------------------------------------------------------------
// The four rows of a wave hold the five words of ONE state: register A = words {4, 0, 1, 2} in rows {0, 1, 2, 3},
// register B = word 3 (row 1 is the copy that counts; the other rows compute along).  The schedule is hades_coop.hpp's
// (hades252_amd/_derive.py::coop_schedule: every round moves the common scale s -> s^5 / Rp^4 / (lam 2^29)):
//   full round     A <- S-box(A + c) in all four rows at once, then B <- S-box(B + c): 6 products
//   partial round  three product slots: row 0 runs the S-box of word 4 (x^2, x^4, x x^4) while rows 1..3 lift words
//                  0, 1, 2 to word 4's new scale with G_r in slot 1 and row 1 lifts word 3 in slot 2 -- one instruction
//                  stream, per-row operands: 3 products
//   linear layer   the words meet in LDS (the wave's own 512 bytes; a wave's LDS operations execute in order, so no
//                  barrier), every row computes the output row(s) of its own word(s): two passes of lane_mds_row
// Constants come from global memory by per-lane loads one round ahead (a round is ~1500 cycles: cold misses hide).
struct LanesTables {
    uint32_t round[68][64];     // per round {A[5][9] plain limbs, G[9], zeros}; row 67 is all zero (fetched ahead by the
                                // last round, never used)
    int32_t final_f[kNL + 7];   // mont(X, final_f) = x * 2^256 (per-lane arithmetic of the last step)
    uint32_t mds[5][8];         // small-integer MDS rows
    uint32_t p16[16];           // limb k of p in lane k, zero for k >= 9
    uint32_t p[kNL], pinv[kNL];
};
struct LanesLds {               // per wave
    // [parity][limb][word]: words 0..4 + three dummy slots (row-private copies of B).  Two buffers: in the helped form a
    // state wave and the helper meet at ONE barrier per full round, so the buffer of round r may still be being read by
    // the peer while the other side already publishes for round r + 1 -- full round r uses xw[r & 1] (what
    // hades_coop.hpp does with its ping-pong buffers); the partial rounds and both hand-overs of word 3 use xw[0], which
    // the helper last read in round 2 (before the barrier of round 3) and reads again only after the hand-over barrier.
    // With that no access of one side can overtake the other's by less than a full barrier, whatever the timing.
    uint32_t xw[2][16][8];
    uint32_t io[8][16];         // [word][limb]: 5 words + dummy rows
};

__device__ __forceinline__ void lanes_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Number of block-wide barriers a helped main wave / the helper wave / an idle wave of the block goes through:
// input hand-over, one per full round (8), the hand-over of word 3 before the trailing full rounds, output hand-over.
constexpr int kLanesBarriers = 11;

// in: this wave's state as five BlsScalars in lanes 0..4 (lane w = word w; other lanes ignored);
// out: the permuted words, fully reduced, in lanes 0..4.
// HELPED = false: the wave does everything itself (register B = word 3 runs its S-box interleaved with A's in the full
//   rounds).  HELPED = true: a helper wave of the same block (lanes_helper) owns word 3 during the eight full rounds --
//   its S-box then runs on another SIMD at the same time instead of doubling this wave's instruction stream; the two
//   meet at one block-wide barrier per full round (the words are exchanged through this state's LDS area anyway), and
//   word 3 changes hands before round 4 (this wave computes its row itself in round 3), before round 63 and at the end.
//   Every wave of the block must then pass kLanesBarriers barriers.
// stamps (diagnostic builds only, tools/lanes_proto.hip): shader-clock readings after the input stage and after the rounds
template <bool HELPED>
__device__ __forceinline__ Fr lanes_perm(const LanesTables *T, LanesLds &L, const Fr &in,
                                         unsigned long long *stamps = nullptr) {
    const int lane = threadIdx.x & 63, row = lane >> 4, k = lane & 15;
    const int word_a = row == 0 ? 4 : row - 1;
    const bool row0 = row == 0;
    LaneConsts K;
#pragma unroll
    for (int i = 0; i < kNL; i++) {
        K.p[i] = T->p[i];
        K.pinv[i] = T->pinv[i];
    }
    // per-lane constants: offsets into a round record (lanes 9..15 read the record's zero padding), MDS row, p limb
    const int off_a = k < kNL ? word_a * kNL + k : 54, off_b = k < kNL ? 3 * kNL + k : 54, off_g = k < kNL ? 5 * kNL + k : 54;
    uint32_t ca_m[5], cb_m[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        ca_m[j] = T->mds[word_a][j];
        cb_m[j] = T->mds[3][j];
    }
    const uint32_t pk = T->p16[k];
    const int slot_b = row == 1 ? 3 : (row == 0 ? 5 : 4 + row);      // rows 0, 2, 3 park their copy of B in 5, 6, 7
    // ---- in: lanes 0..4 convert their word to nine limbs, the rows pick them up through LDS
    if (lane < 5) {
        const F29 f = to_f29(in);
#pragma unroll
        for (int i = 0; i < 16; i++) L.io[lane][i] = i < kNL ? (uint32_t)f.l[i] : 0u;
    }
    lanes_fence();
    if constexpr (HELPED) __syncthreads();                           // the helper picks up word 3
    uint32_t A = L.io[word_a][k], B = L.io[3][k];
    // round r's constants of this lane sit at rec[off_*]; rec advances by one 256-byte record per round (a scalar
    // pointer: the per-lane part of the address never changes); every round fetches the next round's constants first
    const uint32_t *rec = &T->round[0][0];
    uint32_t c_a = rec[off_a], c_b = rec[off_b], c_g = rec[off_g];
    if (stamps != nullptr) {
        asm volatile("" : "+v"(A), "+v"(B), "+v"(c_a), "+v"(c_b), "+v"(c_g));
        stamps[0] = __builtin_amdgcn_s_memtime();
    }
    // Three loops (4 full, 59 partial, 4 full rounds: src/strategies.rs:144-156), each with a straight-line body.
    auto full_round = [&](bool take_word3, int par) {                // par: exchange buffer = round & 1 (helped form)
        uint32_t (&xw)[16][8] = L.xw[HELPED ? par : 0];
        rec += 64;
        const uint32_t n_a = rec[off_a], n_b = rec[off_b], n_g = rec[off_g];
        unsigned long long ts0 = 0, ts1 = 0;
        if (stamps != nullptr) ts0 = __builtin_amdgcn_s_memtime();
        uint32_t x[5];
        A += c_a;
        if constexpr (HELPED) {
            A = lane_sbox(K, A);                                                      // words 4, 0, 1, 2
            xw[k][word_a] = A;
            __syncthreads();                                                          // the helper has stored word 3
            const uint4 q = *reinterpret_cast<const uint4 *>(&xw[k][0]);
            x[4] = xw[k][4];
            x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
        } else {
            B += c_b;
            uint32_t v[2] = {A, B};                                                  // words 4, 0, 1, 2 and word 3:
            lane_sbox_n<2>(K, v);                                                     // two S-boxes, statements interleaved
            A = v[0];
            B = v[1];
            xw[k][word_a] = A;
            lanes_fence();
            const uint4 q = *reinterpret_cast<const uint4 *>(&xw[k][0]);            // .w (word 3) is stale: not used
            x[4] = xw[k][4];
            x[0] = q.x; x[1] = q.y; x[2] = q.z;
            x[3] = wave_bcast_row<1>(B);
        }
        if (stamps != nullptr) {
            asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]));
            ts1 = __builtin_amdgcn_s_memtime();
        }
        A = lane_mds_row(ca_m, x, pk);
        if (!HELPED || take_word3) B = lane_mds_row(cb_m, x, pk);    // helped: word 3 stays with the helper between full rounds
        if (stamps != nullptr) {
            asm volatile("" : "+v"(A), "+v"(B));
            stamps[4] += ts1 - ts0;
            stamps[5] += __builtin_amdgcn_s_memtime() - ts1;
        }
        c_a = n_a;
        c_b = n_b;
        c_g = n_g;
        asm volatile("" : "+v"(A), "+v"(B));
    };
    // Partial round: the words that are ready early meet in LDS while word 4's last product is still running (its
    // ~500 cycles hide the round trip); the late word crosses the rows by two half-exchanges.
    auto partial_round = [&]() {
        rec += 64;
        const uint32_t n_a = rec[off_a], n_b = rec[off_b], n_g = rec[off_g];
        unsigned long long ts0 = 0, ts1 = 0;
        if (stamps != nullptr) ts0 = __builtin_amdgcn_s_memtime();
        uint32_t x[5];
        A += c_a;                                                                     // (words 0..3 have no constant here)
        const LaneForms ab = lane_bcasts(A);
        const uint32_t b1 = row0 ? A : c_g;
        const uint32_t p1 = lane_mont_mul(K, ab, b1, lane_shifts(b1));            // row 0: x^2;  rows 1..3: w G
        const uint32_t a2 = row0 ? p1 : B, b2 = row0 ? p1 : c_g;
        const uint32_t p2 = lane_mont_mul(K, lane_bcasts(a2), b2, lane_shifts(b2));       // row 0: x^4;  row 1: w_3 G
        L.xw[0][k][row0 ? 5 : word_a] = p1;                                           // words 0, 1, 2 (row 0 parks x^2)
        L.xw[0][k][slot_b] = p2;                                                      // word 3
        lanes_fence();
        const uint4 q = *reinterpret_cast<const uint4 *>(&L.xw[0][k][0]);
        x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
        const uint32_t p3 = lane_mont_mul(K, ab, p2, lane_shifts(p2));            // row 0: x x^4
        x[4] = wave_bcast_row<0>(p3);
        if (stamps != nullptr) {
            asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]));
            ts1 = __builtin_amdgcn_s_memtime();
        }
        A = lane_mds_row(ca_m, x, pk);
        B = lane_mds_row(cb_m, x, pk);
        if (stamps != nullptr) {
            asm volatile("" : "+v"(A), "+v"(B));
            stamps[2] += ts1 - ts0;
            stamps[3] += __builtin_amdgcn_s_memtime() - ts1;
        }
        c_a = n_a;
        c_b = n_b;
        c_g = n_g;
        asm volatile("" : "+v"(A), "+v"(B));
    };
#pragma unroll 1
    for (int r = 0; r < 3; r++) full_round(false, r & 1);
    full_round(true, 1);                                             // round 3: this wave takes word 3 for the partial rounds
#pragma unroll 1
    for (int r = 4; r < 63; r++) partial_round();
    if constexpr (HELPED) {                                          // word 3 goes back to the helper
        L.xw[0][k][slot_b] = B;
        __syncthreads();
    }
#pragma unroll 1
    for (int r = 63; r < 67; r++) full_round(false, r & 1);
    if (stamps != nullptr) stamps[1] = __builtin_amdgcn_s_memtime();
    // ---- out: the words go back to one lane each for the final product and the full reduction
    if constexpr (HELPED) __syncthreads();                           // the helper has stored the final word 3 in io[3]
    L.io[word_a][k] = A;
    if constexpr (!HELPED) L.io[row == 1 ? 3 : 5][k] = B;
    lanes_fence();
    F29 f;
#pragma unroll
    for (int i = 0; i < kNL; i++) f.l[i] = (int32_t)L.io[lane < 5 ? lane : 0][i];
    return finalize(mont_mul_const(f, T->final_f));
}

// The helper wave of a block of NSTATES helped main waves: row s owns word 3 of state s during the full rounds.
template <int NSTATES>
__device__ __forceinline__ void lanes_helper(const LanesTables *T, LanesLds (&Ls)[NSTATES]) {
    const int lane = threadIdx.x & 63, row = lane >> 4, k = lane & 15;
    LanesLds &L = Ls[row < NSTATES ? row : NSTATES - 1];            // a spare row works along on the last state ...
    const int slot = row < NSTATES ? 3 : 7;                          // ... and parks its results in a dummy slot
    LaneConsts K;
#pragma unroll
    for (int i = 0; i < kNL; i++) {
        K.p[i] = T->p[i];
        K.pinv[i] = T->pinv[i];
    }
    const int off_b = k < kNL ? 3 * kNL + k : 54;
    uint32_t cb_m[5];
#pragma unroll
    for (int j = 0; j < 5; j++) cb_m[j] = T->mds[3][j];
    const uint32_t pk = T->p16[k];
    // word 3's round constants of the eight full rounds, fetched up front: a load issued when the constant is needed
    // would make this wave late at every barrier
    uint32_t cb[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cb[i] = T->round[i < 4 ? i : 59 + i][off_b];
    __syncthreads();                                                 // the main waves have stored their inputs
    uint32_t H = L.io[3][k];
#pragma unroll 1
    for (int i = 0; i < 8; i++) {
        if (i == 4) {                                                // partial rounds: word 3 is with the main waves
            __syncthreads();
            H = L.xw[0][k][3];
        }
        uint32_t (&xw)[16][8] = L.xw[i < 4 ? (i & 1) : ((i + 1) & 1)];   // rounds 0..3 and 63..66: buffer = round & 1
        uint32_t c = cb[0];
#pragma unroll
        for (int j = 1; j < 8; j++) c = i == j ? cb[j] : c;
        H += c;
        H = lane_sbox(K, H);
        xw[k][slot] = H;
        __syncthreads();
        uint32_t x[5];
        const uint4 q = *reinterpret_cast<const uint4 *>(&xw[k][0]);
        x[4] = xw[k][4];
        x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
        H = lane_mds_row(cb_m, x, pk);
        asm volatile("" : "+v"(H));
    }
    L.io[row < NSTATES ? 3 : 7][k] = H;
    __syncthreads();
}

// an idle wave of a helped block (no state to work on) only keeps the barrier count
__device__ __forceinline__ void lanes_idle() {
    for (int i = 0; i < kLanesBarriers; i++) __syncthreads();
}

// ---- four permutations per wave: one state per ROW ---------------------------------------------------------------
// For batches too large for one wave per state (more than one wave per SIMD would queue) and too small for one state per
// lane: row s of the wave holds state s, its five words in five registers (lane k = limb k).  No word of a state ever
// leaves its row, so there is no exchange at all; the words run through the products one after the other.  With nothing
// idle to lift words 0..3, the schedule is the THROUGHPUT kernel's (hades252_amd/_derive.py::fast_schedule): word 4
// comes back to the common scale with K_r -- four products per partial round, fifteen per full round -- and words 0..3
// meet only the linear layer in the partial rounds (their constants are pushed through it, effective_constants()).
// Tables: LanesTables with HADES_ROWS_ROUND_INIT {A[5][9] plain limbs, K[9], zeros} and the fast schedule's final factor.
// Limb-exact Python replay with the word bounds asserted: tests/test_fast_model.py::rows_perm_model.
struct RowsLds {                // per wave
    uint32_t io[4][5][16];      // [state][word][limb]
};

// in: lane 5 s + w (s < 4, w < 5) holds word w of the wave's state s as a BlsScalar (other lanes ignored);
// out: the permuted words, fully reduced, in the same lanes.
// klin: HADES_ROWS_KLIN_INIT, [59][9][16]: K_r of partial round 4 + i as the per-lane constants of lane_lin
__device__ __forceinline__ Fr rows_perm(const LanesTables *T, const uint32_t (*klin)[kNL][16], RowsLds &L, const Fr &in) {
    const int lane = threadIdx.x & 63, row = lane >> 4, k = lane & 15;
    LaneConsts K;
#pragma unroll
    for (int i = 0; i < kNL; i++) {
        K.p[i] = T->p[i];
        K.pinv[i] = T->pinv[i];
    }
    uint32_t cm[5][5];
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) cm[i][j] = T->mds[i][j];
    const uint32_t pk = T->p16[k], pk1 = row_shr<1>(pk);
    if (lane < 20) {
        const F29 f = to_f29(in);
        const int s_in = lane / 5, w_in = lane - 5 * s_in;
#pragma unroll
        for (int i = 0; i < 16; i++) L.io[s_in][w_in][i] = i < kNL ? (uint32_t)f.l[i] : 0u;
    }
    lanes_fence();
    uint32_t w[5], c[5], ck;
#pragma unroll
    for (int j = 0; j < 5; j++) w[j] = L.io[row][j][k];
    const uint32_t *rec = &T->round[0][0];
    auto fetch = [&](uint32_t (&cc)[5], uint32_t &kk) {
#pragma unroll
        for (int j = 0; j < 5; j++) cc[j] = rec[k < kNL ? j * kNL + k : 54];
        kk = rec[k < kNL ? 5 * kNL + k : 54];
    };
    fetch(c, ck);
    auto linear = [&]() {
        uint32_t y[5];
#pragma unroll
        for (int i = 0; i < 5; i++) y[i] = lane_mds_row(cm[i], w, pk);
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = y[i];
    };
    auto full_round = [&]() {
        rec += 64;
        uint32_t n[5], nk;
        fetch(n, nk);                                                // next round's constants arrive during this one
#pragma unroll
        for (int j = 0; j < 5; j++) w[j] = lane_sbox(K, w[j] + c[j]);
        linear();
#pragma unroll
        for (int j = 0; j < 5; j++) c[j] = n[j];
        ck = nk;
#pragma unroll
        for (int j = 0; j < 5; j++) asm volatile("" : "+v"(w[j]));
    };
    // the linear-map constants of K_r, one partial round ahead like the round constants (a per-lane load: lane k = limb k)
    const uint32_t *kl = &klin[0][0][0] + k;
    uint32_t ek[kNL], en[kNL];
#pragma unroll
    for (int i = 0; i < kNL; i++) ek[i] = kl[16 * i];
    auto partial_round = [&](bool last) {
        rec += 64;
        uint32_t n[5], nk;
        fetch(n, nk);
        if (!last) kl += kNL * 16;
#pragma unroll
        for (int i = 0; i < kNL; i++) en[i] = kl[16 * i];
        const uint32_t v5 = lane_sbox(K, w[4] + c[4]);
        w[4] = lane_lin(K, v5, ek, pk, pk1);                         // back to the common scale: v5 K_r / Rp
        linear();
#pragma unroll
        for (int j = 0; j < 5; j++) c[j] = n[j];
        ck = nk;
#pragma unroll
        for (int i = 0; i < kNL; i++) ek[i] = en[i];
#pragma unroll
        for (int j = 0; j < 5; j++) asm volatile("" : "+v"(w[j]));
    };
#pragma unroll 1
    for (int r = 0; r < 4; r++) full_round();
#pragma unroll 1
    for (int r = 4; r < 63; r++) partial_round(r == 62);
#pragma unroll 1
    for (int r = 63; r < 67; r++) full_round();
#pragma unroll
    for (int j = 0; j < 5; j++) L.io[row][j][k] = w[j];
    lanes_fence();
    F29 f;
    {
        const int s_out = lane < 20 ? lane / 5 : 0, w_out = lane < 20 ? lane - 5 * s_out : 0;
#pragma unroll
        for (int i = 0; i < kNL; i++) f.l[i] = (int32_t)L.io[s_out][w_out][i];
    }
    return finalize(mont_mul_const(f, T->final_f));
}

}  // namespace hades
