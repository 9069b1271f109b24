// hades_literal.hpp -- the permutation exactly as the reference schedules it.
//
// One permutation per lane, state (5 x 8 u32 = 40 VGPRs) in registers across all 67 rounds.
// Every step mirrors the reference one to one, on Montgomery-form values:
//   perm                 src/strategies.rs:140-157   4 full + 59 partial + 4 full, one cursor
//   apply_full_round     src/strategies.rs:107-119
//   apply_partial_round  src/strategies.rs:79-93     ARK on all words, S-box on the LAST word
//   add_round_key        src/strategies/scalar.rs:23-30
//   quintic_s_box        src/strategies/scalar.rs:32-34   (v^2)^2 * v
//   mul_matrix           src/strategies/scalar.rs:36-49   dense 5x5, result[k] += M[k][j]*v[j]
// 25 + 3 (or 15) full Montgomery products per round.  This is the parity anchor for the
// scale-tracked fast kernel (hades_fast.hpp), which computes the same field elements with
// ~4x fewer multiplies.
#pragma once
#include "fr32.hpp"

namespace hades {

// Constant tables live in device memory; every access below uses a wave-uniform index, so hipcc
// emits scalar loads (s_load_dwordx8 through the scalar data cache) and the constants arrive in
// SGPRs -- a wave-wide broadcast that costs no VGPRs, no LDS bandwidth and no VALU issue.
struct LiteralView {
    const uint32_t (*ark)[8];   // ROUND_CONSTANTS[0..960), Montgomery form
    const uint32_t (*mds)[8];   // MDS_MATRIX row-major, Montgomery form
};

__device__ __forceinline__ Fr load_const(const uint32_t (*tab)[8], int idx) {
    Fr c;
#pragma unroll
    for (int i = 0; i < 8; i++) c.l[i] = tab[idx][i];
    return c;
}

// `cursor` = position of the constants iterator (src/strategies.rs:33-41); perm() uses 5 * round
__device__ __forceinline__ void lit_add_round_key(const LiteralView &T, int cursor, Fr (&st)[5]) {
#pragma unroll
    for (int w = 0; w < 5; w++) st[w] = fr_add(st[w], load_const(T.ark, cursor + w));
}

__device__ __forceinline__ Fr lit_quintic_s_box(const Fr &v) {
    Fr v2 = fr_mul_call(v, v);
    Fr v4 = fr_mul_call(v2, v2);
    return fr_mul_call(v4, v);
}

__device__ __forceinline__ void lit_mul_matrix(const LiteralView &T, Fr (&st)[5]) {
    Fr res[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        res[k] = fr_mul_call(load_const(T.mds, 5 * k + 0), st[0]);
#pragma unroll
        for (int j = 1; j < 5; j++) res[k] = fr_add(res[k], fr_mul_call(load_const(T.mds, 5 * k + j), st[j]));
    }
#pragma unroll
    for (int k = 0; k < 5; k++) st[k] = res[k];
}

__device__ __forceinline__ void lit_full_round(const LiteralView &T, int cursor, Fr (&st)[5]) {
    lit_add_round_key(T, cursor, st);
#pragma unroll
    for (int w = 0; w < 5; w++) st[w] = lit_quintic_s_box(st[w]);
    lit_mul_matrix(T, st);
}

__device__ __forceinline__ void lit_partial_round(const LiteralView &T, int cursor, Fr (&st)[5]) {
    lit_add_round_key(T, cursor, st);
    st[4] = lit_quintic_s_box(st[4]);
    lit_mul_matrix(T, st);
}

__device__ __forceinline__ void lit_perm(const LiteralView &T, Fr (&st)[5]) {
    int cursor = 0;
#pragma unroll 1
    for (int i = 0; i < 4; i++, cursor += 5) lit_full_round(T, cursor, st);
#pragma unroll 1
    for (int i = 0; i < 59; i++, cursor += 5) lit_partial_round(T, cursor, st);
#pragma unroll 1
    for (int i = 0; i < 4; i++, cursor += 5) lit_full_round(T, cursor, st);
}

}  // namespace hades
