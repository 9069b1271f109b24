// hades_coop.cuh -- the low-latency schedule of the permutation: five waves per state.
//
// k_perm_fast keeps a whole state in one lane: ideal for throughput, but a permutation is then ~89 k
// DEPENDENT VALU instructions of one wave -- 187 us however few states there are (measured: a lone wave
// already issues this code at the pipe's rate, tools/ubench3.hip section 2 vs 3, so neither more waves nor
// instruction-level parallelism can shorten it).  The reference's real call shape is ONE permutation
// (README.md:60-61), and the small levels of a Merkle tree are chains of such calls.
//
// Here one lane still owns one state (same radix-2^29 arithmetic, same bounds, same bits), but its five
// WORDS live on five waves of a 320-thread block:
//   full round     every wave: round key + S-box of its own word (387 multiply-adds instead of 5 x 387)
//   partial round  wave 4: round key + S-box of word 4; waves 0..3, meanwhile, lift their words to the scale
//                  word 4 will have AFTER its S-box (one constant product G_r, off the critical path) --
//                  k_perm_fast instead brings word 4 back DOWN with K_r, a product on the critical path
//   linear layer   words are exchanged through LDS (ping-pong buffers, ONE barrier per round); every wave
//                  computes its own output row of the small-integer MDS (53 multiply-adds instead of 265)
// Critical path per round: one S-box + one row (~1.1 us) instead of 2.2 us (partial) / 5.3 us (full).
// Constants: hades252_amd/_derive.py::coop_schedule (every round compounds the scale s -> s^5/Rp^4/(lam 2^29)).
// tests/test_fast_model.py::coop_perm_model replays this kernel limb for limb with the word bounds asserted.
#pragma once
#include "hades_fast.cuh"

namespace hades {

constexpr int kCoopWaves = 5;
constexpr int kCoopThreads = kCoopWaves * kWave;   // 320
constexpr int kCoopStates = kWave;                 // states per block

// Which word a wave owns.  The five waves of a block land on the CU's four SIMDs round-robin (wave i on SIMD
// i % 4: tools/ubench3.hip "coop" section records HW_ID), so waves 0 and 4 share a SIMD.  Word 4 -- the only
// S-box of a partial round, the critical path -- must have a SIMD to itself; the sharing pair gets two of the
// cheap words (one constant product each in a partial round).  If the hardware placed waves differently the
// kernel would only be slower, never wrong.
__device__ __forceinline__ int coop_word_of_wave(int wave) {
    return wave == 1 ? 4 : (wave == 4 ? 1 : wave);      // waves 0,1,2,3,4 -> words 0,4,2,3,1
}

struct CoopTables {
    int32_t round[67][64];    // per round {A[5][9] (balanced limbs), G[9], pad}
    int32_t final_f[kNL + 7];
    int32_t mds[5][8];        // small-integer MDS rows (wave-uniform row fetch)
};

// LDS of a cooperative block: the word exchange (ping-pong) and an AoS staging area for coalesced I/O
struct CoopLds {
    int32_t xs[2][5][kNL][kWave];                               // 23 040 B
    __attribute__((aligned(16))) uint8_t stage[kWave * 176];    // 64 records, padded like staging.cuh
};

// One output row of small_mds (hades_fast.cuh): st_i <- (sum_j C[i][j] X_j - m p) / 2^29, normalised.
// Same operations in the same order as row i there, hence the same limbs.
__device__ __forceinline__ F29 small_mds_row(const int32_t *crow, const F29 (&x)[5]) {
    F29 r;
    int64_t acc = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) mac(acc, x[j].l[0], crow[j]);
    const int32_t m = (int32_t)((uint32_t)acc & kMask29);
    acc >>= kLB;
#pragma unroll
    for (int k = 1; k < kNL; k++) {
#pragma unroll
        for (int j = 0; j < 5; j++) mac(acc, x[j].l[k], crow[j]);
        mac(acc, m, NEGP29[k]);
        r.l[k - 1] = (int32_t)((uint32_t)acc & kMask29);
        acc >>= kLB;
    }
    r.l[kNL - 1] = (int32_t)acc;
    return r;
}

// The 67 rounds on one word per wave.  `mine` = this wave's word of this lane's state (to_f29 of the
// in-memory BlsScalar); returns the final word, still scaled (finalize(mont_mul_const(., final_f)) yields
// the BlsScalar).  wv is wave-uniform.  Block-wide barriers inside: all five waves must call it together.
__device__ __forceinline__ F29 coop_rounds(const CoopTables *T, CoopLds &L, int wv, F29 mine) {
    const int lane = threadIdx.x & (kWave - 1);
    int32_t crow[5];
#pragma unroll
    for (int j = 0; j < 5; j++) crow[j] = T->mds[wv][j];
#pragma unroll 1
    for (int r = 0; r < 67; r++) {
        const int32_t *rec = T->round[r];
        const bool full = r < 4 || r >= 63;
        if (full || wv == 4) {
            add_lazy(mine, rec + wv * kNL);
            mine = sbox29(mine);
        } else {
            mine = mont_mul_const(mine, rec + 5 * kNL);
        }
        int32_t(*buf)[kNL][kWave] = L.xs[r & 1];
#pragma unroll
        for (int k = 0; k < kNL; k++) buf[wv][k][lane] = mine.l[k];
        __syncthreads();
        F29 x[5];
#pragma unroll
        for (int j = 0; j < 5; j++)
#pragma unroll
            for (int k = 0; k < kNL; k++) x[j].l[k] = buf[j][k][lane];
        mine = small_mds_row(crow, x);
#pragma unroll
        for (int k = 0; k < kNL; k++) limb_fence(mine.l[k]);
    }
    return mine;
}

__device__ __forceinline__ Fr coop_finish(const CoopTables *T, const F29 &mine) {
    return finalize(mont_mul_const(mine, T->final_f));
}

}  // namespace hades
