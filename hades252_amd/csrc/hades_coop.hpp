// hades_coop.hpp -- the low-latency schedule of the permutation: five waves per state.
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
//   constants      copied to LDS once per block and broadcast-read one round ahead
// Critical path per round: one S-box + one row (~1.1 us) instead of 2.2 us (partial) / 5.3 us (full).
// Constants: hades252_amd/_derive.py::coop_schedule (every round compounds the scale s -> s^5/Rp^4/(lam 2^29)).
// tests/test_fast_model.py::coop_perm_model replays this kernel limb for limb with the word bounds asserted.
#pragma once
#include "hades_fast.hpp"

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

// LDS of a cooperative block: the round constants (copied once: a lone block cannot hide 67 cold scalar-cache
// misses behind other waves the way the throughput kernel does -- this is the north-star's "constants broadcast
// from LDS"), the word exchange (ping-pong by round parity) and an AoS staging area for coalesced I/O.
struct CoopLds {
    __attribute__((aligned(16))) int32_t rc[67][64];            // 17 152 B: CoopTables::round
    int32_t xw[2][5][kNL][kWave];                               // 23 040 B
    __attribute__((aligned(16))) uint8_t stage[kWave * 176];    // 64 records, padded like staging.hpp
};

// first thing a cooperative kernel does (followed by a barrier before the first coop_rounds)
__device__ __forceinline__ void coop_load_constants(const CoopTables *T, CoopLds &L) {
    const uint4 *src = reinterpret_cast<const uint4 *>(&T->round[0][0]);
    uint4 *dst = reinterpret_cast<uint4 *>(&L.rc[0][0]);
    for (int i = threadIdx.x; i < 67 * 64 / 4; i += kCoopThreads) dst[i] = src[i];
}

// One output row of small_mds (hades_fast.hpp): st_i <- (sum_j C[i][j] X_j - m p) / 2^29, normalised.
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

__device__ __forceinline__ void coop_put(int32_t (*dst)[kWave], const F29 &v, int lane) {
#pragma unroll
    for (int k = 0; k < kNL; k++) dst[k][lane] = v.l[k];
}
__device__ __forceinline__ F29 coop_get(const int32_t (*src)[kWave], int lane) {
    F29 v;
#pragma unroll
    for (int k = 0; k < kNL; k++) v.l[k] = src[k][lane];
    return v;
}
// this wave's 9 constant limbs of a round: every lane reads the same LDS word (a broadcast read)
__device__ __forceinline__ void coop_constants(const CoopLds &L, int r, int wv, bool full, int32_t (&c)[kNL]) {
    const int off = (full || wv == 4) ? wv * kNL : 5 * kNL;       // A_w, or G for words 0..3 of a partial round
#pragma unroll
    for (int k = 0; k < kNL; k++) c[k] = L.rc[r][off + k];
}

// The 67 rounds on one word per wave.  `mine` = this wave's word of this lane's state (to_f29 of the
// in-memory BlsScalar); returns the final word, still scaled (finalize(mont_mul_const(., final_f)) yields
// the BlsScalar).  wv (the word this wave owns) is wave-uniform.  One block-wide barrier per round: all five
// waves must call it together, after coop_load_constants + a barrier.  A kernel that calls it again (chains of
// permutations) puts a barrier between the calls: round 66 and the next round 0 use the same exchange buffer.
//   own step (S-box, or the G_r product for words 0..3 of a partial round) -> publish -> barrier -> read the
//   other four words -> own row.  The next round's constants are fetched from LDS before the step, so their
//   latency is hidden; ping-pong exchange buffers make one barrier per round sufficient.
// (A two-barrier variant that lets word 4's wave skip the exchange latency was measured 4 % SLOWER: the two
//  waves that share a SIMD become the bottleneck at the earlier barrier -- profiles/r2/coop_probe.txt.)
__device__ __forceinline__ F29 coop_rounds(const CoopTables *T, CoopLds &L, int wv, F29 mine) {
    const int lane = threadIdx.x & (kWave - 1);
    int32_t crow[5];
#pragma unroll
    for (int j = 0; j < 5; j++) crow[j] = T->mds[wv][j];
    int32_t c[kNL], cn[kNL];
    coop_constants(L, 0, wv, true, c);
#pragma unroll 1
    for (int r = 0; r < 67; r++) {
        const bool full = r < 4 || r >= 63;
        if (r + 1 < 67) coop_constants(L, r + 1, wv, r + 1 < 4 || r + 1 >= 63, cn);
        if (full || wv == 4) {
            add_lazy(mine, c);
            mine = sbox29(mine);
        } else {
            F29 g;
#pragma unroll
            for (int k = 0; k < kNL; k++) g.l[k] = c[k];
            mine = mont_mul(mine, g);
        }
        int32_t(*buf)[kNL][kWave] = L.xw[r & 1];
        coop_put(buf[wv], mine, lane);
        __syncthreads();
        F29 x[5];
#pragma unroll
        for (int j = 0; j < 5; j++) x[j] = coop_get(buf[j], lane);
        mine = small_mds_row(crow, x);
#pragma unroll
        for (int k = 0; k < kNL; k++) {
            limb_fence(mine.l[k]);
            c[k] = cn[k];
        }
    }
    return mine;
}

__device__ __forceinline__ Fr coop_finish(const CoopTables *T, const F29 &mine) {
    return finalize(mont_mul_const(mine, T->final_f));
}

}  // namespace hades
