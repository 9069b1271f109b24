// hades252.hip -- kernels + C ABI of libhades252 (gfx950 only).  See include/hades252.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <thread>
#include <vector>

#include "../../include/hades252.h"
#include "fr32.hpp"
#include "hades_constants.inc"
#include "hades_literal.hpp"
#include "staging.hpp"
#include "hades_fast.hpp"
#include "k_perm_fast.hpp"
#include "hades_coop.hpp"
#include "hades_lanes.hpp"

using namespace hades;

// ------------------------------------------------------------------------------------------
// constant tables (code-object globals: one copy per device, loaded with the module)
// ------------------------------------------------------------------------------------------
// all 960 ROUND_CONSTANTS (src/round_constants.rs:18): perm() consumes the first 335, the trait's
// add_round_key / apply_*_round accept any cursor (src/strategies.rs:33-41)
__device__ const uint32_t d_ark_mont[HADES_N_ARK][8] = HADES_ARK_MONT_INIT;
__device__ const uint32_t d_mds_mont[25][8] = HADES_MDS_MONT_INIT;
// R^2 mod p (from_raw / from_bytes multiplier) and 1 (to_bytes multiplier), 8 x u32
__device__ const uint32_t d_r2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                     0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};

// d_fast (the throughput kernel's round records) is defined next to its kernel in k_perm_fast.hpp
// low-latency schedule (hades_coop.hpp)
__device__ const CoopTables d_coop = {HADES_COOP_ROUND_INIT, HADES_COOP_FINAL_F, HADES_FAST_MDS_SMALL};
// lane-split schedule (hades_lanes.hpp): the coop schedule with plain-limb round constants + the reduction constants
__device__ const LanesTables d_lanes = {HADES_LANES_ROUND_INIT, HADES_COOP_FINAL_F, HADES_FAST_MDS_SMALL, HADES_P29,
                                        HADES_P29, HADES_NEG_PINV29};
// trace kernel: U_r with mont(X_after_round_r, U_r) = x * 2^256
__device__ const int32_t d_trace_u[67][16] = HADES_FAST_TRACE_U_INIT;
// ... + D_r: the partial-round constants of words 0..3 that the shipped schedule defers (hades_fast.hpp item 5)
__device__ const uint32_t d_trace_d[67][5][8] = HADES_FAST_TRACE_D_INIT;
// witness kernel: un-scaling factors {u_in,u2,u4,u5,w1,u_post} and additive corrections {d1[5], d2[5]} per round
struct WitnessTables {
    int32_t u[67][64];
    uint32_t d[67][10][8];
};
__device__ const WitnessTables d_wit = {HADES_WITNESS_U_INIT, HADES_WITNESS_D_INIT};
// generic radix-2^29 field ops (hades252_fr_op_dev)
__device__ const int32_t d_rp_mod_p[16] = HADES_RP_MOD_P29;
// per-operation kernels on the same path (hades252_amd/_derive.py)
__device__ const int32_t d_op_k[16] = HADES_OP_K29;
__device__ const int32_t d_op_w[16] = HADES_OP_W29;
__device__ const int32_t d_op_w_full[16] = HADES_OP_W_FULL29;
// wire format (from_bytes / to_bytes) on the same path
__device__ const int32_t d_rp_times_r[16] = HADES_RP_TIMES_R29;
__device__ const int32_t d_rp_over_r[16] = HADES_RP_OVER_R29;
__device__ const int32_t d_rp2_over_r[16] = HADES_RP2_OVER_R29;

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ Fr load_word(const uint8_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 lo = q[0], hi = q[1];
    Fr w;
    w.l[0] = lo.x; w.l[1] = lo.y; w.l[2] = lo.z; w.l[3] = lo.w;
    w.l[4] = hi.x; w.l[5] = hi.y; w.l[6] = hi.z; w.l[7] = hi.w;
    return w;
}
__device__ __forceinline__ void store_word(uint8_t *p, const Fr &w) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(w.l[0], w.l[1], w.l[2], w.l[3]);
    q[1] = make_uint4(w.l[4], w.l[5], w.l[6], w.l[7]);
}
__device__ __forceinline__ Fr zero_word() {
    Fr w;
#pragma unroll
    for (int i = 0; i < 8; i++) w.l[i] = 0;
    return w;
}

enum Op { OP_PERM = 0, OP_ARK, OP_MDS, OP_FULL, OP_PARTIAL };

// st[4] <- st[3] <- ... <- st[0] <- st[4]: loops over the five words rotate the state through ONE code body
__device__ __forceinline__ void rotate_right(F29 (&st)[5]) {
    const F29 t = st[4];
    st[4] = st[3];
    st[3] = st[2];
    st[2] = st[1];
    st[1] = st[0];
    st[0] = t;
}

template <int OP>
__global__ void __launch_bounds__(kBlock) k_states_literal(uint8_t *states, size_t n, int cursor) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr st[5];
    wave_load_records<5>(states, rec0, n, slab, st);
    LiteralView V{d_ark_mont, d_mds_mont};
    if constexpr (OP == OP_PERM) lit_perm(V, st);
    if constexpr (OP == OP_ARK) lit_add_round_key(V, cursor, st);
    if constexpr (OP == OP_MDS) lit_mul_matrix(V, st);
    if constexpr (OP == OP_FULL) lit_full_round(V, cursor, st);
    if constexpr (OP == OP_PARTIAL) lit_partial_round(V, cursor, st);
    wave_store_records<5>(states, rec0, n, slab, st);
}

// Per-round trace: the state after every round (what the PLONK gadget needs as witnesses,
// reference src/strategies/gadget.rs:41-133), round-major: trace[r] is a whole AoS batch.
// Literal variant (the reference's schedule; parity anchor for the fast one).
__global__ void __launch_bounds__(kBlock) k_perm_trace_literal(const uint8_t *__restrict__ states,
                                                               uint8_t *__restrict__ trace, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr st[5];
    wave_load_records<5>(states, rec0, n, slab, st);
    LiteralView V{d_ark_mont, d_mds_mont};
#pragma unroll 1
    for (int r = 0; r < 67; r++) {
        if (r < 4 || r >= 63)
            lit_full_round(V, 5 * r, st);
        else
            lit_partial_round(V, 5 * r, st);
        wave_store_records<5>(trace + (size_t)r * n * 160, rec0, n, slab, st);
    }
}

// Scale-tracked trace (the shipped one): the rounds of k_perm_fast; after each round every word is
// brought back to the in-memory BlsScalar with ONE constant product (U_r = 2^256 * Rp / s_{r+1},
// hades252_amd/_derive.py), a full reduction, and -- in partial rounds, whose word 0..3 constants the
// schedule defers -- one field addition of the known offset D_r: 5 extra products per round instead of
// the literal schedule's 28 / 40 full-width products.
__global__ void __launch_bounds__(kBlock, 3) k_perm_trace_fast(const uint8_t *__restrict__ states,
                                                               uint8_t *__restrict__ trace, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    F29 st[5];
    {
        Fr in[5];
        wave_load_records<5>(states, rec0, n, slab, in);
#pragma unroll
        for (int w = 0; w < 5; w++) st[w] = to_f29(in[w]);
    }
#pragma unroll 1
    for (int r = 0; r < 67; r++) {
        fast_round(d_fast.round[r], r < 4 || r >= 63, st);
        const int32_t *u = d_trace_u[r];
#pragma unroll
        for (int w = 0; w < 5; w++) {
            Fr v = finalize(mont_mul_const(st[w], u));
            if (r >= 4 && r < 63) v = fr_add(v, load_const(d_trace_d[r], w));
            slab_put<5>(slab, w, v);
        }
        slab_flush<5>(trace + (size_t)r * n * 160, rec0, n, slab);
    }
}

// Full gadget witness: EVERY gate output of the reference's GadgetStrategy for every state -- the 972 values a
// PLONK prover assigns per permutation (src/strategies/gadget.rs:41-133: round-0 key additions, v^2 / v^4 / v^5 of
// each S-box, and per linear layer the 3-term partial sums r1[j] and the rows r2[j] with the NEXT round's constant
// appended).  Wire-major output: wires[g] is a batch of n scalars (32 B, in-memory BlsScalar), g in gate order.
// The rounds are those of k_perm_fast; each value is un-scaled with one constant product, fully reduced and, where
// the shipped schedule defers constants, corrected by a known offset (hades252_amd/_derive.py::witness_schedule;
// limb-exact replay: tests/test_fast_model.py::witness_model).  Loops over words rotate the state so that every
// piece of code exists once (I-cache).
__device__ __forceinline__ void store_wire(uint8_t *wires, size_t n, int wire, size_t rec, bool live, const Fr &v) {
    if (live) {
        uint4 *q = reinterpret_cast<uint4 *>(wires + ((size_t)wire * n + rec) * 32);
        q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
        q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    }
}

__global__ void __launch_bounds__(kBlock, 3) k_perm_witness(const uint8_t *__restrict__ states,
                                                            uint8_t *__restrict__ wires, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    const size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    const size_t rec = rec0 + (threadIdx.x & (kWave - 1));
    const bool live = rec < n;
    F29 st[5];
    {
        Fr in[5];
        wave_load_records<5>(states, rec0, n, slab, in);
#pragma unroll
        for (int w = 0; w < 5; w++) st[w] = to_f29(in[w]);
    }
    int wire = 0;
#pragma unroll 1
    for (int r = 0; r < 67; r++) {
        const int32_t *rc = d_fast.round[r];
        const int32_t *u = d_wit.u[r];
        const bool full = r < 4 || r >= 63;
        if (full) {
#pragma unroll
            for (int w = 0; w < 4; w++) add_lazy(st[w], rc + w * kNL);
        }
        add_lazy(st[4], rc + 4 * kNL);
        if (r == 0) {
#pragma unroll 1
            for (int i = 0; i < 5; i++) {               // state after the first round key: word 4 - i sits at st[4]
                store_wire(wires, n, wire + 4 - i, rec, live, finalize(mont_mul_const(st[4], u)));
                rotate_right(st);
            }
            wire += 5;
        }
        // S-boxes: v^2, v^4, v^5 (partial round: word 4 only, then the K_r product that re-scales it)
        const int cnt = full ? 5 : 1;
#pragma unroll 1
        for (int i = 0; i < cnt; i++) {
            const int w = full ? 4 - i : 0;             // gate order: word 0 first (a partial round has one S-box)
            const F29 v2 = mont_sqr(st[4]);
            store_wire(wires, n, wire + 3 * w, rec, live, finalize(mont_mul_const(v2, u + kNL)));
            const F29 v4 = mont_sqr(v2);
            store_wire(wires, n, wire + 3 * w + 1, rec, live, finalize(mont_mul_const(v4, u + 2 * kNL)));
            F29 v5 = mont_mul(v4, st[4]);
            if (!full) v5 = mont_mul_const(v5, rc + 5 * kNL);
            store_wire(wires, n, wire + 3 * w + 2, rec, live, finalize(mont_mul_const(v5, u + 3 * kNL)));
            st[4] = v5;
            if (full) rotate_right(st);
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(st[4].l[k]);
        }
        wire += 3 * cnt;
        // r1[j] = M[j][0] z0 + M[j][1] z1 + M[j][2] z2: three columns of the small-integer layer, one-limb step
#pragma unroll 1
        for (int j = 0; j < 5; j++) {
            const int32_t c0 = d_coop.mds[j][0], c1 = d_coop.mds[j][1], c2 = d_coop.mds[j][2];
            F29 y;
            int64_t acc = 0;
            mac(acc, st[0].l[0], c0);
            mac(acc, st[1].l[0], c1);
            mac(acc, st[2].l[0], c2);
            const int32_t m = (int32_t)((uint32_t)acc & kMask29);
            acc >>= kLB;
#pragma unroll
            for (int k = 1; k < kNL; k++) {
                mac(acc, st[0].l[k], c0);
                mac(acc, st[1].l[k], c1);
                mac(acc, st[2].l[k], c2);
                mac(acc, m, NEGP29[k]);
                y.l[k - 1] = (int32_t)((uint32_t)acc & kMask29);
                acc >>= kLB;
            }
            y.l[kNL - 1] = (int32_t)acc;
            Fr v = finalize(mont_mul_const(y, u + 4 * kNL));
            if (!full) v = fr_add(v, load_const(d_wit.d[r], j));
            store_wire(wires, n, wire + 2 * j, rec, live, v);
        }
        small_mds(st);
        // r2[j] = row j of the linear layer + the next round's constant
#pragma unroll 1
        for (int i = 0; i < 5; i++) {
            const int j = 4 - i;
            Fr v = finalize(mont_mul_const(st[4], u + 5 * kNL));
            v = fr_add(v, load_const(d_wit.d[r], 5 + j));
            store_wire(wires, n, wire + 2 * j + 1, rec, live, v);
            rotate_right(st);
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(st[4].l[k]);
        }
        wire += 10;
#pragma unroll
        for (int w = 0; w < 5; w++)
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(st[w].l[k]);
    }
}

// Generic batched BlsScalar operations (reference call sites src/strategies/scalar.rs:28,33,44;
// src/round_constants.rs:41): out[i] = a[i] (op) b[i] on Montgomery limbs, fully reduced.
// IMPL 0: the saturated 8 x u32 CIOS arithmetic of fr32.hpp (what the literal kernels use);
// IMPL 1: the radix-2^29 signed-limb arithmetic of the shipped kernel (to_f29, mont_fips, finalize).
// These exist so that tests can drive BOTH device arithmetics through the computations that produced
// the reference's constant blobs (tests/test_gpu_blob_kat.py), and as a13's batched surface.
enum FrOp { FR_ADD = 0, FR_MUL = 1, FR_SQUARE = 2, FR_FROM_RAW = 3 };
template <int IMPL>
__global__ void __launch_bounds__(kBlock) k_fr_op(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n, int op) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<1>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr x[1], y[1];
    wave_load_records<1>(a, rec0, n, slab, x);
    if (op == FR_ADD || op == FR_MUL) {
        wave_load_records<1>(b, rec0, n, slab, y);
    } else if (op == FR_SQUARE) {
        y[0] = x[0];
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) y[0].l[i] = d_r2[i];
    }
    Fr r[1];
    if constexpr (IMPL == 0) {
        r[0] = (op == FR_ADD) ? fr_add(x[0], y[0]) : fr_mul(x[0], y[0]);
    } else {
        F29 xa = to_f29(x[0]), yb = to_f29(y[0]);
        if (op == FR_ADD) {
            add_lazy(xa, yb.l);                                    // limbs < 2^30
            r[0] = finalize(mont_mul_const(xa, d_rp_mod_p));       // (a + b) * Rp / Rp
        } else {
            F29 t = (op == FR_SQUARE) ? mont_sqr(xa) : mont_mul(xa, yb);   // a b / Rp
            r[0] = finalize(mont_mul_const(t, d_rp2_over_r));      // * (Rp^2 / 2^256) / Rp = a b / 2^256
        }
    }
    wave_store_records<1>(out, rec0, n, slab, r);
}

// The trait's per-operation methods on the radix-2^29 path: same field elements as the literal forms above (kept
// for add_round_key, which is five additions), a sixth to a tenth of the instructions -- mul_matrix is the
// small-integer layer + one un-scaling product per word instead of 25 full products.  Round keys are added in the
// memory format first (any cursor over all 960 constants), so the result of every method is the unique reduced
// BlsScalar, bit-identical to the literal kernels and the oracle.
template <int OP>
__global__ void __launch_bounds__(kBlock, 3) k_states_fast(uint8_t *states, size_t n, int cursor) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr in[5];
    wave_load_records<5>(states, rec0, n, slab, in);
    if constexpr (OP != OP_MDS) {
        LiteralView V{d_ark_mont, d_mds_mont};
        lit_add_round_key(V, cursor, in);
    }
    F29 st[5];
#pragma unroll
    for (int w = 0; w < 5; w++) st[w] = to_f29(in[w]);
    if constexpr (OP == OP_FULL) {
#pragma unroll 1
        for (int i = 0; i < 5; i++) {                 // one S-box body, the state rotates through it
            st[4] = sbox29(st[4]);
            rotate_right(st);
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(st[4].l[k]);
        }
    }
    if constexpr (OP == OP_PARTIAL) st[4] = mont_mul_const(sbox29(st[4]), d_op_k);
    small_mds(st);
    const int32_t *u = OP == OP_FULL ? d_op_w_full : d_op_w;
    Fr out[5];
#pragma unroll
    for (int w = 0; w < 5; w++) out[w] = finalize(mont_mul_const(st[w], u));
    wave_store_records<5>(states, rec0, n, slab, out);
}

__global__ void __launch_bounds__(kBlock) k_sbox(uint8_t *scalars, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<1>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr st[1];
    wave_load_records<1>(scalars, rec0, n, slab, st);
    st[0] = finalize(mont_mul_const(sbox29(to_f29(st[0])), d_op_k));
    wave_store_records<1>(scalars, rec0, n, slab, st);
}

// canonical bytes <-> Montgomery limbs (BlsScalar::from_bytes / to_bytes): 64 B of HBM traffic and ONE constant
// product per scalar.  The product runs on the radix-2^29 path (mont_mul_const: 153 multiply-adds; the saturated
// 8x32 product these kernels used in round 1 is ~620 instructions and made them VALU-bound at 3.4-4.5 TB/s).
// No LDS: every lane reads and writes its own 32 bytes with two 16-byte accesses -- a wave's two instructions
// together cover 2 KiB contiguous, the second hits the lines the first fetched -- and takes kWirePerThread scalars
// in a grid-stride loop to keep more bytes in flight.  `out` may be `in` (lane-private in-place update).
constexpr int kWirePerThread = 4;
constexpr int32_t kRpOverR = 1 << (kLB * kNL - 256);        // 2^261 / 2^256
template <int MODE>   // 0 = to_bytes (x / 2^256), 1 = from_bytes (a * 2^256, inputs >= p rejected)
__global__ void __launch_bounds__(kBlock) k_wire(const uint8_t *in, uint8_t *out, size_t n, int *bad_count) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    const int32_t *factor = MODE == 1 ? d_rp_times_r : d_rp_over_r;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + i * 32);
        uint4 lo = p[0], hi = p[1];
        Fr a;
        a.l[0] = lo.x; a.l[1] = lo.y; a.l[2] = lo.z; a.l[3] = lo.w;
        a.l[4] = hi.x; a.l[5] = hi.y; a.l[6] = hi.z; a.l[7] = hi.w;
        // to_bytes: the factor Rp / 2^256 = 32 is a single limb: 81 multiply-adds instead of 153 (5.0 -> 5.3 TB/s at 2^26
        // scalars; issuing the next scalar's loads before this one's arithmetic changed nothing: profiles/r3/wire_bw.txt)
        Fr m = finalize(MODE == 1 ? mont_mul_const(to_f29(a), factor) : mont_mul_small(to_f29(a), kRpOverR));
        if (MODE == 1 && !fr_is_canonical(a)) {
#pragma unroll
            for (int k = 0; k < 8; k++) m.l[k] = 0;
            if (bad_count != nullptr) atomicAdd(bad_count, 1);
        }
        uint4 *q = reinterpret_cast<uint4 *>(out + i * 32);
        q[0] = make_uint4(m.l[0], m.l[1], m.l[2], m.l[3]);
        q[1] = make_uint4(m.l[4], m.l[5], m.l[6], m.l[7]);
    }
}

// One Merkle level, one parent per lane: parent = perm([tag, c_0 .. c_{ARITY-1}, 0 ..])[out_idx], ARITY = 1 .. 4
// (arity 4 fills the state: the caller shape of dusk-poseidon, README.md:9; smaller arities leave zero words).
// The level may be ragged: n_children need not be a multiple of ARITY; a child position past the end of the level takes
// the digest at `pad` (device memory, 32 B; NULL = the zero scalar) -- the "empty subtree" digest of that level.
__device__ __forceinline__ Fr load_pad(const uint8_t *pad) { return pad != nullptr ? load_word(pad) : zero_word(); }

template <int ARITY>
__global__ void __launch_bounds__(kBlock, 4) k_merkle_level_fast(const uint8_t *__restrict__ children, size_t n_children,
                                                                 uint8_t *__restrict__ parents, size_t n_parents,
                                                                 Fr tag, int out_idx, const uint8_t *__restrict__ pad) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<ARITY>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr ch[ARITY];
    wave_load_scalars<ARITY>(children, rec0, n_children, slab, ch);
    const size_t first = (rec0 + (threadIdx.x & (kWave - 1))) * ARITY;
    if (first + ARITY > n_children) {                      // at most one lane of the grid with live data gets here
        const Fr pd = load_pad(pad);
#pragma unroll
        for (int w = 0; w < ARITY; w++)
            if (first + w >= n_children) ch[w] = pd;
    }
    Fr st[5];
    st[0] = tag;
#pragma unroll
    for (int w = 1; w < 5; w++) st[w] = w <= ARITY ? ch[w <= ARITY ? w - 1 : 0] : zero_word();
    Fr out[1];
    fast_perm<1>(&d_fast, st, out, out_idx);
    wave_store_records<1>(parents, rec0, n_parents, slab, out);
}

// Path verification: lane q recomputes the root from leaf q and its opening (the siblings of hades252_merkle_open_dev:
// level l, child order, own position (index / ARITY^l) % ARITY skipped) -- `depth` dependent permutations per lane.
template <int ARITY>
__global__ void __launch_bounds__(kBlock, 3) k_merkle_verify(const uint8_t *__restrict__ leaves,
                                                             const uint64_t *__restrict__ indices,
                                                             const uint8_t *__restrict__ paths, size_t n_queries, int depth,
                                                             Fr tag, int out_idx, uint8_t *__restrict__ roots) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<1>(lds);
    const size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    const size_t q = rec0 + (threadIdx.x & (kWave - 1));
    const bool live = q < n_queries;
    Fr node[1];
    wave_load_records<1>(leaves, rec0, n_queries, slab, node);
    uint64_t idx = live ? indices[q] : 0;
    const uint8_t *mine = paths + q * (size_t)depth * (ARITY - 1) * 32;
#pragma unroll 1
    for (int l = 0; l < depth; l++) {
        const int pos = (int)(idx % ARITY);
        idx /= ARITY;
        Fr st[5];
        st[0] = tag;
#pragma unroll
        for (int w = 1; w < 5; w++) st[w] = zero_word();
#pragma unroll
        for (int c = 0; c < ARITY; c++) {                 // child c: the node itself at `pos`, else the next sibling
            Fr v = node[0];
            if (c != pos && live) v = load_word(mine + ((size_t)l * (ARITY - 1) + (c < pos ? c : c - 1)) * 32);
            st[1 + c] = v;
        }
        fast_perm<1>(&d_fast, st, node, out_idx);
    }
    wave_store_records<1>(roots, rec0, n_queries, slab, node);
}

// ---- low-latency kernels: five waves per state (hades_coop.hpp) --------------------------------------
// In-place permutation of up to 64 states per 320-thread block.
__global__ void __launch_bounds__(kCoopThreads) k_perm_coop(uint8_t *states, size_t n) {
    __shared__ CoopLds L;
    const int wv = coop_word_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));   // the word this wave owns
    const int lane = threadIdx.x & (kWave - 1);
    const size_t rec0 = (size_t)blockIdx.x * kCoopStates;
    const size_t total = n * 10, chunk0 = rec0 * 10;
    uint4 *g = reinterpret_cast<uint4 *>(states + rec0 * 160);
    coop_load_constants(&d_coop, L);
#pragma unroll
    for (int c = threadIdx.x; c < kCoopStates * 10; c += kCoopThreads) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (chunk0 + c < total) v = g[c];
        const int rec = c / 10, part = c - rec * 10;
        *reinterpret_cast<uint4 *>(L.stage + rec * 176 + part * 16) = v;
    }
    __syncthreads();
    Fr w;
    {
        const uint4 *p = reinterpret_cast<const uint4 *>(L.stage + lane * 176 + wv * 32);
        uint4 lo = p[0], hi = p[1];
        w.l[0] = lo.x; w.l[1] = lo.y; w.l[2] = lo.z; w.l[3] = lo.w;
        w.l[4] = hi.x; w.l[5] = hi.y; w.l[6] = hi.z; w.l[7] = hi.w;
    }
    const F29 fin = coop_rounds(&d_coop, L, wv, to_f29(w));     // 67 barriers: everyone has read `stage` by now
    const Fr o = coop_finish(&d_coop, fin);
    {
        uint4 *p = reinterpret_cast<uint4 *>(L.stage + lane * 176 + wv * 32);
        p[0] = make_uint4(o.l[0], o.l[1], o.l[2], o.l[3]);
        p[1] = make_uint4(o.l[4], o.l[5], o.l[6], o.l[7]);
    }
    __syncthreads();
#pragma unroll
    for (int c = threadIdx.x; c < kCoopStates * 10; c += kCoopThreads) {
        const int rec = c / 10, part = c - rec * 10;
        uint4 v = *reinterpret_cast<const uint4 *>(L.stage + rec * 176 + part * 16);
        if (chunk0 + c < total) g[c] = v;
    }
}

// ---- lowest-latency kernels: one state per WAVE, every field element spread over a 16-lane row (hades_lanes.hpp) ---
// Two forms, four waves per block (one per SIMD) either way:
//   HELPED   three states per block + a helper wave that owns word 3 of all three during the full rounds (its S-box then
//            runs beside the main waves' instead of doubling their instruction stream): 50 us -- up to 768 states, one
//            block per CU;
//   plain    four states per block, every wave does everything itself: 54 us -- for 769 .. 1 024 states, where the helped
//            form would put a second block on some CUs.
constexpr int kLanesWaves = 4;
struct LanesAlways {
    __device__ __forceinline__ bool operator()(size_t) const { return true; }
};
// `wanted(rec)` (wave-uniform) lets a kernel drop records it does not need; such a wave idles like one past the end
template <bool HELPED, class Wanted = LanesAlways>
__device__ __forceinline__ bool lanes_role(LanesLds *L, size_t n, size_t &rec, Wanted wanted = Wanted()) {
    const int wave = threadIdx.x >> 6;                                                   // false: this wave is done
    if constexpr (HELPED) {
        if (wave == kLanesWaves - 1) {
            lanes_helper<kLanesWaves - 1>(&d_lanes, *reinterpret_cast<LanesLds(*)[kLanesWaves - 1]>(L));
            return false;
        }
        rec = (size_t)blockIdx.x * (kLanesWaves - 1) + wave;
        if (rec >= n || !wanted(rec)) {
            lanes_idle();
            return false;
        }
        return true;
    } else {
        rec = (size_t)blockIdx.x * kLanesWaves + wave;
        return rec < n && wanted(rec);                               // no block-wide barrier anywhere: idle waves leave
    }
}

// In-place permutation, one state per wave.
template <bool HELPED>
__global__ void __launch_bounds__(kLanesWaves *kWave) k_perm_lanes(uint8_t *states, size_t n) {
    __shared__ LanesLds L[kLanesWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    size_t rec;
    if (!lanes_role<HELPED>(L, n, rec)) return;
    uint8_t *mine = states + rec * 160 + (lane < 5 ? lane : 0) * 32;
    const Fr in = lane < 5 ? load_word(mine) : zero_word();
    const Fr out = lanes_perm<HELPED>(&d_lanes, L[wave], in);
    if (lane < 5) store_word(mine, out);
}

// One Merkle level, one parent per wave: parent = perm([tag, c_0 .. c_{ARITY-1}, 0 ..])[out_idx]; ragged levels and
// `pad` as in k_merkle_level_fast.
template <int ARITY, bool HELPED>
__global__ void __launch_bounds__(kLanesWaves *kWave) k_merkle_lanes(const uint8_t *__restrict__ children, size_t n_children,
                                                                     uint8_t *__restrict__ parents, size_t n_parents,
                                                                     Fr tag, int out_idx, const uint8_t *__restrict__ pad) {
    __shared__ LanesLds L[kLanesWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    size_t rec;
    if (!lanes_role<HELPED>(L, n_parents, rec)) return;
    Fr in = zero_word();
    if (lane == 0) in = tag;
    if (lane >= 1 && lane <= ARITY) {
        const size_t c = rec * ARITY + (lane - 1);
        in = c < n_children ? load_word(children + c * 32) : load_pad(pad);
    }
    const Fr out = lanes_perm<HELPED>(&d_lanes, L[wave], in);
    if (lane == out_idx) store_word(parents + rec * 32, out);
}

// Incremental update, one level: query q names a changed LEAF indices[q]; its ancestor on this level is parent
// p = indices[q] / span (span = ARITY^(level+1)), recomputed from the level below (already up to date) and written in
// place.  A query whose predecessor has the same ancestor leaves it to the predecessor (sorted index lists do each
// ancestor once; unsorted ones may repeat work, never miss any: the first query of every run computes it; concurrent
// writers of one parent write identical bytes).  Leaf indices >= n_leaves are ignored.
struct UpdateWanted {
    const uint64_t *indices;
    size_t n_leaves;
    uint64_t span;
    __device__ __forceinline__ bool operator()(size_t q) const {
        const uint64_t i = indices[q];
        if (i >= n_leaves) return false;
        if (q == 0) return true;
        const uint64_t j = indices[q - 1];
        return j >= n_leaves || j / span != i / span;
    }
};

template <int ARITY>
__device__ __forceinline__ Fr update_child(const uint8_t *__restrict__ children, size_t n_children, size_t parent, int w,
                                           const uint8_t *__restrict__ pad) {
    const size_t c = parent * ARITY + w;
    return c < n_children ? load_word(children + c * 32) : load_pad(pad);
}

template <int ARITY>
__global__ void __launch_bounds__(kBlock, 4) k_merkle_update_fast(const uint8_t *__restrict__ children, size_t n_children,
                                                                  uint8_t *__restrict__ parents,
                                                                  const uint64_t *__restrict__ indices, size_t n_updates,
                                                                  size_t n_leaves, uint64_t span, Fr tag, int out_idx,
                                                                  const uint8_t *__restrict__ pad) {
    const size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const UpdateWanted wanted{indices, n_leaves, span};
    if (q >= n_updates || !wanted(q)) return;
    const size_t parent = indices[q] / span;
    Fr st[5];
    st[0] = tag;
#pragma unroll
    for (int w = 1; w < 5; w++) st[w] = w <= ARITY ? update_child<ARITY>(children, n_children, parent, w - 1, pad) : zero_word();
    Fr out[1];
    fast_perm<1>(&d_fast, st, out, out_idx);
    store_word(parents + parent * 32, out[0]);
}

template <int ARITY, bool HELPED>
__global__ void __launch_bounds__(kLanesWaves *kWave) k_merkle_update_lanes(const uint8_t *__restrict__ children,
                                                                            size_t n_children, uint8_t *__restrict__ parents,
                                                                            const uint64_t *__restrict__ indices,
                                                                            size_t n_updates, size_t n_leaves, uint64_t span,
                                                                            Fr tag, int out_idx,
                                                                            const uint8_t *__restrict__ pad) {
    __shared__ LanesLds L[kLanesWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    size_t q;
    if (!lanes_role<HELPED>(L, n_updates, q, UpdateWanted{indices, n_leaves, span})) return;
    const size_t parent = indices[q] / span;
    Fr in = zero_word();
    if (lane == 0) in = tag;
    if (lane >= 1 && lane <= ARITY) in = update_child<ARITY>(children, n_children, parent, lane - 1, pad);
    const Fr out = lanes_perm<HELPED>(&d_lanes, L[wave], in);
    if (lane == out_idx) store_word(parents + parent * 32, out);
}

// Fused Merkle levels: block b takes the children of parents [64b, 64b + 64) of one level (n_parents in
// all) and runs `n_levels` tree levels without leaving the CU: level j has 64 / ARITY^j parents per block,
// its digests become the next level's children through LDS.  The caller guarantees that the block's parent
// count is divisible by ARITY^(n_levels-1) (trees with a power-of-ARITY leaf count are).
//   out_all  (may be NULL) receives EVERY level: level j (n_parents / ARITY^j digests of 32 B) at byte offset
//            32 * sum_{i<j} n_parents / ARITY^i  -- the layout of hades252_merkle_build_dev;
//   out_last (may be NULL) receives the last level run: n_parents / ARITY^(n_levels-1) digests.
template <int ARITY>
__global__ void __launch_bounds__(kCoopThreads) k_merkle_coop(const uint8_t *__restrict__ children,
                                                             uint8_t *__restrict__ out_all,
                                                             uint8_t *__restrict__ out_last, size_t n_parents, Fr tag,
                                                             int out_idx, int n_levels) {
    __shared__ CoopLds L;
    const int wv = coop_word_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));   // the word this wave owns
    const int lane = threadIdx.x & (kWave - 1);
    const size_t par0 = (size_t)blockIdx.x * kCoopStates;
    int valid = (int)(n_parents - par0 < (size_t)kCoopStates ? n_parents - par0 : (size_t)kCoopStates);
    coop_load_constants(&d_coop, L);
    // children of this block: valid * ARITY digests, contiguous -> stage[child index * 32]
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(children + par0 * ARITY * 32);
        const int chunks = valid * ARITY * 2;
#pragma unroll
        for (int c = threadIdx.x; c < kCoopStates * ARITY * 2; c += kCoopThreads)
            if (c < chunks) *reinterpret_cast<uint4 *>(L.stage + c * 16) = g[c];
    }
    __syncthreads();
    size_t level_off = 0;                    // byte offset of the current level inside out_all
    size_t level_n = n_parents;              // digests in the current level (whole tree level)
    size_t blk_first = par0;                 // index of this block's first digest in the current level
#pragma unroll 1
    for (int j = 0; j < n_levels; j++) {
        Fr w;
        if (wv == 0) {
            w = tag;
        } else if (wv <= ARITY) {
            const uint4 *p = reinterpret_cast<const uint4 *>(L.stage + (lane * ARITY + (wv - 1)) * 32);
            uint4 lo = p[0], hi = p[1];
            w.l[0] = lo.x; w.l[1] = lo.y; w.l[2] = lo.z; w.l[3] = lo.w;
            w.l[4] = hi.x; w.l[5] = hi.y; w.l[6] = hi.z; w.l[7] = hi.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) w.l[i] = 0;
        }
        const F29 fin = coop_rounds(&d_coop, L, wv, to_f29(w));   // barriers inside: `stage` has been read
        if (wv == out_idx) {
            const Fr o = coop_finish(&d_coop, fin);
            if (lane < valid) {
                const uint4 lo = make_uint4(o.l[0], o.l[1], o.l[2], o.l[3]);
                const uint4 hi = make_uint4(o.l[4], o.l[5], o.l[6], o.l[7]);
                uint4 *p = reinterpret_cast<uint4 *>(L.stage + lane * 32);
                p[0] = lo;
                p[1] = hi;
                if (out_all != nullptr) {
                    uint4 *q = reinterpret_cast<uint4 *>(out_all + level_off + (blk_first + lane) * 32);
                    q[0] = lo;
                    q[1] = hi;
                }
                if (out_last != nullptr && j == n_levels - 1) {
                    uint4 *q = reinterpret_cast<uint4 *>(out_last + (blk_first + lane) * 32);
                    q[0] = lo;
                    q[1] = hi;
                }
            }
        }
        __syncthreads();
        level_off += level_n * 32;
        level_n /= ARITY;
        blk_first /= ARITY;
        valid /= ARITY;
    }
}

// Openings (authentication paths): for query t with leaf index idx, level l = 0 .. depth-1, the ARITY-1
// siblings of the path node at that level, in child order with the path node's own position skipped:
//   paths[t][l][s] (32 B each).  Level 0 siblings are leaves, level l >= 1 siblings are digests of tree level
//   l-1 (layout of hades252_merkle_build_dev; level sizes n_l = ceil(n_{l-1} / ARITY)).  A sibling position past the
//   end of its level is the level's padding digest pad[l] (NULL = zero).  One thread per 16-byte half digest.
template <int ARITY>
__global__ void __launch_bounds__(kBlock) k_merkle_open(const uint8_t *__restrict__ leaves,
                                                        const uint8_t *__restrict__ tree, size_t n_leaves, int depth,
                                                        const uint64_t *__restrict__ indices, size_t n_queries,
                                                        uint8_t *__restrict__ paths, const uint8_t *__restrict__ pad) {
    const size_t per_query = (size_t)depth * (ARITY - 1) * 2;
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (tid >= n_queries * per_query) return;
    const size_t t = tid / per_query;
    const int rem = (int)(tid - t * per_query);
    const int l = rem / ((ARITY - 1) * 2), sh = rem - l * (ARITY - 1) * 2, s = sh >> 1, half = sh & 1;
    size_t node = indices[t];                 // index of the path node at level l (level 0 = leaves)
    if (node >= n_leaves) {                   // never read outside the tree: an invalid index yields an all-zero path
        *reinterpret_cast<uint4 *>(paths + tid * 16) = make_uint4(0, 0, 0, 0);
        return;
    }
    const uint8_t *level = leaves;
    size_t level_n = n_leaves, off = 0;
    for (int i = 0; i < l; i++) {
        node /= ARITY;
        level_n = (level_n + ARITY - 1) / ARITY;
        level = tree + off;
        off += level_n * 32;
    }
    const size_t first = node - node % ARITY;
    const int pos = (int)(node % ARITY);
    const int sib = s < pos ? s : s + 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (first + sib < level_n)
        v = *reinterpret_cast<const uint4 *>(level + (first + sib) * 32 + half * 16);
    else if (pad != nullptr)
        v = *reinterpret_cast<const uint4 *>(pad + (size_t)l * 32 + half * 16);
    *reinterpret_cast<uint4 *>(paths + tid * 16) = v;
}

// Batched sponge over the permutation (the caller shape of dusk-poseidon's sponge hash, reference
// README.md:9; that crate is NOT part of the reference tree, so the convention -- capacity word,
// padding -- is a parameter and parity is pinned only to this repo's oracle: CONVENTION UNPINNED).
// Lane i hashes message i = scalars[off_i .. off_i + len_i): state = [capacity, 0, 0, 0, 0]; every block
// of 4 scalars is added to words 1..4 and followed by a permutation; pad_mode 1 appends a single 1
// (then zeros) first; at least one permutation.  Digest = word 1.
//   * variable length: `offsets` / `lengths` per message (NULL: message i = [i*fixed_len, (i+1)*fixed_len));
//     every lane runs to its WAVE's maximum block count and latches its digest after its own last block
//     (later permutations of that lane work on don't-care data).  Callers with very ragged batches should
//     bucket messages by block count so that the 64 messages of a wave are alike.
//   * message blocks are staged through the wave's LDS slab: 8 lanes fetch the 128 contiguous bytes of one
//     message block, 8 messages per load instruction -- no lane walks HBM with a message-sized stride.
// orders this wave's LDS traffic (other lanes' slab writes before my reads, my reads before the next writes)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = __shfl((uint32_t)v, src, kWave), hi = __shfl((uint32_t)(v >> 32), src, kWave);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ Fr one_mont_word() {                    // 1 * 2^256 mod p
    Fr one;                                                         // (member by member: a table would live in scratch)
    one.l[0] = 0xfffffffeu; one.l[1] = 0x00000001u; one.l[2] = 0x00034802u; one.l[3] = 0x5884b7fau;
    one.l[4] = 0xecbc4ff5u; one.l[5] = 0x998c4fefu; one.l[6] = 0xacc5056fu; one.l[7] = 0x1824b159u;
    return one;
}
// c ? a : b, limb by limb (v_cndmask; a ternary over whole scalars may be turned into a table in scratch)
__device__ __forceinline__ Fr fr_select(bool c, const Fr &a, const Fr &b) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}

__global__ void __launch_bounds__(kBlock, 3) k_sponge(const uint8_t *__restrict__ scalars,
                                                      const uint64_t *__restrict__ offsets,
                                                      const uint64_t *__restrict__ lengths,
                                                      uint8_t *__restrict__ digests, size_t n_msgs, size_t fixed_len,
                                                      Fr capacity, int pad_mode, size_t n_scalars, int *bad_count,
                                                      const uint32_t *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<4>(lds);
    constexpr int kRec = lds_rec_bytes(4);
    const int lane = threadIdx.x & (kWave - 1);
    const size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    const bool live = rec0 + lane < n_msgs;
    // `order` (may be NULL): the messages sorted by block count (k_sponge_* below), so that the 64 messages of a wave
    // need about the same number of permutations; slot rec0 + lane then hashes message order[rec0 + lane]
    const size_t me = !live ? 0 : (order != nullptr ? (size_t)order[rec0 + lane] : rec0 + lane);
    const uint64_t off = live ? (offsets != nullptr ? offsets[me] : (uint64_t)me * fixed_len) : 0;
    uint64_t len = live ? (lengths != nullptr ? lengths[me] : (uint64_t)fixed_len) : 0;
    // a message that does not lie inside the pool is never read: it is hashed as the empty message and counted
    if (live && (off > n_scalars || len > n_scalars - off)) {
        len = 0;
        if (bad_count != nullptr) atomicAdd(bad_count, 1);
    }
    uint64_t blocks = (len + (pad_mode == 1 ? 1 : 0) + 3) / 4;
    if (blocks == 0) blocks = 1;
    if (!live) blocks = 0;
    // wave-uniform trip count: the slab is wave-private and a wave's LDS operations execute in order, so the
    // staging below needs no block-wide barrier (only compiler fences)
    uint64_t mx = blocks;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint64_t other = shfl_u64(mx, lane ^ o);
        mx = other > mx ? other : mx;
    }

    const Fr one_mont = one_mont_word();
    Fr st[5];
    st[0] = capacity;
#pragma unroll
    for (int w = 1; w < 5; w++)
#pragma unroll
        for (int i = 0; i < 8; i++) st[w].l[i] = 0;
    Fr dig = st[1];
#pragma unroll 1
    for (uint64_t t = 0; t < mx; t++) {
        // stage block t of all 64 messages: lane = (message 8k + lane/8, 16-byte part lane%8)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int m = 8 * k + (lane >> 3), part = lane & 7;
            const uint64_t moff = shfl_u64(off, m), mlen = shfl_u64(len, m);
            const uint64_t idx = 4 * t + (part >> 1);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < mlen) v = *reinterpret_cast<const uint4 *>(scalars + (moff + idx) * 32 + (part & 1) * 16);
            *reinterpret_cast<uint4 *>(slab + m * kRec + part * 16) = v;
        }
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 *p = reinterpret_cast<const uint4 *>(slab + lane * kRec + k * 32);
            uint4 lo = p[0], hi = p[1];
            Fr v;
            v.l[0] = lo.x; v.l[1] = lo.y; v.l[2] = lo.z; v.l[3] = lo.w;
            v.l[4] = hi.x; v.l[5] = hi.y; v.l[6] = hi.z; v.l[7] = hi.w;
            if (pad_mode == 1 && 4 * t + k == len) v = one_mont;      // staged value is zero there
            st[1 + k] = fr_add(st[1 + k], v);
        }
        wave_lds_fence();
        Fr out[5];
        fast_perm<5>(&d_fast, st, out, 0);
#pragma unroll
        for (int w = 0; w < 5; w++) st[w] = out[w];
        if (t + 1 == blocks) dig = st[1];
    }
    if (order != nullptr) {                    // scattered: every lane stores its own 32 bytes
        if (live) store_word(digests + me * 32, dig);
        return;
    }
    slab_put<1>(slab, 0, dig);
    slab_flush<1>(digests, rec0, n_msgs, slab);
}

// ---- small batches: one message / state / query per WAVE (hades_lanes.hpp) ---------------------------------------
// The sponge is a chain of dependent permutations per message, so a batch of a few messages (the extreme: ONE long
// message) is pure latency: ~51 us per block here instead of ~175 us with one message per lane.  Same two forms as
// k_perm_lanes.  The helped form needs the same number of permutations from every wave of a block: all run to the
// block's maximum block count and latch their digest after their own last block (as the lanes of a wave do in k_sponge).
struct SpongeGeom {
    uint64_t off, len, blocks;
    bool bad;
};
__device__ __forceinline__ SpongeGeom sponge_geom(const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ lengths,
                                                  size_t me, size_t fixed_len, size_t n_scalars, int pad_mode) {
    SpongeGeom g;
    g.off = offsets != nullptr ? offsets[me] : (uint64_t)me * fixed_len;
    g.len = lengths != nullptr ? lengths[me] : (uint64_t)fixed_len;
    g.bad = g.off > n_scalars || g.len > n_scalars - g.off;          // not inside the pool: never read, hashed as empty
    if (g.bad) g.len = 0;
    g.blocks = (g.len + (pad_mode == 1 ? 1 : 0) + 3) / 4;
    if (g.blocks == 0) g.blocks = 1;
    return g;
}

template <bool HELPED>
__global__ void __launch_bounds__(kLanesWaves *kWave) k_sponge_lanes(const uint8_t *__restrict__ scalars,
                                                                     const uint64_t *__restrict__ offsets,
                                                                     const uint64_t *__restrict__ lengths,
                                                                     uint8_t *__restrict__ digests, size_t n_msgs,
                                                                     size_t fixed_len, Fr capacity, int pad_mode,
                                                                     size_t n_scalars, int *bad_count) {
    __shared__ LanesLds L[kLanesWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    constexpr int kPer = HELPED ? kLanesWaves - 1 : kLanesWaves;
    const size_t me = (size_t)blockIdx.x * kPer + wave;
    uint64_t trips = 0;
    if constexpr (HELPED) {
#pragma unroll
        for (int s = 0; s < kPer; s++) {
            const size_t m = (size_t)blockIdx.x * kPer + s;
            if (m < n_msgs) {
                const uint64_t b = sponge_geom(offsets, lengths, m, fixed_len, n_scalars, pad_mode).blocks;
                trips = b > trips ? b : trips;
            }
        }
        if (wave == kPer) {
            for (uint64_t t = 0; t < trips; t++)
                lanes_helper<kPer>(&d_lanes, *reinterpret_cast<LanesLds(*)[kPer]>(L));
            return;
        }
        if (me >= n_msgs) {
            for (uint64_t t = 0; t < trips; t++) lanes_idle();
            return;
        }
    } else {
        if (me >= n_msgs) return;
    }
    const SpongeGeom g = sponge_geom(offsets, lengths, me, fixed_len, n_scalars, pad_mode);
    if constexpr (!HELPED) trips = g.blocks;
    if (g.bad && lane == 0 && bad_count != nullptr) atomicAdd(bad_count, 1);
    auto block_word = [&](uint64_t t) {                              // lane 1 + k: scalar 4t + k of the message
        Fr v = zero_word();
        if (lane >= 1 && lane <= 4) {
            const uint64_t idx = 4 * t + (uint64_t)(lane - 1);
            if (idx < g.len)
                v = load_word(scalars + (g.off + idx) * 32);
            else if (pad_mode == 1 && idx == g.len)
                v = one_mont_word();
        }
        return v;
    };
    Fr st = lane == 0 ? capacity : zero_word();
    Fr dig = zero_word(), nxt = block_word(0);
#pragma unroll 1
    for (uint64_t t = 0; t < trips; t++) {
        if (lane >= 1 && lane <= 4) st = fr_add(st, nxt);
        nxt = block_word(t + 1);                                     // in flight during the permutation
        st = lanes_perm<HELPED>(&d_lanes, L[wave], st);
        if (t + 1 == g.blocks) dig = st;
    }
    if (lane == 1) store_word(digests + me * 32, dig);
}

// streaming absorb, one state per wave
template <bool HELPED>
__global__ void __launch_bounds__(kLanesWaves *kWave) k_sponge_absorb_lanes(uint8_t *states, const uint8_t *__restrict__ blocks,
                                                                            size_t n, int blocks_each) {
    __shared__ LanesLds L[kLanesWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    constexpr int kPer = HELPED ? kLanesWaves - 1 : kLanesWaves;
    const size_t me = (size_t)blockIdx.x * kPer + wave;
    if constexpr (HELPED) {
        if (wave == kPer) {
            for (int t = 0; t < blocks_each; t++) lanes_helper<kPer>(&d_lanes, *reinterpret_cast<LanesLds(*)[kPer]>(L));
            return;
        }
        if (me >= n) {
            for (int t = 0; t < blocks_each; t++) lanes_idle();
            return;
        }
    } else {
        if (me >= n) return;
    }
    uint8_t *mine = states + me * 160 + (lane < 5 ? lane : 0) * 32;
    const uint8_t *blk = blocks + me * (size_t)blocks_each * 128 + (lane >= 1 && lane <= 4 ? lane - 1 : 0) * 32;
    Fr st = lane < 5 ? load_word(mine) : zero_word();
    Fr nxt = load_word(blk);
#pragma unroll 1
    for (int t = 0; t < blocks_each; t++) {
        if (lane >= 1 && lane <= 4) st = fr_add(st, nxt);
        if (t + 1 < blocks_each) nxt = load_word(blk + (size_t)(t + 1) * 128);
        st = lanes_perm<HELPED>(&d_lanes, L[wave], st);
    }
    if (lane < 5) store_word(mine, st);
}

// path verification, one query per wave: `depth` dependent permutations
template <int ARITY, bool HELPED>
__global__ void __launch_bounds__(kLanesWaves *kWave) k_merkle_verify_lanes(const uint8_t *__restrict__ leaves,
                                                                            const uint64_t *__restrict__ indices,
                                                                            const uint8_t *__restrict__ paths,
                                                                            size_t n_queries, int depth, Fr tag, int out_idx,
                                                                            uint8_t *__restrict__ roots) {
    __shared__ LanesLds L[kLanesWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    constexpr int kPer = HELPED ? kLanesWaves - 1 : kLanesWaves;
    const size_t q = (size_t)blockIdx.x * kPer + wave;
    if constexpr (HELPED) {
        if (wave == kPer) {
            for (int l = 0; l < depth; l++) lanes_helper<kPer>(&d_lanes, *reinterpret_cast<LanesLds(*)[kPer]>(L));
            return;
        }
        if (q >= n_queries) {
            for (int l = 0; l < depth; l++) lanes_idle();
            return;
        }
    } else {
        if (q >= n_queries) return;
    }
    uint64_t idx = indices[q];
    const uint8_t *mine = paths + q * (size_t)depth * (ARITY - 1) * 32;
    Fr node = load_word(leaves + q * 32);                            // every lane holds the path node
    auto sibling = [&](int l, uint64_t at) {                         // lane 1 + c: child c of level l, unless it is the node
        Fr v = zero_word();
        const int pos = (int)(at % ARITY), c = lane - 1;
        if (lane >= 1 && lane <= ARITY && c != pos) v = load_word(mine + ((size_t)l * (ARITY - 1) + (c < pos ? c : c - 1)) * 32);
        return v;
    };
    Fr sib = sibling(0, idx);
#pragma unroll 1
    for (int l = 0; l < depth; l++) {
        const int pos = (int)(idx % ARITY);
        idx /= ARITY;
        const Fr in = fr_select(lane == 0, tag, fr_select(lane == pos + 1, node, sib));
        if (l + 1 < depth) sib = sibling(l + 1, idx);                // in flight during the permutation
        const Fr out = lanes_perm<HELPED>(&d_lanes, L[wave], in);
#pragma unroll
        for (int i = 0; i < 8; i++) node.l[i] = __builtin_amdgcn_readlane(out.l[i], out_idx);
    }
    if (lane == 0) store_word(roots + q * 32, node);
}

// ---- mid-size batches (up to kCoopMaxStates): five waves per message / state / query (hades_coop.hpp) ------------
// Same chains on the five-waves arithmetic: ~106 us per dependent permutation instead of ~160 with one per lane.  A block
// holds 64 chains (lane = chain, wave = state word); every wave runs the block's maximum trip count (coop_rounds
// contains block barriers) and the results are latched per lane.
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint64_t other = shfl_u64(v, lane ^ o);
        v = other > v ? other : v;
    }
    return v;
}

__global__ void __launch_bounds__(kCoopThreads) k_sponge_coop(const uint8_t *__restrict__ scalars,
                                                             const uint64_t *__restrict__ offsets,
                                                             const uint64_t *__restrict__ lengths,
                                                             uint8_t *__restrict__ digests, size_t n_msgs, size_t fixed_len,
                                                             Fr capacity, int pad_mode, size_t n_scalars, int *bad_count) {
    __shared__ CoopLds L;
    const int wv = coop_word_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));   // the word this wave owns
    const int lane = threadIdx.x & (kWave - 1);
    const size_t me = (size_t)blockIdx.x * kCoopStates + lane;
    const bool live = me < n_msgs;
    coop_load_constants(&d_coop, L);
    SpongeGeom g = {0, 0, 0, false};
    if (live) g = sponge_geom(offsets, lengths, me, fixed_len, n_scalars, pad_mode);
    if (g.bad && wv == 0 && bad_count != nullptr) atomicAdd(bad_count, 1);
    const uint64_t trips = wave_max_u64(g.blocks);
    auto block_word = [&](uint64_t t) {                              // wave 1 + k: scalar 4t + k of the lane's message
        Fr v = zero_word();
        if (wv >= 1) {
            const uint64_t idx = 4 * t + (uint64_t)(wv - 1);
            if (idx < g.len)
                v = load_word(scalars + (g.off + idx) * 32);
            else if (pad_mode == 1 && idx == g.len && live)
                v = one_mont_word();
        }
        return v;
    };
    Fr st = wv == 0 ? capacity : zero_word();
    Fr dig = zero_word(), nxt = block_word(0);
    __syncthreads();                                                 // the constants are in LDS
#pragma unroll 1
    for (uint64_t t = 0; t < trips; t++) {
        if (wv >= 1) st = fr_add(st, nxt);
        nxt = block_word(t + 1);
        st = coop_finish(&d_coop, coop_rounds(&d_coop, L, wv, to_f29(st)));
        if (t + 1 == g.blocks) dig = st;
    }
    if (wv == 1 && live) store_word(digests + me * 32, dig);
}

__global__ void __launch_bounds__(kCoopThreads) k_sponge_absorb_coop(uint8_t *states, const uint8_t *__restrict__ blocks,
                                                                    size_t n, int blocks_each) {
    __shared__ CoopLds L;
    const int wv = coop_word_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    const int lane = threadIdx.x & (kWave - 1);
    const size_t me = (size_t)blockIdx.x * kCoopStates + lane;
    const bool live = me < n;
    coop_load_constants(&d_coop, L);
    uint8_t *mine = states + (live ? me : 0) * 160 + wv * 32;
    const uint8_t *blk = blocks + (live ? me : 0) * (size_t)blocks_each * 128 + (wv >= 1 ? wv - 1 : 0) * 32;
    Fr st = load_word(mine), nxt = load_word(blk);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < blocks_each; t++) {
        if (wv >= 1) st = fr_add(st, nxt);
        if (t + 1 < blocks_each) nxt = load_word(blk + (size_t)(t + 1) * 128);
        st = coop_finish(&d_coop, coop_rounds(&d_coop, L, wv, to_f29(st)));
    }
    if (live) store_word(mine, st);
}

// incremental update, one level (see k_merkle_update_fast): lane = query, wave = state word
template <int ARITY>
__global__ void __launch_bounds__(kCoopThreads) k_merkle_update_coop(const uint8_t *__restrict__ children, size_t n_children,
                                                                    uint8_t *__restrict__ parents,
                                                                    const uint64_t *__restrict__ indices, size_t n_updates,
                                                                    size_t n_leaves, uint64_t span, Fr tag, int out_idx,
                                                                    const uint8_t *__restrict__ pad) {
    __shared__ CoopLds L;
    const int wv = coop_word_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    const int lane = threadIdx.x & (kWave - 1);
    const size_t q = (size_t)blockIdx.x * kCoopStates + lane;
    coop_load_constants(&d_coop, L);
    const UpdateWanted wanted{indices, n_leaves, span};
    const bool mine = q < n_updates && wanted(q);
    const size_t parent = mine ? indices[q] / span : 0;
    Fr in = zero_word();
    if (wv == 0) in = tag;
    if (wv >= 1 && wv <= ARITY && mine) in = update_child<ARITY>(children, n_children, parent, wv - 1, pad);
    __syncthreads();
    const F29 fin = coop_rounds(&d_coop, L, wv, to_f29(in));
    if (wv == out_idx && mine) store_word(parents + parent * 32, coop_finish(&d_coop, fin));
}

// path verification: the digest of a level leaves wave `out_idx` and enters the wave of its child position through LDS
template <int ARITY>
__global__ void __launch_bounds__(kCoopThreads) k_merkle_verify_coop(const uint8_t *__restrict__ leaves,
                                                                    const uint64_t *__restrict__ indices,
                                                                    const uint8_t *__restrict__ paths, size_t n_queries,
                                                                    int depth, Fr tag, int out_idx,
                                                                    uint8_t *__restrict__ roots) {
    __shared__ CoopLds L;
    const int wv = coop_word_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    const int lane = threadIdx.x & (kWave - 1);
    const size_t q = (size_t)blockIdx.x * kCoopStates + lane;
    const bool live = q < n_queries;
    coop_load_constants(&d_coop, L);
    uint64_t idx = live ? indices[q] : 0;
    const uint8_t *mine = paths + (live ? q : 0) * (size_t)depth * (ARITY - 1) * 32;
    Fr node = load_word(leaves + (live ? q : 0) * 32);
    auto sibling = [&](int l, uint64_t at) {                         // wave 1 + c: child c of level l, unless it is the node
        Fr v = zero_word();
        const int pos = (int)(at % ARITY), c = wv - 1;
        if (wv >= 1 && wv <= ARITY && c != pos) v = load_word(mine + ((size_t)l * (ARITY - 1) + (c < pos ? c : c - 1)) * 32);
        return v;
    };
    Fr sib = sibling(0, idx);
    __syncthreads();
#pragma unroll 1
    for (int l = 0; l < depth; l++) {
        const int pos = (int)(idx % ARITY);
        idx /= ARITY;
        const Fr in = fr_select(wv == 0, tag, fr_select(wv == pos + 1, node, sib));
        if (l + 1 < depth) sib = sibling(l + 1, idx);
        const F29 fin = coop_rounds(&d_coop, L, wv, to_f29(in));
        if (wv == out_idx) {
            const Fr o = coop_finish(&d_coop, fin);
            uint4 *p = reinterpret_cast<uint4 *>(L.stage + lane * 32);
            p[0] = make_uint4(o.l[0], o.l[1], o.l[2], o.l[3]);
            p[1] = make_uint4(o.l[4], o.l[5], o.l[6], o.l[7]);
        }
        __syncthreads();
        {
            const uint4 *p = reinterpret_cast<const uint4 *>(L.stage + lane * 32);
            const uint4 lo = p[0], hi = p[1];
            node.l[0] = lo.x; node.l[1] = lo.y; node.l[2] = lo.z; node.l[3] = lo.w;
            node.l[4] = hi.x; node.l[5] = hi.y; node.l[6] = hi.z; node.l[7] = hi.w;
        }
        __syncthreads();                                             // everyone has the digest before it is overwritten
    }
    if (wv == 0 && live) store_word(roots + q * 32, node);
}

// ---- ragged batches: counting sort of the message indices by block count --------------------------------
// Three small launches over scratch = {counters[kSpongeBuckets + 1] (u32), order[n_msgs] (u32)}:
//   count: histogram of min(blocks, kSpongeBuckets - 1);  scan: exclusive prefix sums (one block);  scatter: every
//   message takes the next free slot of its bucket.  The order inside a bucket depends on atomics and is irrelevant:
//   every digest goes to its own message's slot.
constexpr int kSpongeBuckets = 1024;
__device__ __forceinline__ uint32_t sponge_bucket(const uint64_t *lengths, size_t i, int pad_mode) {
    uint64_t b = (lengths[i] + (pad_mode == 1 ? 1 : 0) + 3) / 4;
    if (b == 0) b = 1;
    return (uint32_t)(b < (uint64_t)kSpongeBuckets ? b : (uint64_t)kSpongeBuckets - 1);
}
__global__ void __launch_bounds__(kBlock) k_sponge_count(const uint64_t *__restrict__ lengths, size_t n_msgs, int pad_mode,
                                                         uint32_t *__restrict__ counters) {
    __shared__ uint32_t hist[kSpongeBuckets];
    for (int i = threadIdx.x; i < kSpongeBuckets; i += kBlock) hist[i] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_msgs; i += stride)
        atomicAdd(&hist[sponge_bucket(lengths, i, pad_mode)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < kSpongeBuckets; i += kBlock)
        if (hist[i]) atomicAdd(&counters[i], hist[i]);
}
// counters[b] <- number of messages in buckets LONGER than b (long messages first: the tail of the grid is short work)
__global__ void __launch_bounds__(kSpongeBuckets) k_sponge_scan(uint32_t *__restrict__ counters) {
    __shared__ uint32_t v[kSpongeBuckets];
    const int b = threadIdx.x;
    v[b] = counters[kSpongeBuckets - 1 - b];          // reversed: slot b holds bucket (last - b)
    __syncthreads();
    for (int d = 1; d < kSpongeBuckets; d <<= 1) {    // inclusive Hillis-Steele scan
        const uint32_t add = b >= d ? v[b - d] : 0;
        __syncthreads();
        v[b] += add;
        __syncthreads();
    }
    counters[kSpongeBuckets - 1 - b] = b ? v[b - 1] : 0;
}
// One tile of kBlock messages per block: ranks inside the tile come from LDS atomics, and a block reserves its slots of
// every bucket it meets with ONE global atomic (2 M messages with ~10 distinct block counts would otherwise queue on ~10
// addresses).
__global__ void __launch_bounds__(kBlock) k_sponge_scatter(const uint64_t *__restrict__ lengths, size_t n_msgs, int pad_mode,
                                                           uint32_t *__restrict__ counters, uint32_t *__restrict__ order) {
    __shared__ uint32_t hist[kSpongeBuckets];          // count of the tile, then the tile's base slot, per bucket
    for (int i = threadIdx.x; i < kSpongeBuckets; i += kBlock) hist[i] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    uint32_t b = 0, rank = 0;
    if (i < n_msgs) {
        b = sponge_bucket(lengths, i, pad_mode);
        rank = atomicAdd(&hist[b], 1u);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < kSpongeBuckets; j += kBlock)
        if (hist[j]) hist[j] = atomicAdd(&counters[j], hist[j]);
    __syncthreads();
    if (i < n_msgs) order[hist[b] + rank] = (uint32_t)i;
}

// ---- streaming sponge: the state lives in device memory between calls -------------------------------------
// absorb: for each of `blocks_each` blocks of 4 scalars, words 1..4 of every state += block, then the permutation
// (what one round of dusk-poseidon's sponge does, README.md:9); blocks[i][t][0..3] is block t of state i.
__global__ void __launch_bounds__(kBlock, 3) k_sponge_absorb(uint8_t *states, const uint8_t *__restrict__ blocks,
                                                             size_t n, int blocks_each) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    const size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    const size_t me = rec0 + (threadIdx.x & (kWave - 1));
    Fr st[5];
    wave_load_records<5>(states, rec0, n, slab, st);
#pragma unroll 1
    for (int t = 0; t < blocks_each; t++) {
        if (me < n) {
            const uint8_t *b = blocks + (me * (size_t)blocks_each + t) * 128;
#pragma unroll
            for (int k = 0; k < 4; k++) st[1 + k] = fr_add(st[1 + k], load_word(b + k * 32));
        }
        Fr out[5];
        fast_perm<5>(&d_fast, st, out, 0);
#pragma unroll
        for (int w = 0; w < 5; w++) st[w] = out[w];
    }
    wave_store_records<5>(states, rec0, n, slab, st);
}
// states[i] = [capacity, 0, 0, 0, 0]
__global__ void __launch_bounds__(kBlock) k_sponge_init(uint8_t *states, size_t n, Fr capacity) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;      // one 32-byte word per thread
    if (i >= n * 5) return;
    store_word(states + i * 32, i % 5 == 0 ? capacity : zero_word());
}
// digests[i] = word `idx` of state i
__global__ void __launch_bounds__(kBlock) k_sponge_squeeze(const uint8_t *__restrict__ states, uint8_t *__restrict__ digests,
                                                           size_t n, int idx) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;      // one 16-byte half word per thread
    if (i >= n * 2) return;
    *reinterpret_cast<uint4 *>(digests + i * 16) =
        *reinterpret_cast<const uint4 *>(states + (i >> 1) * 160 + (size_t)idx * 32 + (i & 1) * 16);
}

__device__ __forceinline__ uint64_t splitmix_limb(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one u64 limb per thread: perfectly coalesced 8-byte stores
__global__ void __launch_bounds__(kBlock) k_gen_b(uint64_t *__restrict__ out, uint64_t first_elem, size_t n_limbs,
                                                  uint64_t seed) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    size_t stride = (size_t)gridDim.x * kBlock;
    for (; i < n_limbs; i += stride) {
        uint64_t z = splitmix_limb(seed, 4 * first_elem + i);
        if ((i & 3) == 3) z &= 0x3fffffffffffffffull;
        out[i] = z;
    }
}

__global__ void __launch_bounds__(kBlock) k_gen_a(uint8_t *__restrict__ out, uint64_t first_elem, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<1>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    uint64_t v = first_elem + rec0 + (threadIdx.x & (kWave - 1));
    Fr a;
#pragma unroll
    for (int i = 0; i < 8; i++) a.l[i] = 0;
    a.l[0] = (uint32_t)v;
    a.l[1] = (uint32_t)(v >> 32);
    Fr r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.l[i] = d_r2[i];
    Fr st[1];
    st[0] = fr_mul(a, r2);
    wave_store_records<1>(out, rec0, n, slab, st);
}

__device__ __forceinline__ uint64_t digest_mix(uint64_t w, uint64_t idx) {
    uint64_t z = w ^ (idx * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull);
    z = (z ^ (z >> 32)) * 0xD6E8FEB86659FD93ull;
    z = (z ^ (z >> 29)) * 0xBF58476D1CE4E5B9ull;
    return z ^ (z >> 32);
}

__global__ void __launch_bounds__(kBlock) k_digest(const uint64_t *__restrict__ words, uint64_t first_index,
                                                   size_t n, unsigned long long *out4) {
    // thread t always sees word indices == t (mod 4) because the stride is a multiple of 4
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    size_t stride = (size_t)gridDim.x * kBlock;
    uint64_t acc = 0;
    for (; i < n; i += stride) acc += digest_mix(words[i], first_index + i);
    __shared__ unsigned long long part[4];
    if (threadIdx.x < 4) part[threadIdx.x] = 0;
    __syncthreads();
    // lanes with equal (lane & 3) reduce together
    for (int off = 32; off >= 4; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & (kWave - 1)) < 4) atomicAdd(&part[threadIdx.x & 3], (unsigned long long)acc);
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&out4[(first_index + threadIdx.x) & 3], part[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static thread_local int tl_last_hip_error = 0;

#define HIP_TRY(expr)                                \
    do {                                             \
        hipError_t e_ = (expr);                      \
        if (e_ != hipSuccess) {                      \
            tl_last_hip_error = (int)e_;             \
            (void)hipGetLastError();                 \
            return HADES252_ERR_HIP;                 \
        }                                            \
    } while (0)

// device buffers are moved with 16-byte vector loads/stores
static inline bool misaligned(const void *p) { return ((uintptr_t)p & 15u) != 0; }
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
static inline size_t lds_for(int nw) { return (size_t)kWavesPerBlock * lds_wave_bytes(nw); }
static constexpr size_t kMaxLaunchRecords = (size_t)1 << 30;   // grid.x * 256 per launch

static int launch_perm_fast(const uint8_t *in, uint8_t *out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_perm_fast, dim3(blocks_for(n)), dim3(kBlock), lds_for(5), s, in, out, n);
    return HADES252_OK;
}
static Fr fr_from_u64(const uint64_t v[4]) {
    Fr r;
    for (int k = 0; k < 4; k++) {
        r.l[2 * k] = (uint32_t)v[k];
        r.l[2 * k + 1] = (uint32_t)(v[k] >> 32);
    }
    return r;
}

// a batch this small is latency-bound: the five-waves-per-state kernel finishes it in less than half the time
// of one per-lane wave (crossover measured on MI355X: profiles/r2/time_paths.txt)
static constexpr size_t kCoopMaxStates = (size_t)1 << 14;
// ... and one this small (at most one wave per SIMD) is fastest with one state per wave, every product spread over a
// 16-lane row (hades_lanes.hpp): about half the latency of the five-waves kernel
static constexpr size_t kLanesMaxStates = (size_t)1 << 10;
// ... with a helper wave per three states while that still means one block per CU (256 CUs x 3)
static constexpr size_t kLanesHelpedMaxStates = 768;

// one parent per lane (any size, any arity, ragged levels)
static void launch_merkle_level(int arity, const uint8_t *children, size_t n_children, uint8_t *parents, size_t n, Fr tag,
                                int out_idx, const uint8_t *pad, hipStream_t s) {
#define HADES_LAUNCH_LEVEL(A)                                                                                         \
    hipLaunchKernelGGL(k_merkle_level_fast<A>, dim3(blocks_for(n)), dim3(kBlock), lds_for(A), s, children, n_children, \
                       parents, n, tag, out_idx, pad)
    switch (arity) {
        case 1: HADES_LAUNCH_LEVEL(1); break;
        case 2: HADES_LAUNCH_LEVEL(2); break;
        case 3: HADES_LAUNCH_LEVEL(3); break;
        default: HADES_LAUNCH_LEVEL(4); break;
    }
#undef HADES_LAUNCH_LEVEL
}

// one parent per wave (small levels: lowest latency)
static void launch_merkle_lanes(int arity, const uint8_t *children, size_t n_children, uint8_t *parents, size_t n, Fr tag,
                                int out_idx, const uint8_t *pad, hipStream_t s) {
    const bool helped = n <= kLanesHelpedMaxStates;
    const unsigned per_block = helped ? kLanesWaves - 1 : kLanesWaves;
    const dim3 grid((unsigned)((n + per_block - 1) / per_block)), block(kLanesWaves * kWave);
#define HADES_LAUNCH_LANES(A)                                                                                          \
    do {                                                                                                               \
        if (helped)                                                                                                    \
            hipLaunchKernelGGL((k_merkle_lanes<A, true>), grid, block, 0, s, children, n_children, parents, n, tag,   \
                               out_idx, pad);                                                                          \
        else                                                                                                           \
            hipLaunchKernelGGL((k_merkle_lanes<A, false>), grid, block, 0, s, children, n_children, parents, n, tag,  \
                               out_idx, pad);                                                                          \
    } while (0)
    switch (arity) {
        case 1: HADES_LAUNCH_LANES(1); break;
        case 2: HADES_LAUNCH_LANES(2); break;
        case 3: HADES_LAUNCH_LANES(3); break;
        default: HADES_LAUNCH_LANES(4); break;
    }
#undef HADES_LAUNCH_LANES
}

// five waves per parent, full levels only (n_children = arity * n_parents); n_levels > 1 only for arity 2 and 4
static void launch_merkle_coop(int arity, const uint8_t *children, uint8_t *out_all, uint8_t *out_last, size_t n_parents,
                               Fr tag, int out_idx, int n_levels, hipStream_t s) {
    const unsigned grid = (unsigned)((n_parents + kCoopStates - 1) / kCoopStates);
#define HADES_LAUNCH_COOP(A)                                                                                  \
    hipLaunchKernelGGL(k_merkle_coop<A>, dim3(grid), dim3(kCoopThreads), 0, s, children, out_all, out_last, \
                       n_parents, tag, out_idx, n_levels)
    switch (arity) {
        case 1: HADES_LAUNCH_COOP(1); break;
        case 2: HADES_LAUNCH_COOP(2); break;
        case 3: HADES_LAUNCH_COOP(3); break;
        default: HADES_LAUNCH_COOP(4); break;
    }
#undef HADES_LAUNCH_COOP
}

// the ancestors of n_updates changed leaves on one level (k_merkle_update_*): one per wave up to kLanesMaxStates
// queries, five waves per ancestor up to kCoopMaxStates, one per lane above
static void launch_merkle_update(int arity, const uint8_t *children, size_t n_children, uint8_t *parents,
                                 const uint64_t *indices, size_t n_updates, size_t n_leaves, uint64_t span, Fr tag,
                                 int out_idx, const uint8_t *pad, hipStream_t s) {
    const bool lanes = n_updates <= kLanesMaxStates, helped = n_updates <= kLanesHelpedMaxStates;
    const unsigned per_block = helped ? kLanesWaves - 1 : kLanesWaves;
    const dim3 grid((unsigned)((n_updates + per_block - 1) / per_block)), block(kLanesWaves * kWave);
#define HADES_LAUNCH_UPDATE(A)                                                                                          \
    do {                                                                                                                \
        if (!lanes && n_updates <= kCoopMaxStates)                                                                      \
            hipLaunchKernelGGL(k_merkle_update_coop<A>, dim3((unsigned)((n_updates + kCoopStates - 1) / kCoopStates)), \
                               dim3(kCoopThreads), 0, s, children, n_children, parents, indices, n_updates, n_leaves,   \
                               span, tag, out_idx, pad);                                                                \
        else if (!lanes)                                                                                                \
            hipLaunchKernelGGL(k_merkle_update_fast<A>, dim3(blocks_for(n_updates)), dim3(kBlock), 0, s, children,     \
                               n_children, parents, indices, n_updates, n_leaves, span, tag, out_idx, pad);             \
        else if (helped)                                                                                                \
            hipLaunchKernelGGL((k_merkle_update_lanes<A, true>), grid, block, 0, s, children, n_children, parents,     \
                               indices, n_updates, n_leaves, span, tag, out_idx, pad);                                  \
        else                                                                                                            \
            hipLaunchKernelGGL((k_merkle_update_lanes<A, false>), grid, block, 0, s, children, n_children, parents,    \
                               indices, n_updates, n_leaves, span, tag, out_idx, pad);                                  \
    } while (0)
    switch (arity) {
        case 2: HADES_LAUNCH_UPDATE(2); break;
        case 3: HADES_LAUNCH_UPDATE(3); break;
        default: HADES_LAUNCH_UPDATE(4); break;
    }
#undef HADES_LAUNCH_UPDATE
}

// One level, the kernel chosen by size: `n_children` children -> ceil(n_children / arity) parents.
static void launch_merkle_any(int arity, const uint8_t *children, size_t n_children, uint8_t *parents, Fr tag, int out_idx,
                              const uint8_t *pad, hipStream_t s) {
    const size_t n_parents = (n_children + arity - 1) / arity;
    if (n_parents <= kLanesMaxStates)
        launch_merkle_lanes(arity, children, n_children, parents, n_parents, tag, out_idx, pad, s);
    else if (n_parents <= kCoopMaxStates && n_children % arity == 0)
        launch_merkle_coop(arity, children, nullptr, parents, n_parents, tag, out_idx, 1, s);
    else
        launch_merkle_level(arity, children, n_children, parents, n_parents, tag, out_idx, pad, s);
}

static int check_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        tl_last_hip_error = (int)e;
        (void)hipGetLastError();
        return HADES252_ERR_NO_DEVICE;
    }
    return n > 0 ? HADES252_OK : HADES252_ERR_NO_DEVICE;
}

extern "C" {

int hades252_rounds(void) { return HADES252_TOTAL_FULL_ROUNDS + HADES252_PARTIAL_ROUNDS; }

int hades252_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *hades252_strerror(int code) {
    switch (code) {
        case HADES252_OK: return "ok";
        case HADES252_ERR_INVALID_ARG: return "invalid argument";
        case HADES252_ERR_HIP: return "HIP runtime error (see hades252_last_hip_error)";
        case HADES252_ERR_NOT_CANONICAL: return "input scalar is not canonical (>= p)";
        case HADES252_ERR_NO_DEVICE: return "no HIP device available";
        case HADES252_ERR_SCRATCH: return "scratch buffer too small";
        case HADES252_ERR_OUT_OF_CONSTANTS: return "Hades252 out of ARK constants";
        default: return "unknown error";
    }
}

int hades252_last_hip_error(void) { return tl_last_hip_error; }

const char *hades252_version(void) { return "hades252-amd 0.1.0 (gfx950)"; }

// ---- perm ---------------------------------------------------------------------------------
int hades252_perm_batch_dev_ex(void *d_states, size_t n_perms, void *stream, int kernel) {
    if (n_perms == 0) return HADES252_OK;
    if (d_states == nullptr || misaligned(d_states)) return HADES252_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    uint8_t *p = (uint8_t *)d_states;
    // small batches are latency-bound: five waves per state (hades_coop.hpp); large ones one state per lane
    if (kernel == HADES252_KERNEL_DEFAULT)
        kernel = n_perms <= kLanesMaxStates ? HADES252_KERNEL_LANES
                                            : (n_perms <= kCoopMaxStates ? HADES252_KERNEL_COOP : HADES252_KERNEL_FAST);
    for (size_t off = 0; off < n_perms; off += kMaxLaunchRecords) {
        size_t n = n_perms - off < kMaxLaunchRecords ? n_perms - off : kMaxLaunchRecords;
        if (kernel == HADES252_KERNEL_LANES) {
            if (n <= kLanesHelpedMaxStates)
                hipLaunchKernelGGL(k_perm_lanes<true>, dim3((unsigned)((n + kLanesWaves - 2) / (kLanesWaves - 1))),
                                   dim3(kLanesWaves * kWave), 0, s, p + off * 160, n);
            else
                hipLaunchKernelGGL(k_perm_lanes<false>, dim3((unsigned)((n + kLanesWaves - 1) / kLanesWaves)),
                                   dim3(kLanesWaves * kWave), 0, s, p + off * 160, n);
        } else if (kernel == HADES252_KERNEL_COOP) {
            hipLaunchKernelGGL(k_perm_coop, dim3((unsigned)((n + kCoopStates - 1) / kCoopStates)), dim3(kCoopThreads), 0,
                               s, p + off * 160, n);
        } else if (kernel == HADES252_KERNEL_LITERAL) {
            hipLaunchKernelGGL(k_states_literal<OP_PERM>, dim3(blocks_for(n)), dim3(kBlock), lds_for(5), s,
                               p + off * 160, n, 0);
        } else if (kernel == HADES252_KERNEL_FAST) {
            int rc = launch_perm_fast(p + off * 160, p + off * 160, n, s);
            if (rc != HADES252_OK) return rc;
        } else {
            return HADES252_ERR_INVALID_ARG;
        }
        HIP_TRY(hipGetLastError());
    }
    return HADES252_OK;
}

int hades252_perm_batch_dev(void *d_states, size_t n_perms, void *stream) {
    return hades252_perm_batch_dev_ex(d_states, n_perms, stream, HADES252_KERNEL_DEFAULT);
}

// ---- page-locked host memory --------------------------------------------------------------------
// The reference's caller owns a `&mut [BlsScalar]` in ordinary (pageable) memory (src/strategies.rs:140).  DMA needs
// page-locked memory; locking and unlocking the caller's buffer on every call costs more than the transfer itself
// for mid-sized batches.  A caller that keeps its states in one long-lived buffer therefore pins it ONCE, either by
// allocating it here (hades252_host_alloc) or by registering its own allocation (hades252_host_register); the
// host-pointer entry points recognise such memory and go straight to DMA.  Per-call registration stays as the
// fallback for everything else.
struct PinnedRange {
    uintptr_t lo, hi;
    bool owned;                 // allocated by hades252_host_alloc (freed by hades252_host_free)
};
static std::mutex g_pin_mu;
static std::vector<PinnedRange> g_pins;

int hades252_host_alloc(void **out, size_t bytes) {
    if (out == nullptr || bytes == 0) return HADES252_ERR_INVALID_ARG;
    *out = nullptr;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    void *p = nullptr;
    // portable: page-locked for every device (hades252_perm_batch_multi); mapped: kernels may access it directly
    HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped));
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        g_pins.push_back({(uintptr_t)p, (uintptr_t)p + bytes, true});
    }
    *out = p;
    return HADES252_OK;
}

static int forget_range(void *p, bool owned) {       // 1 = found and removed
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (size_t i = 0; i < g_pins.size(); i++)
        if (g_pins[i].lo == (uintptr_t)p && g_pins[i].owned == owned) {
            g_pins.erase(g_pins.begin() + i);
            return 1;
        }
    return 0;
}

int hades252_host_free(void *p) {
    if (p == nullptr) return HADES252_OK;
    if (!forget_range(p, true)) return HADES252_ERR_INVALID_ARG;       // not from hades252_host_alloc
    HIP_TRY(hipHostFree(p));
    return HADES252_OK;
}

int hades252_host_register(void *p, size_t bytes) {
    if (p == nullptr || bytes == 0) return HADES252_ERR_INVALID_ARG;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pins.push_back({(uintptr_t)p, (uintptr_t)p + bytes, false});
    return HADES252_OK;
}

int hades252_host_unregister(void *p) {
    if (p == nullptr) return HADES252_OK;
    if (!forget_range(p, false)) return HADES252_ERR_INVALID_ARG;      // not registered through this library
    HIP_TRY(hipHostUnregister(p));
    return HADES252_OK;
}

// is [p, p + bytes) page-locked already?  First the ranges this library handed out or registered, then the
// runtime's own view (memory the caller pinned with hipHostMalloc / hipHostRegister directly).
static bool host_range_pinned(const void *p, size_t bytes) {
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (const PinnedRange &r : g_pins)
            if (lo >= r.lo && hi <= r.hi) return true;
    }
    hipPointerAttribute_t a0, a1;
    if (hipPointerGetAttributes(&a0, p) != hipSuccess ||
        hipPointerGetAttributes(&a1, (const uint8_t *)p + (bytes - 1)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a0.type == hipMemoryTypeHost && a1.type == hipMemoryTypeHost;
}

int hades252_host_is_pinned(const void *p, size_t bytes) {
    if (p == nullptr || bytes == 0) return 0;
    return host_range_pinned(p, bytes) ? 1 : 0;
}

// ---- host-pointer path ------------------------------------------------------------------------
// A pooled "pipe" per concurrent host call: three streams (host->device copies, kernels, device->host copies),
// kPipeSlots chunk buffers in device memory and the events that chain them, so that a call pays neither hipMalloc /
// hipFree nor stream / event creation (about 1 ms together) -- the reference's callers issue many small calls.
// Pipes are created on demand, handed out exclusively and returned; the pool is bounded by the peak number of
// concurrent calls and by kPipeSlots x 40 MiB of device memory per pipe.
constexpr int kPipeSlots = 6;
struct HostPipe {
    int device = -1;
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
    void *buf = nullptr;          // kPipeSlots slots of slot_cap bytes
    size_t slot_cap = 0;
    hipEvent_t in_done[kPipeSlots] = {}, k_done[kPipeSlots] = {}, out_done[kPipeSlots] = {};
    void *pinned = nullptr;       // small-call staging: page-locked host memory the kernels access directly
    void *pinned_dev = nullptr;   // ... and its device-side address
};
// Calls of at most this many states skip both DMA copies: the states are copied (by the CPU) into a
// page-locked buffer that the kernel reads and writes over PCIe itself -- one launch + one synchronisation.
static constexpr size_t kPinnedStates = 256;
static std::mutex g_pool_mu;
static std::vector<HostPipe> g_pool;

static void destroy_pipe(HostPipe &p) {
    if (p.pinned) (void)hipHostFree(p.pinned);
    if (p.buf) (void)hipFree(p.buf);
    for (int i = 0; i < kPipeSlots; i++) {
        if (p.in_done[i]) (void)hipEventDestroy(p.in_done[i]);
        if (p.k_done[i]) (void)hipEventDestroy(p.k_done[i]);
        if (p.out_done[i]) (void)hipEventDestroy(p.out_done[i]);
    }
    if (p.s_in) (void)hipStreamDestroy(p.s_in);
    if (p.s_k) (void)hipStreamDestroy(p.s_k);
    if (p.s_out) (void)hipStreamDestroy(p.s_out);
    (void)hipGetLastError();
    p = HostPipe();
}

static void release_pipe(const HostPipe &p) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(p);
}

// slot_bytes == 0: a small call (needs the page-locked staging buffer, no device buffer)
static int acquire_pipe(size_t slot_bytes, HostPipe &out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HostPipe p;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        int best = -1;
        for (int i = 0; i < (int)g_pool.size(); i++) {
            if (g_pool[i].device != dev) continue;
            if (best < 0) {
                best = i;
            } else if (slot_bytes == 0) {
                // small call: a pipe that already has its staging buffer, and the smallest device buffer among those
                // (big buffers stay available to concurrent large calls)
                const bool bp = g_pool[best].pinned != nullptr, ip = g_pool[i].pinned != nullptr;
                if ((ip && !bp) || (ip == bp && g_pool[i].slot_cap < g_pool[best].slot_cap)) best = i;
            } else {
                // large call: the smallest buffer that fits, else the largest
                const size_t bc = g_pool[best].slot_cap, ic = g_pool[i].slot_cap;
                if (bc >= slot_bytes ? (ic >= slot_bytes && ic < bc) : ic > bc) best = i;
            }
        }
        if (best >= 0) {
            p = g_pool[best];
            g_pool.erase(g_pool.begin() + best);
        }
    }
    auto fail = [&](hipError_t e) {
        tl_last_hip_error = (int)e;
        (void)hipGetLastError();
        destroy_pipe(p);                       // nothing half-built ever returns to the pool
        return HADES252_ERR_HIP;
    };
    hipError_t e = hipSuccess;
    if (p.device < 0) {
        p.device = dev;
        if ((e = hipStreamCreateWithFlags(&p.s_in, hipStreamNonBlocking)) != hipSuccess) return fail(e);
        if ((e = hipStreamCreateWithFlags(&p.s_k, hipStreamNonBlocking)) != hipSuccess) return fail(e);
        if ((e = hipStreamCreateWithFlags(&p.s_out, hipStreamNonBlocking)) != hipSuccess) return fail(e);
        for (int i = 0; i < kPipeSlots; i++) {
            if ((e = hipEventCreateWithFlags(&p.in_done[i], hipEventDisableTiming)) != hipSuccess) return fail(e);
            if ((e = hipEventCreateWithFlags(&p.k_done[i], hipEventDisableTiming)) != hipSuccess) return fail(e);
            if ((e = hipEventCreateWithFlags(&p.out_done[i], hipEventDisableTiming)) != hipSuccess) return fail(e);
        }
    }
    if (slot_bytes == 0 && p.pinned_dev == nullptr) {
        if (p.pinned) (void)hipHostFree(p.pinned);
        p.pinned = nullptr;
        if ((e = hipHostMalloc(&p.pinned, kPinnedStates * 160, hipHostMallocMapped)) != hipSuccess) return fail(e);
        if ((e = hipHostGetDevicePointer(&p.pinned_dev, p.pinned, 0)) != hipSuccess) return fail(e);
    }
    if (p.slot_cap < slot_bytes) {
        if (p.buf) (void)hipFree(p.buf);
        p.buf = nullptr;
        p.slot_cap = 0;
        if ((e = hipMalloc(&p.buf, slot_bytes * kPipeSlots)) != hipSuccess) return fail(e);
        p.slot_cap = slot_bytes;
    }
    out = p;
    return HADES252_OK;
}

static size_t host_chunk_states(size_t n_perms) {
    // Chunks small enough that the exposed first copy-in and last copy-out are a small part of the call (about 32
    // chunks), large enough that a chunk's kernel is a full-rate launch (>= 2^16 states) and at most 40 MiB.
    static const size_t forced = []() -> size_t {
        const char *e = getenv("HADES252_HOST_CHUNK");
        return e ? (size_t)strtoull(e, nullptr, 0) : 0;
    }();
    if (forced) return forced;
    size_t c = (size_t)1 << 16;
    while (c < ((size_t)1 << 18) && c * 32 < n_perms) c <<= 1;
    return c;
}

// Host batch on the current device.  `bytes_format` inputs have already been validated (all < p).
//   n <= 256           the kernel works on a page-locked staging buffer over PCIe (no DMA copy at all)
//   one chunk          copy in, kernel, copy out on one stream
//   several chunks     three streams chained by events over kPipeSlots chunk buffers: chunk c+1 travels to the device
//                      and chunk c-1 back to the host (PCIe is full duplex) while chunk c is being permuted.  Memory the
//                      caller has not page-locked is locked here for the duration of the call when it can be.
// Roads not taken, measured on this pool (tools/host_pipe_probe.hip, profiles/r3/host_path.txt): a copy-out KERNEL
// storing into the caller's memory doubles the duration of the permutation kernel running beside it and slows the
// copy-in (its posted writes clog the fabric queues): 27-34 GB/s each way at any grid size; the permutation kernel
// storing its results over PCIe itself runs every chunk in lockstep (compute, then a burst of stores): 29-37 GB/s;
// DMA both ways: 43.6 GB/s = 92 % of the 47.4 GB/s the link gives bare copies in both directions at once.
static int perm_batch_host_on_current_device(uint64_t *states, size_t n_perms, bool bytes_format,
                                             bool assume_pinned = false) {
    if (n_perms == 0) return HADES252_OK;
    if (states == nullptr) return HADES252_ERR_INVALID_ARG;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    auto run_kernels = [&](void *d, size_t n, hipStream_t st) {
        if (!bytes_format) return hades252_perm_batch_dev(d, n, st);
        int r = hades252_from_bytes_dev(d, d, n * 5, nullptr, st);
        if (r == HADES252_OK) r = hades252_perm_batch_dev(d, n, st);
        if (r == HADES252_OK) r = hades252_to_bytes_dev(d, d, n * 5, st);
        return r;
    };
    HostPipe pipe;
    if (n_perms <= kPinnedStates) {
        rc = acquire_pipe(0, pipe);
        if (rc != HADES252_OK) return rc;
        memcpy(pipe.pinned, states, n_perms * 160);
        rc = run_kernels(pipe.pinned_dev, n_perms, pipe.s_k);
        hipError_t e = hipStreamSynchronize(pipe.s_k);
        if (rc == HADES252_OK && e == hipSuccess) memcpy(states, pipe.pinned, n_perms * 160);
        release_pipe(pipe);                                 // only now: the staging buffer belongs to the pipe
        if (rc != HADES252_OK) return rc;
        if (e != hipSuccess) {
            tl_last_hip_error = (int)e;
            (void)hipGetLastError();
            return HADES252_ERR_HIP;
        }
        return HADES252_OK;
    }
    const size_t chunk = n_perms < host_chunk_states(n_perms) ? n_perms : host_chunk_states(n_perms);
    const size_t n_chunks = (n_perms + chunk - 1) / chunk;
    uint8_t *h = (uint8_t *)states;
    rc = acquire_pipe(chunk * 160, pipe);
    if (rc != HADES252_OK) return rc;
    bool registered = false;
    auto finish = [&](int code) {
        (void)hipStreamSynchronize(pipe.s_in);
        (void)hipStreamSynchronize(pipe.s_k);
        (void)hipStreamSynchronize(pipe.s_out);
        (void)hipGetLastError();
        release_pipe(pipe);
        if (registered) (void)hipHostUnregister(h);
        return code;
    };
    // Memory the caller has not pinned: page-lock it in place for the duration of the call, so the chunk copies are true
    // DMA and overlap with the kernels (pageable copies are staged by the runtime at ~15 GB/s).  If registration is
    // refused the pageable path is used; HADES252_HOST_PIN=0 disables the attempt.
    static const bool pin_enabled = []() {
        const char *e = getenv("HADES252_HOST_PIN");
        return !(e && e[0] == '0');
    }();
    if (!assume_pinned && pin_enabled && n_perms * 160 >= ((size_t)8 << 20) && !host_range_pinned(h, n_perms * 160)) {
        if (hipHostRegister(h, n_perms * 160, hipHostRegisterDefault) == hipSuccess)
            registered = true;
        else
            (void)hipGetLastError();
    }
#define TRY_FIN(expr)                                \
    do {                                             \
        hipError_t e_ = (expr);                      \
        if (e_ != hipSuccess) {                      \
            tl_last_hip_error = (int)e_;             \
            (void)hipGetLastError();                 \
            return finish(HADES252_ERR_HIP);         \
        }                                            \
    } while (0)
    if (n_chunks == 1) {
        TRY_FIN(hipMemcpyAsync(pipe.buf, h, n_perms * 160, hipMemcpyHostToDevice, pipe.s_k));
        rc = run_kernels(pipe.buf, n_perms, pipe.s_k);
        if (rc != HADES252_OK) return finish(rc);
        TRY_FIN(hipMemcpyAsync(h, pipe.buf, n_perms * 160, hipMemcpyDeviceToHost, pipe.s_k));
        TRY_FIN(hipStreamSynchronize(pipe.s_k));
        return finish(HADES252_OK);
    }
    // The host runs at most kPipeSlots chunks ahead of the device: it waits for the chunk that last used a slot before
    // enqueuing the next one into it.  (A deep backlog of copies, kernels and event waits degrades the overlap --
    // measured: 128 chunks enqueued at once run at a third of the rate of 32.  The link is the bottleneck and has
    // kPipeSlots - 1 chunks queued while the host sleeps, so the wake-up latency is hidden.)
    for (size_t c = 0; c < n_chunks; c++) {
        const int k = (int)(c % kPipeSlots);
        const size_t off = c * chunk, n = n_perms - off < chunk ? n_perms - off : chunk;
        void *d = (uint8_t *)pipe.buf + (size_t)k * pipe.slot_cap;
        if (c >= (size_t)kPipeSlots) TRY_FIN(hipEventSynchronize(pipe.out_done[k]));   // chunk c - kPipeSlots left slot k
        TRY_FIN(hipMemcpyAsync(d, h + off * 160, n * 160, hipMemcpyHostToDevice, pipe.s_in));
        TRY_FIN(hipEventRecord(pipe.in_done[k], pipe.s_in));
        TRY_FIN(hipStreamWaitEvent(pipe.s_k, pipe.in_done[k], 0));
        rc = run_kernels(d, n, pipe.s_k);
        if (rc != HADES252_OK) return finish(rc);
        TRY_FIN(hipEventRecord(pipe.k_done[k], pipe.s_k));
        TRY_FIN(hipStreamWaitEvent(pipe.s_out, pipe.k_done[k], 0));
        TRY_FIN(hipMemcpyAsync(h + off * 160, d, n * 160, hipMemcpyDeviceToHost, pipe.s_out));
        TRY_FIN(hipEventRecord(pipe.out_done[k], pipe.s_out));
    }
    TRY_FIN(hipStreamSynchronize(pipe.s_out));           // the last copy-out is behind everything else
    TRY_FIN(hipStreamSynchronize(pipe.s_k));
    TRY_FIN(hipStreamSynchronize(pipe.s_in));
#undef TRY_FIN
    return finish(HADES252_OK);
}

int hades252_perm_batch(uint64_t *states, size_t n_perms) {
    return perm_batch_host_on_current_device(states, n_perms, false);
}

// input validation only (BlsScalar::from_bytes fails for values >= p before anything is computed)
static bool all_canonical(const uint8_t *bytes, size_t n_scalars) {
    static const uint64_t kP[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull,
                                   0x73eda753299d7d48ull};
    for (size_t i = 0; i < n_scalars; i++) {
        uint64_t v[4];
        memcpy(v, bytes + 32 * i, 32);
        bool less = false;
        for (int k = 3; k >= 0; k--) {
            if (v[k] != kP[k]) {
                less = v[k] < kP[k];
                break;
            }
        }
        if (!less) return false;
    }
    return true;
}

int hades252_perm_batch_bytes(uint8_t *states, size_t n_perms) {
    if (n_perms == 0) return HADES252_OK;
    if (states == nullptr) return HADES252_ERR_INVALID_ARG;
    // reject the whole batch up front, so a failing call leaves the buffer untouched
    if (!all_canonical(states, n_perms * 5)) return HADES252_ERR_NOT_CANONICAL;
    return perm_batch_host_on_current_device((uint64_t *)states, n_perms, true);
}

// Host batch sharded over `n_workers` host threads, worker g taking the contiguous range
// [n g / W, n (g+1) / W) on device g -- or, with HADES252_MULTI_VIRTUAL, on device g % (visible devices), which lets a
// box with fewer GPUs than workers run the very code an 8-GPU node runs (several workers then share a device, each
// with its own pipe).
int hades252_perm_batch_multi_ex(uint64_t *states, size_t n_perms, int n_workers, unsigned flags) {
    if (flags & ~(unsigned)HADES252_MULTI_VIRTUAL) return HADES252_ERR_INVALID_ARG;
    if (n_perms == 0) return HADES252_OK;
    if (states == nullptr) return HADES252_ERR_INVALID_ARG;
    const int avail = hades252_device_count();
    if (avail <= 0) return HADES252_ERR_NO_DEVICE;
    const bool virt = (flags & HADES252_MULTI_VIRTUAL) != 0;
    if (n_workers <= 0) n_workers = avail;
    if (n_workers > (virt ? 64 : avail)) return HADES252_ERR_INVALID_ARG;
    if ((size_t)n_workers > n_perms) n_workers = (int)n_perms;
    // Page-lock the caller's buffer ONCE for all devices (shards share boundary pages: per-shard registration
    // would overlap and be refused for some shards, by a race); portable = visible to every device.
    bool registered = false;
    bool pinned = host_range_pinned(states, n_perms * 160);
    if (!pinned) {
        const char *e = getenv("HADES252_HOST_PIN");
        if (!(e && e[0] == '0') && n_perms * 160 >= ((size_t)8 << 20)) {
            if (hipHostRegister(states, n_perms * 160, hipHostRegisterPortable) == hipSuccess)
                registered = true;
            else
                (void)hipGetLastError();
        }
    }
    std::vector<int> rcs(n_workers, HADES252_OK);
    std::vector<int> hip_errs(n_workers, 0);
    std::vector<std::thread> threads;
    for (int g = 0; g < n_workers; g++) {
        threads.emplace_back([&, g]() {
            size_t b = n_perms * (size_t)g / n_workers, e = n_perms * (size_t)(g + 1) / n_workers;
            hipError_t err = hipSetDevice(virt ? g % avail : g);
            if (err != hipSuccess) {
                rcs[g] = HADES252_ERR_HIP;
                hip_errs[g] = (int)err;
                return;
            }
            // registered or pinned by the caller: skip the per-shard attempt; neither (refused / disabled): also skip
            // it -- a sub-range attempt would only repeat the refusal -- and use the pageable path
            rcs[g] = perm_batch_host_on_current_device(states + 20 * b, e - b, false, true);
            hip_errs[g] = tl_last_hip_error;
        });
    }
    for (auto &t : threads) t.join();
    if (registered) (void)hipHostUnregister(states);
    for (int g = 0; g < n_workers; g++)
        if (rcs[g] != HADES252_OK) {
            tl_last_hip_error = hip_errs[g];
            return rcs[g];
        }
    return HADES252_OK;
}

int hades252_perm_batch_multi(uint64_t *states, size_t n_perms, int n_devices) {
    return hades252_perm_batch_multi_ex(states, n_perms, n_devices, 0);
}

int hades252_perm_trace_dev_ex(const void *d_states, void *d_trace, size_t n_perms, void *stream, int kernel) {
    if (n_perms == 0) return HADES252_OK;
    if (d_states == nullptr || d_trace == nullptr || n_perms > kMaxLaunchRecords || misaligned(d_states) ||
        misaligned(d_trace))
        return HADES252_ERR_INVALID_ARG;
    if (kernel == HADES252_KERNEL_DEFAULT) kernel = HADES252_KERNEL_FAST;
    if (kernel == HADES252_KERNEL_LITERAL)
        hipLaunchKernelGGL(k_perm_trace_literal, dim3(blocks_for(n_perms)), dim3(kBlock), lds_for(5),
                           (hipStream_t)stream, (const uint8_t *)d_states, (uint8_t *)d_trace, n_perms);
    else if (kernel == HADES252_KERNEL_FAST)
        hipLaunchKernelGGL(k_perm_trace_fast, dim3(blocks_for(n_perms)), dim3(kBlock), lds_for(5),
                           (hipStream_t)stream, (const uint8_t *)d_states, (uint8_t *)d_trace, n_perms);
    else
        return HADES252_ERR_INVALID_ARG;
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_witness_wires(void) { return HADES_WITNESS_WIRES; }

int hades252_perm_witness_dev(const void *d_states, void *d_wires, size_t n_perms, void *stream) {
    if (n_perms == 0) return HADES252_OK;
    if (d_states == nullptr || d_wires == nullptr || n_perms > kMaxLaunchRecords || misaligned(d_states) ||
        misaligned(d_wires))
        return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_perm_witness, dim3(blocks_for(n_perms)), dim3(kBlock), lds_for(5), (hipStream_t)stream,
                       (const uint8_t *)d_states, (uint8_t *)d_wires, n_perms);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_perm_trace_dev(const void *d_states, void *d_trace, size_t n_perms, void *stream) {
    return hades252_perm_trace_dev_ex(d_states, d_trace, n_perms, stream, HADES252_KERNEL_DEFAULT);
}

// ---- per-op --------------------------------------------------------------------------------
// `cursor` = position of the constants iterator the trait methods take (src/strategies.rs:33-41);
// the reference panics with "Hades252 out of ARK constants" when it runs dry (:40).
static int states_op_at(int op, void *d_states, size_t n_states, long cursor, void *stream) {
    if (cursor < 0) return HADES252_ERR_INVALID_ARG;
    if (cursor + HADES252_WIDTH > HADES_N_ARK) return HADES252_ERR_OUT_OF_CONSTANTS;
    if (n_states == 0) return HADES252_OK;
    if (d_states == nullptr || n_states > kMaxLaunchRecords || misaligned(d_states)) return HADES252_ERR_INVALID_ARG;
    const dim3 grid(blocks_for(n_states)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    uint8_t *p = (uint8_t *)d_states;
    switch (op) {
        case OP_ARK: hipLaunchKernelGGL(k_states_literal<OP_ARK>, grid, block, lds_for(5), s, p, n_states, (int)cursor); break;
        case OP_FULL: hipLaunchKernelGGL(k_states_fast<OP_FULL>, grid, block, lds_for(5), s, p, n_states, (int)cursor); break;
        default: hipLaunchKernelGGL(k_states_fast<OP_PARTIAL>, grid, block, lds_for(5), s, p, n_states, (int)cursor); break;
    }
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}
int hades252_add_round_key_at_dev(void *d_states, size_t n_states, int cursor, void *stream) {
    return states_op_at(OP_ARK, d_states, n_states, cursor, stream);
}
int hades252_apply_full_round_at_dev(void *d_states, size_t n_states, int cursor, void *stream) {
    return states_op_at(OP_FULL, d_states, n_states, cursor, stream);
}
int hades252_apply_partial_round_at_dev(void *d_states, size_t n_states, int cursor, void *stream) {
    return states_op_at(OP_PARTIAL, d_states, n_states, cursor, stream);
}
int hades252_add_round_key_dev(void *d_states, size_t n_states, int round, void *stream) {
    return states_op_at(OP_ARK, d_states, n_states, 5L * round, stream);
}
int hades252_apply_full_round_dev(void *d_states, size_t n_states, int round, void *stream) {
    return states_op_at(OP_FULL, d_states, n_states, 5L * round, stream);
}
int hades252_apply_partial_round_dev(void *d_states, size_t n_states, int round, void *stream) {
    return states_op_at(OP_PARTIAL, d_states, n_states, 5L * round, stream);
}

int hades252_fr_op_dev(int op, int impl, const void *d_a, const void *d_b, void *d_out, size_t n, void *stream) {
    if (op < FR_ADD || op > FR_FROM_RAW || (impl != 0 && impl != 1)) return HADES252_ERR_INVALID_ARG;
    if (n == 0) return HADES252_OK;
    const bool binary = (op == FR_ADD || op == FR_MUL);
    if (d_a == nullptr || d_out == nullptr || (binary && d_b == nullptr) || n > kMaxLaunchRecords || misaligned(d_a) ||
        misaligned(d_out) || (binary && misaligned(d_b)))
        return HADES252_ERR_INVALID_ARG;
    if (impl == 0)
        hipLaunchKernelGGL(k_fr_op<0>, dim3(blocks_for(n)), dim3(kBlock), lds_for(1), (hipStream_t)stream,
                           (const uint8_t *)d_a, (const uint8_t *)d_b, (uint8_t *)d_out, n, op);
    else
        hipLaunchKernelGGL(k_fr_op<1>, dim3(blocks_for(n)), dim3(kBlock), lds_for(1), (hipStream_t)stream,
                           (const uint8_t *)d_a, (const uint8_t *)d_b, (uint8_t *)d_out, n, op);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_mul_matrix_dev(void *d_states, size_t n_states, void *stream) {
    if (n_states == 0) return HADES252_OK;
    if (d_states == nullptr || n_states > kMaxLaunchRecords || misaligned(d_states)) return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_states_fast<OP_MDS>, dim3(blocks_for(n_states)), dim3(kBlock), lds_for(5),
                       (hipStream_t)stream, (uint8_t *)d_states, n_states, 0);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_quintic_s_box_dev(void *d_scalars, size_t n_scalars, void *stream) {
    if (n_scalars == 0) return HADES252_OK;
    if (d_scalars == nullptr || n_scalars > kMaxLaunchRecords || misaligned(d_scalars)) return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_sbox, dim3(blocks_for(n_scalars)), dim3(kBlock), lds_for(1), (hipStream_t)stream,
                       (uint8_t *)d_scalars, n_scalars);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

// ---- wire format ----------------------------------------------------------------------------
int hades252_from_bytes_dev(const void *d_bytes, void *d_limbs, size_t n_scalars, int *d_bad_count, void *stream) {
    if (n_scalars == 0) return HADES252_OK;
    if (d_bytes == nullptr || d_limbs == nullptr || n_scalars > kMaxLaunchRecords || misaligned(d_bytes) ||
        misaligned(d_limbs))
        return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_wire<1>, dim3(blocks_for((n_scalars + kWirePerThread - 1) / kWirePerThread)), dim3(kBlock), 0,
                       (hipStream_t)stream, (const uint8_t *)d_bytes, (uint8_t *)d_limbs, n_scalars, d_bad_count);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_to_bytes_dev(const void *d_limbs, void *d_bytes, size_t n_scalars, void *stream) {
    if (n_scalars == 0) return HADES252_OK;
    if (d_bytes == nullptr || d_limbs == nullptr || n_scalars > kMaxLaunchRecords || misaligned(d_bytes) ||
        misaligned(d_limbs))
        return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_wire<0>, dim3(blocks_for((n_scalars + kWirePerThread - 1) / kWirePerThread)), dim3(kBlock), 0,
                       (hipStream_t)stream, (const uint8_t *)d_limbs, (uint8_t *)d_bytes, n_scalars, (int *)nullptr);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

// ---- Merkle ----------------------------------------------------------------------------------
static int log_arity(size_t n, int arity) {          // n = arity^k -> k, else -1
    if (arity < 2 || arity > 4) return -1;
    int k = 0;
    while (n > 1) {
        if (n % arity) return -1;
        n /= arity;
        k++;
    }
    return n == 1 ? k : -1;
}

// levels above the leaves of a tree over n_leaves leaves: n_l = ceil(n_{l-1} / arity) until one node is left
int hades252_merkle_depth(size_t n_leaves, int arity) {
    if (arity < 2 || arity > 4 || n_leaves < 2) return -1;
    int d = 0;
    while (n_leaves > 1) {
        n_leaves = (n_leaves + arity - 1) / arity;
        d++;
    }
    return d;
}

int hades252_merkle_level_pad_dev(const void *d_children, size_t n_children, void *d_parents, int arity,
                                  const uint64_t tag_mont[4], int out_idx, const void *d_pad, void *stream) {
    if (arity < 1 || arity > 4) return HADES252_ERR_INVALID_ARG;
    if (n_children == 0) return HADES252_OK;
    const size_t n_parents = (n_children + arity - 1) / arity;
    if (d_children == nullptr || d_parents == nullptr || tag_mont == nullptr || out_idx < 0 || out_idx >= 5 ||
        n_parents > kMaxLaunchRecords || misaligned(d_children) || misaligned(d_parents) || misaligned(d_pad))
        return HADES252_ERR_INVALID_ARG;
    launch_merkle_any(arity, (const uint8_t *)d_children, n_children, (uint8_t *)d_parents, fr_from_u64(tag_mont), out_idx,
                      (const uint8_t *)d_pad, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_merkle_level_dev(const void *d_children, void *d_parents, size_t n_parents, int arity,
                              const uint64_t tag_mont[4], int out_idx, void *stream) {
    if (arity < 1 || arity > 4 || n_parents > kMaxLaunchRecords) return HADES252_ERR_INVALID_ARG;
    return hades252_merkle_level_pad_dev(d_children, n_parents * (size_t)arity, d_parents, arity, tag_mont, out_idx, nullptr,
                                         stream);
}

int hades252_merkle4_level_dev(const void *d_children, void *d_parents, size_t n_parents, const uint64_t tag_mont[4],
                               int out_idx, void *stream) {
    return hades252_merkle_level_dev(d_children, d_parents, n_parents, 4, tag_mont, out_idx, stream);
}

static int sponge_launch(const void *d_scalars, const uint64_t *d_offsets, const uint64_t *d_lengths, size_t n_msgs,
                         size_t fixed_len, const uint64_t capacity_mont[4], int pad_mode, void *d_digests, void *stream,
                         size_t n_scalars, int *d_bad_count, const uint32_t *d_order) {
    if (n_msgs <= kLanesMaxStates) {                    // a few messages: one per wave (any `order` is irrelevant there)
        const bool helped = n_msgs <= kLanesHelpedMaxStates;
        const unsigned per = helped ? kLanesWaves - 1 : kLanesWaves;
        const dim3 grid((unsigned)((n_msgs + per - 1) / per)), block(kLanesWaves * kWave);
        if (helped)
            hipLaunchKernelGGL(k_sponge_lanes<true>, grid, block, 0, (hipStream_t)stream, (const uint8_t *)d_scalars,
                               d_offsets, d_lengths, (uint8_t *)d_digests, n_msgs, fixed_len, fr_from_u64(capacity_mont),
                               pad_mode, n_scalars, d_bad_count);
        else
            hipLaunchKernelGGL(k_sponge_lanes<false>, grid, block, 0, (hipStream_t)stream, (const uint8_t *)d_scalars,
                               d_offsets, d_lengths, (uint8_t *)d_digests, n_msgs, fixed_len, fr_from_u64(capacity_mont),
                               pad_mode, n_scalars, d_bad_count);
        HIP_TRY(hipGetLastError());
        return HADES252_OK;
    }
    if (n_msgs <= kCoopMaxStates && d_order == nullptr) {           // mid-size: five waves per message
        hipLaunchKernelGGL(k_sponge_coop, dim3((unsigned)((n_msgs + kCoopStates - 1) / kCoopStates)), dim3(kCoopThreads), 0,
                           (hipStream_t)stream, (const uint8_t *)d_scalars, d_offsets, d_lengths, (uint8_t *)d_digests,
                           n_msgs, fixed_len, fr_from_u64(capacity_mont), pad_mode, n_scalars, d_bad_count);
        HIP_TRY(hipGetLastError());
        return HADES252_OK;
    }
    hipLaunchKernelGGL(k_sponge, dim3(blocks_for(n_msgs)), dim3(kBlock), lds_for(4), (hipStream_t)stream,
                       (const uint8_t *)d_scalars, d_offsets, d_lengths, (uint8_t *)d_digests, n_msgs, fixed_len,
                       fr_from_u64(capacity_mont), pad_mode, n_scalars, d_bad_count, d_order);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_sponge_hash_dev(const void *d_msgs, size_t n_msgs, size_t msg_len, const uint64_t capacity_mont[4],
                             int pad_mode, void *d_digests, void *stream) {
    if (n_msgs == 0) return HADES252_OK;
    if (d_digests == nullptr || capacity_mont == nullptr || (d_msgs == nullptr && msg_len > 0) ||
        (pad_mode != 0 && pad_mode != 1) || n_msgs > kMaxLaunchRecords || misaligned(d_msgs) || misaligned(d_digests))
        return HADES252_ERR_INVALID_ARG;
    return sponge_launch(d_msgs, nullptr, nullptr, n_msgs, msg_len, capacity_mont, pad_mode, d_digests, stream,
                         n_msgs * msg_len, nullptr, nullptr);
}

size_t hades252_sponge_sort_scratch_bytes(size_t n_msgs) {
    return ((size_t)kSpongeBuckets + n_msgs) * 4 + 16;
}

// d_scratch != NULL (hades252_sponge_sort_scratch_bytes(n_msgs) bytes): the messages are first sorted by block count on
// the device, so that a wave's 64 lanes hash messages of (nearly) the same length -- ragged batches then keep > 90 % of
// the lanes doing useful permutations instead of ~50 %.  Same digests either way.
int hades252_sponge_hash_var_ex_dev(const void *d_scalars, size_t n_scalars, const uint64_t *d_offsets,
                                    const uint64_t *d_lengths, size_t n_msgs, const uint64_t capacity_mont[4], int pad_mode,
                                    void *d_digests, int *d_bad_count, void *d_scratch, size_t scratch_bytes, void *stream) {
    if (n_msgs == 0) return HADES252_OK;
    if (d_digests == nullptr || capacity_mont == nullptr || d_offsets == nullptr || d_lengths == nullptr ||
        (d_scalars == nullptr && n_scalars > 0) || (pad_mode != 0 && pad_mode != 1) || n_msgs > kMaxLaunchRecords ||
        misaligned(d_scalars) || misaligned(d_digests))
        return HADES252_ERR_INVALID_ARG;
    const uint32_t *order = nullptr;
    if (d_scratch != nullptr) {
        if (scratch_bytes < hades252_sponge_sort_scratch_bytes(n_msgs)) return HADES252_ERR_SCRATCH;
        if (misaligned(d_scratch)) return HADES252_ERR_INVALID_ARG;
    }
    // up to kCoopMaxStates messages the batch is one round of blocks either way and takes as long as its longest message:
    // the latency forms (one message per wave / five waves per message) are used and sorting buys nothing
    if (d_scratch != nullptr && n_msgs > kCoopMaxStates) {
        hipStream_t s = (hipStream_t)stream;
        uint32_t *counters = (uint32_t *)d_scratch, *ord = counters + kSpongeBuckets + 4;
        HIP_TRY(hipMemsetAsync(counters, 0, (size_t)kSpongeBuckets * 4, s));
        const unsigned grid = (unsigned)(blocks_for(n_msgs) < 2048 ? blocks_for(n_msgs) : 2048);
        hipLaunchKernelGGL(k_sponge_count, dim3(grid), dim3(kBlock), 0, s, d_lengths, n_msgs, pad_mode, counters);
        hipLaunchKernelGGL(k_sponge_scan, dim3(1), dim3(kSpongeBuckets), 0, s, counters);
        hipLaunchKernelGGL(k_sponge_scatter, dim3(blocks_for(n_msgs)), dim3(kBlock), 0, s, d_lengths, n_msgs, pad_mode,
                           counters, ord);
        HIP_TRY(hipGetLastError());
        order = ord;
    }
    return sponge_launch(d_scalars, d_offsets, d_lengths, n_msgs, 0, capacity_mont, pad_mode, d_digests, stream,
                         n_scalars, d_bad_count, order);
}

int hades252_sponge_hash_var_dev(const void *d_scalars, size_t n_scalars, const uint64_t *d_offsets,
                                 const uint64_t *d_lengths, size_t n_msgs, const uint64_t capacity_mont[4], int pad_mode,
                                 void *d_digests, int *d_bad_count, void *stream) {
    return hades252_sponge_hash_var_ex_dev(d_scalars, n_scalars, d_offsets, d_lengths, n_msgs, capacity_mont, pad_mode,
                                           d_digests, d_bad_count, nullptr, 0, stream);
}

// ---- streaming sponge ---------------------------------------------------------------------------
int hades252_sponge_init_dev(void *d_states, size_t n_states, const uint64_t capacity_mont[4], void *stream) {
    if (n_states == 0) return HADES252_OK;
    if (d_states == nullptr || capacity_mont == nullptr || n_states > kMaxLaunchRecords / 5 || misaligned(d_states))
        return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_sponge_init, dim3(blocks_for(n_states * 5)), dim3(kBlock), 0, (hipStream_t)stream,
                       (uint8_t *)d_states, n_states, fr_from_u64(capacity_mont));
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_sponge_absorb_dev(void *d_states, const void *d_blocks, size_t n_states, int blocks_each, void *stream) {
    if (blocks_each < 0) return HADES252_ERR_INVALID_ARG;
    if (n_states == 0 || blocks_each == 0) return HADES252_OK;
    if (d_states == nullptr || d_blocks == nullptr || n_states > kMaxLaunchRecords || misaligned(d_states) ||
        misaligned(d_blocks))
        return HADES252_ERR_INVALID_ARG;
    if (n_states <= kLanesMaxStates) {
        const bool helped = n_states <= kLanesHelpedMaxStates;
        const unsigned per = helped ? kLanesWaves - 1 : kLanesWaves;
        const dim3 grid((unsigned)((n_states + per - 1) / per)), block(kLanesWaves * kWave);
        if (helped)
            hipLaunchKernelGGL(k_sponge_absorb_lanes<true>, grid, block, 0, (hipStream_t)stream, (uint8_t *)d_states,
                               (const uint8_t *)d_blocks, n_states, blocks_each);
        else
            hipLaunchKernelGGL(k_sponge_absorb_lanes<false>, grid, block, 0, (hipStream_t)stream, (uint8_t *)d_states,
                               (const uint8_t *)d_blocks, n_states, blocks_each);
        HIP_TRY(hipGetLastError());
        return HADES252_OK;
    }
    if (n_states <= kCoopMaxStates) {
        hipLaunchKernelGGL(k_sponge_absorb_coop, dim3((unsigned)((n_states + kCoopStates - 1) / kCoopStates)),
                           dim3(kCoopThreads), 0, (hipStream_t)stream, (uint8_t *)d_states, (const uint8_t *)d_blocks,
                           n_states, blocks_each);
        HIP_TRY(hipGetLastError());
        return HADES252_OK;
    }
    hipLaunchKernelGGL(k_sponge_absorb, dim3(blocks_for(n_states)), dim3(kBlock), lds_for(5), (hipStream_t)stream,
                       (uint8_t *)d_states, (const uint8_t *)d_blocks, n_states, blocks_each);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_sponge_squeeze_dev(const void *d_states, void *d_digests, size_t n_states, int word, void *stream) {
    if (word < 0 || word >= 5) return HADES252_ERR_INVALID_ARG;
    if (n_states == 0) return HADES252_OK;
    if (d_states == nullptr || d_digests == nullptr || n_states > kMaxLaunchRecords / 2 || misaligned(d_states) ||
        misaligned(d_digests))
        return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_sponge_squeeze, dim3(blocks_for(n_states * 2)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const uint8_t *)d_states, (uint8_t *)d_digests, n_states, word);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

size_t hades252_merkle_tree_bytes(size_t n_leaves, int arity) {
    if (hades252_merkle_depth(n_leaves, arity) < 1) return 0;
    size_t total = 0;
    while (n_leaves > 1) {
        n_leaves = (n_leaves + arity - 1) / arity;
        total += n_leaves;                                     // n_1 + n_2 + ... + 1 digests
    }
    return total * 32;
}

size_t hades252_merkle_scratch_bytes(size_t n_leaves, int arity) {
    // two ping-pong buffers: level 1 (n_1 digests) and level 2 (n_2); a single-level tree needs none
    if (hades252_merkle_depth(n_leaves, arity) < 2) return 0;
    const size_t n1 = (n_leaves + arity - 1) / arity, n2 = (n1 + arity - 1) / arity;
    return (n1 + n2) * 32;
}
/* arity-4 form; 0 also for a one-level tree (4 leaves need no scratch) -- hades252_merkle_depth tells valid from invalid */
size_t hades252_merkle4_scratch_bytes(size_t n_leaves) { return hades252_merkle_scratch_bytes(n_leaves, 4); }

// The whole tree over any number of leaves >= 2, arity 2 .. 4.  Levels with more than kCoopMaxStates parents run one
// parent per lane (throughput); full levels of 1 025 .. 16 384 parents run five waves per parent, with arity 2 / 4 and a
// power-of-arity level taking 64 parents per block through several levels inside the CU (k_merkle_coop) as long as
// the next level is still that large; levels of at most kLanesMaxStates parents run one parent per wave
// (k_merkle_lanes: ~51 us per level instead of ~104).  Ragged levels: a child position past the end of level l takes
// pad[l] (device table of depth digests, NULL = zeros).
// tree != NULL: every level is kept (layout of hades252_merkle_build_dev); else ping-pong in buf_a / buf_b.
static int merkle_run(const uint8_t *leaves, size_t n_leaves, int arity, uint8_t *tree, uint8_t *buf_a, uint8_t *buf_b,
                      uint8_t *root, const Fr &tag, int out_idx, const uint8_t *pad, hipStream_t s) {
    const uint8_t *src = leaves;
    size_t n = n_leaves, off = 0;
    bool to_a = true;
    int level = 0;
    while (n > 1) {
        const size_t parents = (n + arity - 1) / arity;
        const uint8_t *pad_l = pad != nullptr ? pad + (size_t)level * 32 : nullptr;
        uint8_t *dst_pp = to_a ? buf_a : buf_b;
        int fused = 1;
        if ((arity == 2 || arity == 4) && parents > kLanesMaxStates && parents <= kCoopMaxStates && log_arity(n, arity) > 0) {
            // fuse while the level after the last fused one is still too large for one-parent-per-wave
            const int max_fused = log_arity(kCoopStates, arity) + 1;                   // 64 parents -> 1 digest
            size_t sz = parents;
            while (fused < max_fused && sz / arity > kLanesMaxStates) {
                sz /= arity;
                fused++;
            }
        }
        if (fused > 1) {
            size_t last_n = parents, span = 0;                    // digests in the last level run; bytes before it
            for (int j = 1; j < fused; j++) {
                span += last_n * 32;
                last_n /= arity;
            }
            uint8_t *out_all = tree != nullptr ? tree + off : nullptr;
            // with a tree every level goes through out_all; the two pointers are __restrict__ in the kernel and must
            // never name the same bytes
            uint8_t *out_last = tree != nullptr ? nullptr : dst_pp;
            launch_merkle_coop(arity, src, out_all, out_last, parents, tag, out_idx, fused, s);
            HIP_TRY(hipGetLastError());
            src = tree != nullptr ? tree + off + span : dst_pp;
            off += span + last_n * 32;
            n = last_n;
            level += fused;
        } else {
            uint8_t *dst = tree != nullptr ? tree + off : (parents == 1 ? root : dst_pp);
            launch_merkle_any(arity, src, n, dst, tag, out_idx, pad_l, s);
            HIP_TRY(hipGetLastError());
            off += parents * 32;
            src = dst;
            n = parents;
            level++;
        }
        to_a = !to_a;
    }
    return HADES252_OK;
}

int hades252_merkle_root_pad_dev(const void *d_leaves, size_t n_leaves, int arity, void *d_scratch, size_t scratch_bytes,
                                 const uint64_t tag_mont[4], int out_idx, const void *d_pad, void *d_root, void *stream) {
    if (d_leaves == nullptr || d_root == nullptr || tag_mont == nullptr || hades252_merkle_depth(n_leaves, arity) < 1 ||
        out_idx < 0 || out_idx >= 5 || misaligned(d_leaves) || misaligned(d_root) || misaligned(d_pad))
        return HADES252_ERR_INVALID_ARG;
    const size_t need = hades252_merkle_scratch_bytes(n_leaves, arity);
    if (need > 0 && (d_scratch == nullptr || scratch_bytes < need)) return HADES252_ERR_SCRATCH;
    if (need > 0 && misaligned(d_scratch)) return HADES252_ERR_INVALID_ARG;
    uint8_t *buf_a = (uint8_t *)d_scratch;
    uint8_t *buf_b = need > 0 ? buf_a + ((n_leaves + arity - 1) / arity) * 32 : nullptr;
    return merkle_run((const uint8_t *)d_leaves, n_leaves, arity, nullptr, buf_a, buf_b, (uint8_t *)d_root,
                      fr_from_u64(tag_mont), out_idx, (const uint8_t *)d_pad, (hipStream_t)stream);
}

int hades252_merkle_root_dev(const void *d_leaves, size_t n_leaves, int arity, void *d_scratch, size_t scratch_bytes,
                             const uint64_t tag_mont[4], int out_idx, void *d_root, void *stream) {
    return hades252_merkle_root_pad_dev(d_leaves, n_leaves, arity, d_scratch, scratch_bytes, tag_mont, out_idx, nullptr,
                                        d_root, stream);
}

int hades252_merkle4_root_dev(const void *d_leaves, size_t n_leaves, void *d_scratch, size_t scratch_bytes,
                              const uint64_t tag_mont[4], int out_idx, void *d_root, void *stream) {
    return hades252_merkle_root_dev(d_leaves, n_leaves, 4, d_scratch, scratch_bytes, tag_mont, out_idx, d_root, stream);
}

int hades252_merkle_build_pad_dev(const void *d_leaves, size_t n_leaves, int arity, const uint64_t tag_mont[4], int out_idx,
                                  const void *d_pad, void *d_tree, void *stream) {
    if (d_leaves == nullptr || d_tree == nullptr || tag_mont == nullptr || hades252_merkle_depth(n_leaves, arity) < 1 ||
        out_idx < 0 || out_idx >= 5 || misaligned(d_leaves) || misaligned(d_tree) || misaligned(d_pad))
        return HADES252_ERR_INVALID_ARG;
    uint8_t *tree = (uint8_t *)d_tree;
    uint8_t *root = tree + hades252_merkle_tree_bytes(n_leaves, arity) - 32;
    return merkle_run((const uint8_t *)d_leaves, n_leaves, arity, tree, nullptr, nullptr, root, fr_from_u64(tag_mont),
                      out_idx, (const uint8_t *)d_pad, (hipStream_t)stream);
}

int hades252_merkle_build_dev(const void *d_leaves, size_t n_leaves, int arity, const uint64_t tag_mont[4], int out_idx,
                              void *d_tree, void *stream) {
    return hades252_merkle_build_pad_dev(d_leaves, n_leaves, arity, tag_mont, out_idx, nullptr, d_tree, stream);
}

// Incremental update: the caller has overwritten the leaves d_leaves[d_indices[q]], q < n_updates; their ancestors in
// d_tree (built by hades252_merkle_build[_pad]_dev with the same parameters) are recomputed bottom-up, one launch per
// level: depth x min(n_updates, n_level) permutations instead of the whole tree.  A level with no more parents than
// updates is simply recomputed whole.
int hades252_merkle_update_dev(const void *d_leaves, void *d_tree, size_t n_leaves, int arity, const uint64_t tag_mont[4],
                               int out_idx, const void *d_pad, const uint64_t *d_indices, size_t n_updates, void *stream) {
    const int depth = hades252_merkle_depth(n_leaves, arity);
    if (depth < 1 || tag_mont == nullptr || out_idx < 0 || out_idx >= 5) return HADES252_ERR_INVALID_ARG;
    if (n_updates == 0) return HADES252_OK;
    if (d_leaves == nullptr || d_tree == nullptr || d_indices == nullptr || n_updates > kMaxLaunchRecords ||
        misaligned(d_leaves) || misaligned(d_tree) || misaligned(d_pad))
        return HADES252_ERR_INVALID_ARG;
    const Fr tag = fr_from_u64(tag_mont);
    hipStream_t s = (hipStream_t)stream;
    const uint8_t *src = (const uint8_t *)d_leaves, *pad = (const uint8_t *)d_pad;
    uint8_t *tree = (uint8_t *)d_tree;
    size_t n = n_leaves, off = 0;
    uint64_t span = 1;
    for (int l = 0; l < depth; l++) {
        const size_t parents = (n + arity - 1) / arity;
        const uint8_t *pad_l = pad != nullptr ? pad + (size_t)l * 32 : nullptr;
        uint8_t *dst = tree + off;
        span *= (uint64_t)arity;
        if (parents <= n_updates)
            launch_merkle_any(arity, src, n, dst, tag, out_idx, pad_l, s);
        else
            launch_merkle_update(arity, src, n, dst, d_indices, n_updates, n_leaves, span, tag, out_idx, pad_l, s);
        HIP_TRY(hipGetLastError());
        off += parents * 32;
        src = dst;
        n = parents;
    }
    return HADES252_OK;
}

// pad[0] = e0 (the digest standing for a missing leaf), pad[l+1] = perm([tag, pad[l] x arity, 0 ..])[out_idx]: the
// roots of empty subtrees, level by level -- the usual padding table of an append-only tree
int hades252_merkle_empty_digests_dev(int arity, int depth, const uint64_t e0_mont[4], const uint64_t tag_mont[4],
                                      int out_idx, void *d_pad, void *stream) {
    if (arity < 2 || arity > 4 || depth < 1 || depth > 64 || e0_mont == nullptr || tag_mont == nullptr || d_pad == nullptr ||
        out_idx < 0 || out_idx >= 5 || misaligned(d_pad))
        return HADES252_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    uint8_t *pad = (uint8_t *)d_pad;
    HIP_TRY(hipMemcpyAsync(pad, e0_mont, 32, hipMemcpyHostToDevice, s));
    const Fr tag = fr_from_u64(tag_mont);
    for (int l = 0; l + 1 < depth; l++) {
        // zero children + padding = a parent whose arity children are all pad[l]
        launch_merkle_lanes(arity, pad, 0, pad + (size_t)(l + 1) * 32, 1, tag, out_idx, pad + (size_t)l * 32, s);
        HIP_TRY(hipGetLastError());
    }
    return HADES252_OK;
}

int hades252_merkle_open_pad_dev(const void *d_leaves, const void *d_tree, size_t n_leaves, int arity,
                                 const uint64_t *d_indices, size_t n_queries, const void *d_pad, void *d_paths,
                                 void *stream) {
    const int depth = hades252_merkle_depth(n_leaves, arity);
    if (depth < 1) return HADES252_ERR_INVALID_ARG;
    if (n_queries == 0) return HADES252_OK;
    if (d_leaves == nullptr || d_tree == nullptr || d_indices == nullptr || d_paths == nullptr || misaligned(d_leaves) ||
        misaligned(d_tree) || misaligned(d_paths) || misaligned(d_pad))
        return HADES252_ERR_INVALID_ARG;
    const size_t threads = n_queries * (size_t)depth * (arity - 1) * 2;
    if (threads > kMaxLaunchRecords) return HADES252_ERR_INVALID_ARG;
#define HADES_LAUNCH_OPEN(A)                                                                                            \
    hipLaunchKernelGGL(k_merkle_open<A>, dim3(blocks_for(threads)), dim3(kBlock), 0, (hipStream_t)stream,             \
                       (const uint8_t *)d_leaves, (const uint8_t *)d_tree, n_leaves, depth, d_indices, n_queries,     \
                       (uint8_t *)d_paths, (const uint8_t *)d_pad)
    switch (arity) {
        case 2: HADES_LAUNCH_OPEN(2); break;
        case 3: HADES_LAUNCH_OPEN(3); break;
        default: HADES_LAUNCH_OPEN(4); break;
    }
#undef HADES_LAUNCH_OPEN
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_merkle_open_dev(const void *d_leaves, const void *d_tree, size_t n_leaves, int arity,
                             const uint64_t *d_indices, size_t n_queries, void *d_paths, void *stream) {
    return hades252_merkle_open_pad_dev(d_leaves, d_tree, n_leaves, arity, d_indices, n_queries, nullptr, d_paths, stream);
}

// Batched path verification: root_t = the root recomputed from leaf t (d_leaves[t], 32 B), its index and its opening
// d_paths[t][l][s] (the layout hades252_merkle_open_dev writes).  One query per lane, `depth` permutations each.
int hades252_merkle_verify_dev(const void *d_leaves, const uint64_t *d_indices, const void *d_paths, size_t n_queries,
                               int depth, int arity, const uint64_t tag_mont[4], int out_idx, void *d_roots, void *stream) {
    if (arity < 1 || arity > 4 || depth < 1 || depth > 64 || out_idx < 0 || out_idx >= 5 || tag_mont == nullptr)
        return HADES252_ERR_INVALID_ARG;
    if (n_queries == 0) return HADES252_OK;
    if (d_leaves == nullptr || d_indices == nullptr || (d_paths == nullptr && arity > 1) || d_roots == nullptr ||
        n_queries > kMaxLaunchRecords || misaligned(d_leaves) || misaligned(d_paths) || misaligned(d_roots))
        return HADES252_ERR_INVALID_ARG;
    const Fr tag = fr_from_u64(tag_mont);
    const bool lanes = n_queries <= kLanesMaxStates, helped = n_queries <= kLanesHelpedMaxStates;
    const unsigned per = helped ? kLanesWaves - 1 : kLanesWaves;
    const dim3 lgrid((unsigned)((n_queries + per - 1) / per)), lblock(kLanesWaves * kWave);
#define HADES_VERIFY_ARGS                                                                                            \
    (const uint8_t *)d_leaves, d_indices, (const uint8_t *)d_paths, n_queries, depth, tag, out_idx, (uint8_t *)d_roots
#define HADES_LAUNCH_VERIFY(A)                                                                                       \
    do {                                                                                                             \
        if (!lanes && n_queries <= kCoopMaxStates)                                                                   \
            hipLaunchKernelGGL(k_merkle_verify_coop<A>, dim3((unsigned)((n_queries + kCoopStates - 1) / kCoopStates)), \
                               dim3(kCoopThreads), 0, (hipStream_t)stream, HADES_VERIFY_ARGS);                       \
        else if (!lanes)                                                                                             \
            hipLaunchKernelGGL(k_merkle_verify<A>, dim3(blocks_for(n_queries)), dim3(kBlock), lds_for(1),           \
                               (hipStream_t)stream, HADES_VERIFY_ARGS);                                              \
        else if (helped)                                                                                             \
            hipLaunchKernelGGL((k_merkle_verify_lanes<A, true>), lgrid, lblock, 0, (hipStream_t)stream,             \
                               HADES_VERIFY_ARGS);                                                                   \
        else                                                                                                         \
            hipLaunchKernelGGL((k_merkle_verify_lanes<A, false>), lgrid, lblock, 0, (hipStream_t)stream,            \
                               HADES_VERIFY_ARGS);                                                                   \
    } while (0)
    switch (arity) {
        case 1: HADES_LAUNCH_VERIFY(1); break;
        case 2: HADES_LAUNCH_VERIFY(2); break;
        case 3: HADES_LAUNCH_VERIFY(3); break;
        default: HADES_LAUNCH_VERIFY(4); break;
    }
#undef HADES_LAUNCH_VERIFY
#undef HADES_VERIFY_ARGS
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

// Forest: n_trees independent trees of leaves_per_tree = arity^k leaves each, leaves contiguous tree after tree.  All
// trees have the same shape, so level l of the whole forest is ONE launch over n_trees * arity^(k-l) parents (a parent
// never straddles two trees); the roots come out contiguous.  Scratch: two ping-pong level buffers.
size_t hades252_merkle_forest_scratch_bytes(size_t n_trees, size_t leaves_per_tree, int arity) {
    const int k = log_arity(leaves_per_tree, arity);
    if (k < 1 || n_trees == 0) return 0;
    if (k == 1) return 0;
    const size_t n1 = n_trees * (leaves_per_tree / arity);
    return (n1 + n1 / arity) * 32;
}

int hades252_merkle_forest_dev(const void *d_leaves, size_t n_trees, size_t leaves_per_tree, int arity, void *d_scratch,
                               size_t scratch_bytes, const uint64_t tag_mont[4], int out_idx, void *d_roots, void *stream) {
    const int k = log_arity(leaves_per_tree, arity);
    if (k < 1 || tag_mont == nullptr || out_idx < 0 || out_idx >= 5) return HADES252_ERR_INVALID_ARG;
    if (n_trees == 0) return HADES252_OK;
    if (d_leaves == nullptr || d_roots == nullptr || misaligned(d_leaves) || misaligned(d_roots) ||
        n_trees > kMaxLaunchRecords / leaves_per_tree)
        return HADES252_ERR_INVALID_ARG;
    const size_t need = hades252_merkle_forest_scratch_bytes(n_trees, leaves_per_tree, arity);
    if (need > 0 && (d_scratch == nullptr || scratch_bytes < need)) return HADES252_ERR_SCRATCH;
    if (need > 0 && misaligned(d_scratch)) return HADES252_ERR_INVALID_ARG;
    const Fr tag = fr_from_u64(tag_mont);
    uint8_t *buf_a = (uint8_t *)d_scratch;
    uint8_t *buf_b = need > 0 ? buf_a + n_trees * (leaves_per_tree / arity) * 32 : nullptr;
    const uint8_t *src = (const uint8_t *)d_leaves;
    size_t n = n_trees * leaves_per_tree;
    bool to_a = true;
    for (int l = 0; l < k; l++) {
        uint8_t *dst = l == k - 1 ? (uint8_t *)d_roots : (to_a ? buf_a : buf_b);
        launch_merkle_any(arity, src, n, dst, tag, out_idx, nullptr, (hipStream_t)stream);
        HIP_TRY(hipGetLastError());
        src = dst;
        n /= arity;
        to_a = !to_a;
    }
    return HADES252_OK;
}

// ---- synthetic / digest ------------------------------------------------------------------------
int hades252_gen_b_dev(void *d_scalars, uint64_t first_elem, size_t n_elems, uint64_t seed, void *stream) {
    if (n_elems == 0) return HADES252_OK;
    if (d_scalars == nullptr) return HADES252_ERR_INVALID_ARG;
    size_t n_limbs = n_elems * 4;
    size_t want = (n_limbs + kBlock - 1) / kBlock;
    unsigned grid = (unsigned)(want < 65536 ? want : 65536);
    hipLaunchKernelGGL(k_gen_b, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (uint64_t *)d_scalars, first_elem,
                       n_limbs, seed);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_gen_a_dev(void *d_scalars, uint64_t first_elem, size_t n_elems, void *stream) {
    if (n_elems == 0) return HADES252_OK;
    if (d_scalars == nullptr || n_elems > kMaxLaunchRecords || misaligned(d_scalars)) return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_gen_a, dim3(blocks_for(n_elems)), dim3(kBlock), lds_for(1), (hipStream_t)stream,
                       (uint8_t *)d_scalars, first_elem, n_elems);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_digest_dev(const void *d_words, uint64_t first_index, size_t n_u64, void *d_out4, void *stream) {
    if (d_out4 == nullptr || (d_words == nullptr && n_u64 > 0)) return HADES252_ERR_INVALID_ARG;
    HIP_TRY(hipMemsetAsync(d_out4, 0, 32, (hipStream_t)stream));
    if (n_u64 == 0) return HADES252_OK;
    size_t want = (n_u64 + kBlock - 1) / kBlock;
    unsigned grid = (unsigned)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(k_digest, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (const uint64_t *)d_words,
                       first_index, n_u64, (unsigned long long *)d_out4);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

}  // extern "C"
