// device_tables.hpp -- constant tables of every kernel, 32-byte word access, lane / wave helpers
// Part of the single translation unit hades252.hip (included there after the arithmetic headers); not a stand-alone header.
#pragma once

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
// the same arithmetic under the throughput kernel's schedule: one state per 16-lane row (rows_perm)
__device__ const LanesTables d_rows = {HADES_ROWS_ROUND_INIT, HADES_FAST_FINAL_F, HADES_FAST_MDS_SMALL, HADES_P29,
                                        HADES_P29, HADES_NEG_PINV29};
// ... and K_r of its 59 partial rounds as the per-lane constants of lane_lin (the linear map by the row)
__device__ const uint32_t d_rows_klin[59][kNL][16] = HADES_ROWS_KLIN_INIT;
// witness and trace kernels (true-form schedule, hades252_amd/_derive.py::witness_schedule): round constants as Rp-form addends minus p
// (c[r]: five words of nine signed-digit limbs; c[67] = zeros), their images under the linear-layer map (ck[r], words 0..3,
// partial rounds), and the two linear maps: in-memory limbs -> Rp form, and Y -> Y lam 2^29 (the ONE constant product of
// the linear layer).  __constant__, not const: see d_wire_from_lin below.
struct WitnessTables {
    int32_t c[68][48];
    int32_t ck[67][48];
    int32_t in_lin[kLinRow];
    int32_t k_lin[kLinRow];
};
__constant__ WitnessTables d_wit = {HADES_WITNESS_C_INIT, HADES_WITNESS_CK_INIT, HADES_WITNESS_IN_LIN, HADES_WITNESS_K_LIN};
// generic radix-2^29 field ops (hades252_fr_op_dev)
__device__ const int32_t d_rp_mod_p[16] = HADES_RP_MOD_P29;
// per-operation kernels on the same path (hades252_amd/_derive.py)
__device__ const int32_t d_op_k_lin[kLinRow] = HADES_OP_K_LIN;                        // linear maps (mont_lin)
__device__ const int32_t d_op_w_lin[kLinRow] = HADES_OP_W_LIN;
__device__ const int32_t d_op_w_full_lin[kLinRow] = HADES_OP_W_FULL_LIN;
// wire format (from_bytes / to_bytes) on the same path
// a -> a * 2^256 as a linear map (mont_lin).  __constant__ and NOT const on purpose: a const table is folded into 81 literal
// s_mov_b32 per scalar (one per multiply-add, on the CU's single scalar unit, plus hazard s_nops around the multiply-add's
// SGPR carry-out); a plain __device__ global is read with VECTOR loads; in the constant address space, its value unknown
// to the compiler, it arrives by one s_load_dwordx8 + s_load_dword per column like the round tables do.
__constant__ int32_t d_wire_from_lin[kLinRow] = HADES_WIRE_FROM_LIN;
__device__ const int32_t d_rp2_over_r[16] = HADES_RP2_OVER_R29;

// ---- access to 32-byte words, small helpers shared by the kernel headers ------------------------------------
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
