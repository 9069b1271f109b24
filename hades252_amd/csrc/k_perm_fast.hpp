// k_perm_fast.hpp -- the shipped hot path as a kernel: one permutation per lane, scale-tracked formulation
// (hades_fast.hpp), AoS records moved through the wave's LDS slab (staging.hpp).  Kept in a file of its own so that
// the committed profiles of THIS kernel (profiles/hbm_traffic.json) are keyed to exactly the sources that determine
// it (hades252_amd/build.py::perm_fast_hash).  Include after hades_constants.inc.
#pragma once
#include "hades_fast.hpp"
#include "staging.hpp"

namespace hades {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

// 67 round records {A[5][9], K[9]} + the final un-scaling factor, 17 KiB.  Every access is wave-uniform: hipcc emits
// s_load_dwordx8/x16 through the scalar cache and the limbs arrive in SGPRs, consumed directly as v_mad_i64_i32
// operands (no VGPR, no LDS bandwidth, no VALU slot: DESIGN.md section 2; A/B against an LDS copy: profiles/r3/).
// (__constant__ and not const: the compiler must not fold the final linear map's 81 multipliers into literal s_mov_b32,
// one per multiply-add -- device_tables.hpp, d_wire_from_lin)
__constant__ FastTables d_fast = {HADES_FAST_ROUND_INIT, HADES_FAST_FINAL_F, HADES_FAST_LIN_INIT, HADES_FAST_FINAL_LIN};

template <int NW>
__device__ __forceinline__ uint8_t *wave_slab(uint8_t *lds) {
    return lds + (threadIdx.x / kWave) * lds_wave_bytes(NW);
}

// __launch_bounds__(256, 4): at least 4 waves per SIMD, i.e. at most 128 VGPRs; the kernel needs 94 and no scratch
// (tests/test_codegen_guard.py), so registers would admit 5.  Residency is 3 waves per SIMD in practice: the 45 KB
// staging slab of a block admits 3 blocks per CU.  The sweep in profiles/r2/residency.txt (2 ... 6 waves per SIMD,
// all within +-0.6 %) shows that nothing more is needed: one wave per SIMD already issues this dependent
// multiply-add chain at the pipe's cadence.
#ifndef HADES_FAST_MINW
#define HADES_FAST_MINW 4
#endif
// `out` may be `in` (the in-place form of Strategy::perm: a wave has loaded all of its records before it stores any,
// and waves own disjoint record ranges -- hence no __restrict__), or a different buffer: the host-pointer path reads a
// chunk from device memory and stores the results straight into the caller's page-locked host memory.
__global__ void __launch_bounds__(kBlock, HADES_FAST_MINW) k_perm_fast(const uint8_t *in, uint8_t *out, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr st[5];
    wave_load_records<5>(in, rec0, n, slab, st);
    Fr res[5];
    fast_perm<5>(&d_fast, st, res, 0);
    wave_store_records<5>(out, rec0, n, slab, res);
}

}  // namespace hades
