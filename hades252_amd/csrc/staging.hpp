// staging.hpp -- coalesced AoS <-> per-lane record movement through LDS.
//
// A batch is an array of 160-byte records (5 x BlsScalar, reference `&mut [BlsScalar]` with
// len == WIDTH, src/strategies.rs:140).  One lane owns one record, so a wave owns 64 records =
// 10 KiB contiguous in HBM.  Lanes never touch HBM with a 160-byte stride: the wave moves its
// 10 KiB as 10 fully coalesced 1-KiB `global_load_dwordx4` / `global_store_dwordx4`
// wave-instructions and redistributes through a wave-private LDS slab.  Records are padded by
// 16 B in LDS so that the per-lane `ds_read_b128` / `ds_write_b128` are bank-conflict free
// (e.g. 176 B = 44 dwords: 16 consecutive lanes start on 16 distinct 4-bank slots).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fr32.hpp"

namespace hades {

constexpr int kWave = 64;
// padded LDS stride of a record of NW scalars: NW*32 + 16 bytes (conflict-free for NW = 1, 4, 5)
__host__ __device__ constexpr int lds_rec_bytes(int nw) { return nw * 32 + 16; }
__host__ __device__ constexpr int lds_wave_bytes(int nw) { return kWave * lds_rec_bytes(nw); }

// Load the records [rec0, rec0+64) (clipped to n_recs) of `base` into per-lane state.
// `slab` is this wave's LDS slab (lds_wave_bytes(NW)).  Lanes past the end get zeros.
// n_scalars = number of 32-byte scalars the buffer holds (a ragged last record reads zeros past it).
template <int NW>
__device__ __forceinline__ void wave_load_scalars(const uint8_t *base, size_t rec0, size_t n_scalars,
                                                  uint8_t *slab, Fr (&st)[NW]) {
    constexpr int kLdsRecBytes = lds_rec_bytes(NW);
    const int lane = threadIdx.x & (kWave - 1);
    const size_t rec_bytes = (size_t)NW * 32;
    const size_t total_chunks = n_scalars * 2;
    const uint4 *g = reinterpret_cast<const uint4 *>(base + rec0 * rec_bytes);
    const size_t chunk0 = rec0 * (size_t)(2 * NW);
#pragma unroll
    for (int k = 0; k < 2 * NW; k++) {
        int c = k * kWave + lane;                 // chunk index inside the wave's slab
        uint4 v = make_uint4(0, 0, 0, 0);
        if (chunk0 + c < total_chunks) v = g[c];
        int rec = c / (2 * NW), part = c - rec * (2 * NW);
        *reinterpret_cast<uint4 *>(slab + rec * kLdsRecBytes + part * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint4 *p = reinterpret_cast<const uint4 *>(slab + lane * kLdsRecBytes + w * 32);
        uint4 lo = p[0], hi = p[1];
        st[w].l[0] = lo.x; st[w].l[1] = lo.y; st[w].l[2] = lo.z; st[w].l[3] = lo.w;
        st[w].l[4] = hi.x; st[w].l[5] = hi.y; st[w].l[6] = hi.z; st[w].l[7] = hi.w;
    }
    __syncthreads();
}

template <int NW>
__device__ __forceinline__ void wave_load_records(const uint8_t *base, size_t rec0, size_t n_recs,
                                                  uint8_t *slab, Fr (&st)[NW]) {
    wave_load_scalars<NW>(base, rec0, n_recs * (size_t)NW, slab, st);
}

// Word w of this lane's record -> the wave's slab (no synchronisation; see slab_flush).
template <int NW>
__device__ __forceinline__ void slab_put(uint8_t *slab, int w, const Fr &v) {
    constexpr int kLdsRecBytes = lds_rec_bytes(NW);
    const int lane = threadIdx.x & (kWave - 1);
    uint4 *p = reinterpret_cast<uint4 *>(slab + lane * kLdsRecBytes + w * 32);
    p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    p[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// Slab (filled with slab_put by every lane) -> records [rec0, rec0+64) (clipped to n_recs).
// Block-wide barriers: every wave of the block must call it the same number of times.
template <int NW>
__device__ __forceinline__ void slab_flush(uint8_t *base, size_t rec0, size_t n_recs, uint8_t *slab) {
    constexpr int kLdsRecBytes = lds_rec_bytes(NW);
    const int lane = threadIdx.x & (kWave - 1);
    __syncthreads();
    const size_t rec_bytes = (size_t)NW * 32;
    const size_t total_chunks = n_recs * (size_t)(2 * NW);
    uint4 *g = reinterpret_cast<uint4 *>(base + rec0 * rec_bytes);
    const size_t chunk0 = rec0 * (size_t)(2 * NW);
#pragma unroll
    for (int k = 0; k < 2 * NW; k++) {
        int c = k * kWave + lane;
        int rec = c / (2 * NW), part = c - rec * (2 * NW);
        uint4 v = *reinterpret_cast<const uint4 *>(slab + rec * kLdsRecBytes + part * 16);
        if (chunk0 + c < total_chunks) g[c] = v;
    }
    __syncthreads();
}

// Store per-lane state to records [rec0, rec0+64) (clipped to n_recs).
// `base` may alias the buffer the records were loaded from (in-place kernels): a wave has loaded all
// of its records before it stores any, and waves own disjoint record ranges -- hence no __restrict__.
template <int NW>
__device__ __forceinline__ void wave_store_records(uint8_t *base, size_t rec0, size_t n_recs,
                                                   uint8_t *slab, const Fr (&st)[NW]) {
    constexpr int kLdsRecBytes = lds_rec_bytes(NW);
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint4 *p = reinterpret_cast<uint4 *>(slab + lane * kLdsRecBytes + w * 32);
        p[0] = make_uint4(st[w].l[0], st[w].l[1], st[w].l[2], st[w].l[3]);
        p[1] = make_uint4(st[w].l[4], st[w].l[5], st[w].l[6], st[w].l[7]);
    }
    __syncthreads();
    const size_t rec_bytes = (size_t)NW * 32;
    const size_t total_chunks = n_recs * (size_t)(2 * NW);
    uint4 *g = reinterpret_cast<uint4 *>(base + rec0 * rec_bytes);
    const size_t chunk0 = rec0 * (size_t)(2 * NW);
#pragma unroll
    for (int k = 0; k < 2 * NW; k++) {
        int c = k * kWave + lane;
        int rec = c / (2 * NW), part = c - rec * (2 * NW);
        uint4 v = *reinterpret_cast<const uint4 *>(slab + rec * kLdsRecBytes + part * 16);
        if (chunk0 + c < total_chunks) g[c] = v;
    }
    __syncthreads();
}

}  // namespace hades
