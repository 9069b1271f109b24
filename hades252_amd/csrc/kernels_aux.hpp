// kernels_aux.hpp -- synthetic input generators and the additive digest (measurement and test support)
// Part of the single translation unit hades252.hip (included there after device_tables.hpp); not a stand-alone header.
#pragma once

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

// one scalar passed BY VALUE into device memory (hades252_merkle_empty_digests_dev: a captured graph must not keep a
// pointer into the caller's host memory)
__global__ void k_store_fr(uint32_t *__restrict__ out, Fr v) {
    if (threadIdx.x < 8) out[threadIdx.x] = v.l[threadIdx.x];
}
