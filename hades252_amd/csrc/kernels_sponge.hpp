// kernels_sponge.hpp -- sponge hashing: one message per lane / per wave / per five waves, the counting sort of ragged batches, streaming absorb
// Part of the single translation unit hades252.hip (included there after kernels_merkle.hpp); not a stand-alone header.
#pragma once

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
        __syncthreads();      // the last round's exchange buffer is the next permutation's first: everyone has read it
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
        __syncthreads();      // (as in k_sponge_coop)
    }
    if (live) store_word(mine, st);
}

// ---- 1 025 .. 4 096 chains: four per wave, one per 16-lane row (hades_lanes.hpp::rows_perm) -----------------------------
// No block barrier: every wave runs to the maximum block count of ITS four messages.
__global__ void __launch_bounds__(kRowsWaves *kWave) k_sponge_rows(const uint8_t *__restrict__ scalars,
                                                                   const uint64_t *__restrict__ offsets,
                                                                   const uint64_t *__restrict__ lengths,
                                                                   uint8_t *__restrict__ digests, size_t n_msgs, size_t fixed_len,
                                                                   Fr capacity, int pad_mode, size_t n_scalars, int *bad_count) {
    __shared__ RowsLds L[kRowsWaves];
    const int wave = threadIdx.x >> 6;
    if (((size_t)blockIdx.x * kRowsWaves + wave) * kRowsPerWave >= n_msgs) return;
    size_t me;
    int word;
    const bool mine = rows_role(n_msgs, me, word);
    SpongeGeom g = {0, 0, 0, false};
    if (mine) g = sponge_geom(offsets, lengths, me, fixed_len, n_scalars, pad_mode);
    if (g.bad && word == 0 && bad_count != nullptr) atomicAdd(bad_count, 1);
    const uint64_t trips = wave_max_u64(g.blocks);
    auto block_word = [&](uint64_t t) {                              // lane 5 s + 1 + k: scalar 4t + k of message s
        Fr v = zero_word();
        if (mine && word >= 1) {
            const uint64_t idx = 4 * t + (uint64_t)(word - 1);
            if (idx < g.len)
                v = load_word(scalars + (g.off + idx) * 32);
            else if (pad_mode == 1 && idx == g.len)
                v = one_mont_word();
        }
        return v;
    };
    Fr st = mine && word == 0 ? capacity : zero_word();
    Fr dig = zero_word(), nxt = block_word(0);
#pragma unroll 1
    for (uint64_t t = 0; t < trips; t++) {
        if (word >= 1) st = fr_add(st, nxt);
        nxt = block_word(t + 1);
        st = rows_perm(&d_rows, d_rows_klin, L[wave], st);
        if (t + 1 == g.blocks) dig = st;
    }
    if (mine && word == 1) store_word(digests + me * 32, dig);
}

__global__ void __launch_bounds__(kRowsWaves *kWave) k_sponge_absorb_rows(uint8_t *states, const uint8_t *__restrict__ blocks,
                                                                          size_t n, int blocks_each) {
    __shared__ RowsLds L[kRowsWaves];
    const int wave = threadIdx.x >> 6;
    if (((size_t)blockIdx.x * kRowsWaves + wave) * kRowsPerWave >= n) return;
    size_t me;
    int word;
    const bool mine = rows_role(n, me, word);
    uint8_t *p = states + (mine ? me : 0) * 160 + word * 32;
    const uint8_t *blk = blocks + (mine ? me : 0) * (size_t)blocks_each * 128 + (word >= 1 ? word - 1 : 0) * 32;
    Fr st = mine ? load_word(p) : zero_word();
    Fr nxt = mine ? load_word(blk) : zero_word();
#pragma unroll 1
    for (int t = 0; t < blocks_each; t++) {
        if (word >= 1) st = fr_add(st, nxt);
        if (mine && t + 1 < blocks_each) nxt = load_word(blk + (size_t)(t + 1) * 128);
        st = rows_perm(&d_rows, d_rows_klin, L[wave], st);
    }
    if (mine) store_word(p, st);
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
