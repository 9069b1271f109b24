// kernels_merkle.hpp -- Merkle levels, fused levels, openings, path verification and incremental update, in all three kernel forms
// Part of the single translation unit hades252.hip (included there after kernels_perm.hpp); not a stand-alone header.
#pragma once

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

// One Merkle level, four parents per wave (one per row: rows_perm); ragged levels and `pad` as above.
template <int ARITY>
__global__ void __launch_bounds__(kRowsWaves *kWave) k_merkle_rows(const uint8_t *__restrict__ children, size_t n_children,
                                                                   uint8_t *__restrict__ parents, size_t n_parents, Fr tag,
                                                                   int out_idx, const uint8_t *__restrict__ pad) {
    __shared__ RowsLds L[kRowsWaves];
    const int wave = threadIdx.x >> 6;
    if (((size_t)blockIdx.x * kRowsWaves + wave) * kRowsPerWave >= n_parents) return;
    size_t rec;
    int word;
    const bool mine = rows_role(n_parents, rec, word);
    Fr in = zero_word();
    if (mine && word == 0) in = tag;
    if (mine && word >= 1 && word <= ARITY) {
        const size_t c = rec * ARITY + (word - 1);
        in = c < n_children ? load_word(children + c * 32) : load_pad(pad);
    }
    const Fr out = rows_perm(&d_rows, d_rows_klin, L[wave], in);
    if (mine && word == out_idx) store_word(parents + rec * 32, out);
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

// the same two chains with four queries per wave, one per 16-lane row (1 025 .. 4 096 queries)
template <int ARITY>
__global__ void __launch_bounds__(kRowsWaves *kWave) k_merkle_update_rows(const uint8_t *__restrict__ children,
                                                                          size_t n_children, uint8_t *__restrict__ parents,
                                                                          const uint64_t *__restrict__ indices, size_t n_updates,
                                                                          size_t n_leaves, uint64_t span, Fr tag, int out_idx,
                                                                          const uint8_t *__restrict__ pad) {
    __shared__ RowsLds L[kRowsWaves];
    const int wave = threadIdx.x >> 6;
    if (((size_t)blockIdx.x * kRowsWaves + wave) * kRowsPerWave >= n_updates) return;
    size_t q;
    int word;
    bool mine = rows_role(n_updates, q, word);
    const UpdateWanted wanted{indices, n_leaves, span};
    mine = mine && wanted(q);
    const size_t parent = mine ? indices[q] / span : 0;
    Fr in = zero_word();
    if (mine && word == 0) in = tag;
    if (mine && word >= 1 && word <= ARITY) in = update_child<ARITY>(children, n_children, parent, word - 1, pad);
    const Fr out = rows_perm(&d_rows, d_rows_klin, L[wave], in);
    if (mine && word == out_idx) store_word(parents + parent * 32, out);
}

template <int ARITY>
__global__ void __launch_bounds__(kRowsWaves *kWave) k_merkle_verify_rows(const uint8_t *__restrict__ leaves,
                                                                          const uint64_t *__restrict__ indices,
                                                                          const uint8_t *__restrict__ paths, size_t n_queries,
                                                                          int depth, Fr tag, int out_idx,
                                                                          uint8_t *__restrict__ roots) {
    __shared__ RowsLds L[kRowsWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    if (((size_t)blockIdx.x * kRowsWaves + wave) * kRowsPerWave >= n_queries) return;
    size_t q;
    int word;
    const bool mine = rows_role(n_queries, q, word);
    uint64_t idx = mine ? indices[q] : 0;
    const uint8_t *path = paths + (mine ? q : 0) * (size_t)depth * (ARITY - 1) * 32;
    Fr node = mine ? load_word(leaves + q * 32) : zero_word();       // every lane of a query holds its path node
    auto sibling = [&](int l, uint64_t at) {                         // lane 5 s + 1 + c: child c of level l, unless it is the node
        Fr v = zero_word();
        const int pos = (int)(at % ARITY), c = word - 1;
        if (mine && word >= 1 && word <= ARITY && c != pos)
            v = load_word(path + ((size_t)l * (ARITY - 1) + (c < pos ? c : c - 1)) * 32);
        return v;
    };
    Fr sib = sibling(0, idx);
    const int src = lane - word + out_idx;                           // the lane of my query that holds the digest
#pragma unroll 1
    for (int l = 0; l < depth; l++) {
        const int pos = (int)(idx % ARITY);
        idx /= ARITY;
        const Fr in = fr_select(word == 0, tag, fr_select(word == pos + 1, node, sib));
        if (l + 1 < depth) sib = sibling(l + 1, idx);
        const Fr out = rows_perm(&d_rows, d_rows_klin, L[wave], in);
#pragma unroll
        for (int i = 0; i < 8; i++) node.l[i] = __shfl(out.l[i], src < kWave ? src : 0, kWave);
    }
    if (mine && word == 0) store_word(roots + q * 32, node);
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
