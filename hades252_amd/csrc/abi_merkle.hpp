// abi_merkle.hpp -- C ABI, device-resident data: Merkle trees over the batched permutation: levels,
// roots, whole trees, openings, verification, updates, forests.  `merkle_run` is the level / fusion policy of a tree build
// (part of the key of the committed counter records, like launch.hpp).
#pragma once

extern "C" {

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
        if ((arity == 2 || arity == 4) && parents > kRowsMaxStates && parents <= kCoopMaxStates && log_arity(n, arity) > 0) {
            // fuse while the level after the last fused one is still too large for the per-row / per-wave kernels
            const int max_fused = log_arity(kCoopStates, arity) + 1;                   // 64 parents -> 1 digest
            size_t sz = parents;
            while (fused < max_fused && sz / arity > kRowsMaxStates) {
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
    // e0 travels as a kernel argument, like the tag: nothing of the caller's host memory is referenced after this
    // call returns, so the whole sequence is graph-capturable like every other _dev entry point
    hipLaunchKernelGGL(k_store_fr, dim3(1), dim3(kWave), 0, s, (uint32_t *)pad, fr_from_u64(e0_mont));
    HIP_TRY(hipGetLastError());
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
        if (!lanes && n_queries <= kRowsMaxStates)                                                                   \
            hipLaunchKernelGGL(k_merkle_verify_rows<A>,                                                              \
                               dim3((unsigned)((n_queries + kRowsWaves * kRowsPerWave - 1) / (kRowsWaves * kRowsPerWave))), \
                               dim3(kRowsWaves * kWave), 0, (hipStream_t)stream, HADES_VERIFY_ARGS);                 \
        else if (!lanes && n_queries <= kCoopMaxStates)                                                              \
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

}  // extern "C"
