// abi_sponge.hpp -- C ABI, device-resident data: sponge hashing over the batched permutation: fixed and
// ragged message batches, the on-device sort by block count, the streaming init / absorb / squeeze form.
#pragma once

extern "C" {

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
    if (n_msgs <= kRowsMaxStates) {                     // four messages per wave, one per 16-lane row
        hipLaunchKernelGGL(k_sponge_rows, dim3((unsigned)((n_msgs + kRowsWaves * kRowsPerWave - 1) / (kRowsWaves * kRowsPerWave))),
                           dim3(kRowsWaves * kWave), 0, (hipStream_t)stream, (const uint8_t *)d_scalars, d_offsets, d_lengths,
                           (uint8_t *)d_digests, n_msgs, fixed_len, fr_from_u64(capacity_mont), pad_mode, n_scalars,
                           d_bad_count);
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
    if (n_states <= kRowsMaxStates) {
        hipLaunchKernelGGL(k_sponge_absorb_rows,
                           dim3((unsigned)((n_states + kRowsWaves * kRowsPerWave - 1) / (kRowsWaves * kRowsPerWave))),
                           dim3(kRowsWaves * kWave), 0, (hipStream_t)stream, (uint8_t *)d_states, (const uint8_t *)d_blocks,
                           n_states, blocks_each);
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

}  // extern "C"
