// launch.hpp -- launch policy: which kernel runs a batch / a tree level of a given size and with what grid, block and LDS.
// The size thresholds live here and nowhere else (exported through hades252_kernel_for / hades252_chain_form_for).
// Part of the key of the committed counter records (build.device_source_hash): geometry decides traffic per launch.
#pragma once

// device buffers are moved with 16-byte vector loads/stores
static inline bool misaligned(const void *p) { return ((uintptr_t)p & 15u) != 0; }
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
static inline size_t lds_for(int nw) { return (size_t)kWavesPerBlock * lds_wave_bytes(nw); }
static constexpr size_t kMaxLaunchRecords = (size_t)1 << 30;   // grid.x * 256 per launch
// Records per launch of the one entry point that takes more than that and loops (hades252_perm_batch_dev_ex).  Test hook:
// HADES252_TEST_MAX_LAUNCH (read once, at the first call) lowers it so that the loop's second and later trips run on a
// batch of a few thousand states (tests/test_gpu_a01_perm.py::test_multi_launch_loop_with_lowered_cap); values below 256
// (less than one block per launch: nothing a test needs, and a stray setting would turn a call into millions of launches)
// are ignored.  Everything else keeps rejecting n > kMaxLaunchRecords.
static size_t max_launch_records() {
    static const size_t v = []() -> size_t {
        const char *e = getenv("HADES252_TEST_MAX_LAUNCH");
        const size_t t = e ? (size_t)strtoull(e, nullptr, 0) : 0;
        return t >= 256 && t < kMaxLaunchRecords ? t : kMaxLaunchRecords;
    }();
    return v;
}

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
// ... and up to one wave per SIMD with four states per wave (one per 16-lane row) beats five waves per state
static constexpr size_t kRowsMaxStates = (size_t)1 << 12;

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

// four parents per wave (levels of 1 025 .. 4 096 parents)
static void launch_merkle_rows(int arity, const uint8_t *children, size_t n_children, uint8_t *parents, size_t n, Fr tag,
                               int out_idx, const uint8_t *pad, hipStream_t s) {
    const dim3 grid((unsigned)((n + kRowsWaves * kRowsPerWave - 1) / (kRowsWaves * kRowsPerWave))), block(kRowsWaves * kWave);
#define HADES_LAUNCH_ROWS(A) \
    hipLaunchKernelGGL(k_merkle_rows<A>, grid, block, 0, s, children, n_children, parents, n, tag, out_idx, pad)
    switch (arity) {
        case 1: HADES_LAUNCH_ROWS(1); break;
        case 2: HADES_LAUNCH_ROWS(2); break;
        case 3: HADES_LAUNCH_ROWS(3); break;
        default: HADES_LAUNCH_ROWS(4); break;
    }
#undef HADES_LAUNCH_ROWS
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
        if (!lanes && n_updates <= kRowsMaxStates)                                                                      \
            hipLaunchKernelGGL(k_merkle_update_rows<A>,                                                                 \
                               dim3((unsigned)((n_updates + kRowsWaves * kRowsPerWave - 1) / (kRowsWaves * kRowsPerWave))), \
                               dim3(kRowsWaves * kWave), 0, s, children, n_children, parents, indices, n_updates, n_leaves, \
                               span, tag, out_idx, pad);                                                                \
        else if (!lanes && n_updates <= kCoopMaxStates)                                                                 \
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
    else if (n_parents <= kRowsMaxStates)
        launch_merkle_rows(arity, children, n_children, parents, n_parents, tag, out_idx, pad, s);
    else if (n_parents <= kCoopMaxStates && n_children % arity == 0)
        launch_merkle_coop(arity, children, nullptr, parents, n_parents, tag, out_idx, 1, s);
    else
        launch_merkle_level(arity, children, n_children, parents, n_parents, tag, out_idx, pad, s);
}

// The size rule of the default dispatch, in one place (exported: hades252_kernel_for / hades252_chain_form_for).
static inline int kernel_for(size_t n) {
    return n <= kLanesMaxStates  ? HADES252_KERNEL_LANES
           : n <= kRowsMaxStates ? HADES252_KERNEL_ROWS
           : n <= kCoopMaxStates ? HADES252_KERNEL_COOP
                                 : HADES252_KERNEL_FAST;
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
