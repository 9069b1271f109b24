// hades252.hip -- the one translation unit of libhades252 (gfx950 only): arithmetic headers, constant tables, the kernel
// headers by domain (kernels_*.hpp) and, below, launch policy + the C ABI of include/hades252.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <condition_variable>
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

#include "device_tables.hpp"
#include "kernels_perm.hpp"
#include "kernels_merkle.hpp"
#include "kernels_sponge.hpp"
#include "kernels_aux.hpp"

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

// ---- fault injection (test hook, include/hades252.h: hades252_fault_inject / HADES252_FAIL_AT) -----------------------
// F(site, call): the call, unless the hook is armed for `site` and this is its nth occurrence -- then the error the
// runtime would have returned.  Disarmed: one relaxed load and a compare.
enum FaultSite { F_NONE = -1, F_MALLOC, F_HOSTMALLOC, F_HOSTREGISTER, F_MEMCPY, F_STREAMCREATE, F_EVENTCREATE, F_SYNC,
                 F_WORKER, F_THREAD, F_N_SITES };
static const char *const kFaultNames[F_N_SITES] = {"malloc", "hostmalloc", "hostregister", "memcpy", "streamcreate",
                                                   "eventcreate", "sync", "worker", "thread"};
static std::atomic<int> g_fault_site{F_NONE};
static std::atomic<long> g_fault_nth{0};
static int fault_arm(const char *spec) {
    if (spec == nullptr || spec[0] == 0) {
        g_fault_site.store(F_NONE);
        return HADES252_OK;
    }
    const char *colon = strchr(spec, ':');
    const size_t len = colon ? (size_t)(colon - spec) : strlen(spec);
    const long nth = colon ? strtol(colon + 1, nullptr, 10) : 1;
    for (int i = 0; i < F_N_SITES; i++)
        if (strlen(kFaultNames[i]) == len && strncmp(kFaultNames[i], spec, len) == 0 && nth >= 1) {
            g_fault_site.store(F_NONE);
            g_fault_nth.store(nth);
            g_fault_site.store(i);
            return HADES252_OK;
        }
    return HADES252_ERR_INVALID_ARG;
}
static const int g_fault_env = fault_arm(getenv("HADES252_FAIL_AT"));      // at load time
static inline bool fault_hit(int site) {
    if (g_fault_site.load(std::memory_order_relaxed) != site) return false;
    if (g_fault_nth.fetch_sub(1) != 1) return false;
    g_fault_site.store(F_NONE);                                             // fires once
    return true;
}
#define F(site, call) (fault_hit(site) ? (site == F_MALLOC || site == F_HOSTMALLOC ? hipErrorOutOfMemory : hipErrorUnknown) : (call))

// std::thread's constructor throws when the system refuses another thread; no exception may cross the C boundary.
template <class Fn>
static bool spawn(std::vector<std::thread> &threads, Fn &&fn) {
    if (fault_hit(F_THREAD)) return false;
    try {
        threads.emplace_back(std::forward<Fn>(fn));
        return true;
    } catch (...) {
        return false;
    }
}

// device buffers are moved with 16-byte vector loads/stores
static inline bool misaligned(const void *p) { return ((uintptr_t)p & 15u) != 0; }
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
static inline size_t lds_for(int nw) { return (size_t)kWavesPerBlock * lds_wave_bytes(nw); }
static constexpr size_t kMaxLaunchRecords = (size_t)1 << 30;   // grid.x * 256 per launch
// Records per launch of the one entry point that takes more than that and loops (hades252_perm_batch_dev_ex).  Test hook:
// HADES252_TEST_MAX_LAUNCH (read once, at the first call) lowers it so that the loop's second and later trips run on a
// batch of a few thousand states (tests/test_gpu_perm.py); everything else keeps rejecting n > kMaxLaunchRecords.
static size_t max_launch_records() {
    static const size_t v = []() -> size_t {
        const char *e = getenv("HADES252_TEST_MAX_LAUNCH");
        const size_t t = e ? (size_t)strtoull(e, nullptr, 0) : 0;
        return t >= 1 && t < kMaxLaunchRecords ? t : kMaxLaunchRecords;
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
    if (kernel == HADES252_KERNEL_DEFAULT) kernel = kernel_for(n_perms);
    const size_t cap = max_launch_records();
    for (size_t off = 0; off < n_perms; off += cap) {
        size_t n = n_perms - off < cap ? n_perms - off : cap;
        if (kernel == HADES252_KERNEL_LANES) {
            if (n <= kLanesHelpedMaxStates)
                hipLaunchKernelGGL(k_perm_lanes<true>, dim3((unsigned)((n + kLanesWaves - 2) / (kLanesWaves - 1))),
                                   dim3(kLanesWaves * kWave), 0, s, p + off * 160, n);
            else
                hipLaunchKernelGGL(k_perm_lanes<false>, dim3((unsigned)((n + kLanesWaves - 1) / kLanesWaves)),
                                   dim3(kLanesWaves * kWave), 0, s, p + off * 160, n);
        } else if (kernel == HADES252_KERNEL_ROWS) {
            hipLaunchKernelGGL(k_perm_rows, dim3((unsigned)((n + kRowsWaves * kRowsPerWave - 1) / (kRowsWaves * kRowsPerWave))),
                               dim3(kRowsWaves * kWave), 0, s, p + off * 160, n);
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

int hades252_kernel_for(size_t n_perms) { return kernel_for(n_perms); }
int hades252_chain_form_for(size_t n_chains) { return kernel_for(n_chains); }
const char *hades252_kernel_name(int kernel, size_t n_perms) {
    if (kernel == HADES252_KERNEL_DEFAULT) kernel = kernel_for(n_perms);
    switch (kernel) {
        case HADES252_KERNEL_LITERAL: return "k_states_literal";
        case HADES252_KERNEL_FAST: return "k_perm_fast";
        case HADES252_KERNEL_COOP: return "k_perm_coop";
        case HADES252_KERNEL_LANES: return "k_perm_lanes";
        case HADES252_KERNEL_ROWS: return "k_perm_rows";
        default: return nullptr;
    }
}

int hades252_fault_inject(const char *spec) { return fault_arm(spec); }

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
// Pipes are created on demand, handed out exclusively and returned.  The pool is bounded per device: at most
// kPoolMaxPipes pipes and at most pool_max_bytes() of device memory (chunk buffers: kPipeSlots x up to 40 MiB per pipe;
// the arena of the one-shot Merkle / sponge calls: whatever the largest call needed) -- release_pipe() strips a
// returning pipe of its arena, then of its chunk buffers, when keeping them would exceed the budget, and destroys it
// when the device already has kPoolMaxPipes; hades252_trim() empties the pool.
constexpr int kPipeSlots = 6;
constexpr int kPoolMaxPipes = 16;
struct HostPipe {
    int device = -1;
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
    void *buf = nullptr;          // kPipeSlots slots of slot_cap bytes
    size_t slot_cap = 0;
    hipEvent_t in_done[kPipeSlots] = {}, k_done[kPipeSlots] = {}, out_done[kPipeSlots] = {};
    void *pinned = nullptr;       // small-call staging: page-locked host memory the kernels access directly
    void *pinned_dev = nullptr;   // ... and its device-side address
    void *aux = nullptr;          // grow-only device arena of the one-shot Merkle / sponge calls (levels, digests, tables)
    size_t aux_cap = 0;
    void *stage = nullptr;        // page-locked staging of the pageable-caller path (perm_batch_host_staged), 120 MiB
};
// Calls of at most this many states skip both DMA copies: the states are copied (by the CPU) into a
// page-locked buffer that the kernel reads and writes over PCIe itself -- one launch + one synchronisation.
static constexpr size_t kPinnedStates = 256;
static std::mutex g_pool_mu;
static std::vector<HostPipe> g_pool;

static void destroy_pipe(HostPipe &p) {
    if (p.pinned) (void)hipHostFree(p.pinned);
    if (p.stage) (void)hipHostFree(p.stage);
    if (p.buf) (void)hipFree(p.buf);
    if (p.aux) (void)hipFree(p.aux);
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

static size_t pool_max_bytes() {
    static const size_t v = []() -> size_t {
        const char *e = getenv("HADES252_POOL_MAX_BYTES");
        return e ? (size_t)strtoull(e, nullptr, 0) : (size_t)1 << 30;
    }();
    return v;
}
static inline size_t pipe_bytes(const HostPipe &p) { return p.slot_cap * kPipeSlots + p.aux_cap; }

// A pipe that saw a failure is never pooled (its streams may hold a sticky error): pass failed = true.
static void release_pipe(HostPipe p, bool failed = false) {
    if (failed) {
        destroy_pipe(p);
        return;
    }
    void *free_aux = nullptr, *free_buf = nullptr, *free_stage = nullptr;
    bool destroy = false;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t held = 0;
        int count = 0, staged = 0;
        for (const HostPipe &q : g_pool)
            if (q.device == p.device) {
                held += pipe_bytes(q);
                count++;
                staged += q.stage != nullptr;
            }
        if (p.stage != nullptr && staged >= 2) {                               // at most two 120 MiB page-locked staging
            free_stage = p.stage;                                              // buffers stay cached per device
            p.stage = nullptr;
        }
        if (count >= kPoolMaxPipes) {
            destroy = true;
        } else {
            if (held + pipe_bytes(p) > pool_max_bytes() && p.aux) {            // the arena goes first ...
                free_aux = p.aux;
                p.aux = nullptr;
                p.aux_cap = 0;
            }
            if (held + pipe_bytes(p) > pool_max_bytes() && p.buf) {            // ... then the chunk buffers
                free_buf = p.buf;
                p.buf = nullptr;
                p.slot_cap = 0;
            }
            g_pool.push_back(p);
        }
    }
    if (destroy) destroy_pipe(p);
    if (free_aux) (void)hipFree(free_aux);
    if (free_buf) (void)hipFree(free_buf);
    if (free_stage) (void)hipHostFree(free_stage);
    if (free_aux || free_buf || free_stage) (void)hipGetLastError();
}

// slot_bytes == 0: a small call (needs the page-locked staging buffer, no device buffer).  want_stage: the call will go
// through the staging threads -- it first looks among the pooled pipes that already own the 120 MiB page-locked staging
// buffer (otherwise a stage-less pipe would allocate a second one while a staged pipe sits idle, and release_pipe would
// free one of the two again: tens of milliseconds of hipHostMalloc / hipHostFree per call).
static int acquire_pipe(size_t slot_bytes, HostPipe &out, bool want_stage = false) {
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
                // large call: (a staged pipe for a staging call, then) the smallest buffer that fits, else the largest
                const size_t bc = g_pool[best].slot_cap, ic = g_pool[i].slot_cap;
                const bool bs = want_stage && g_pool[best].stage != nullptr, is = want_stage && g_pool[i].stage != nullptr;
                if (is != bs) {
                    if (is) best = i;
                } else if (bc >= slot_bytes ? (ic >= slot_bytes && ic < bc) : ic > bc) {
                    best = i;
                }
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
        if ((e = F(F_STREAMCREATE, hipStreamCreateWithFlags(&p.s_in, hipStreamNonBlocking))) != hipSuccess) return fail(e);
        if ((e = F(F_STREAMCREATE, hipStreamCreateWithFlags(&p.s_k, hipStreamNonBlocking))) != hipSuccess) return fail(e);
        if ((e = F(F_STREAMCREATE, hipStreamCreateWithFlags(&p.s_out, hipStreamNonBlocking))) != hipSuccess) return fail(e);
        for (int i = 0; i < kPipeSlots; i++) {
            if ((e = F(F_EVENTCREATE, hipEventCreateWithFlags(&p.in_done[i], hipEventDisableTiming))) != hipSuccess) return fail(e);
            if ((e = F(F_EVENTCREATE, hipEventCreateWithFlags(&p.k_done[i], hipEventDisableTiming))) != hipSuccess) return fail(e);
            if ((e = F(F_EVENTCREATE, hipEventCreateWithFlags(&p.out_done[i], hipEventDisableTiming))) != hipSuccess) return fail(e);
        }
    }
    if (slot_bytes == 0 && p.pinned_dev == nullptr) {
        if (p.pinned) (void)hipHostFree(p.pinned);
        p.pinned = nullptr;
        if ((e = F(F_HOSTMALLOC, hipHostMalloc(&p.pinned, kPinnedStates * 160, hipHostMallocMapped))) != hipSuccess) return fail(e);
        if ((e = hipHostGetDevicePointer(&p.pinned_dev, p.pinned, 0)) != hipSuccess) return fail(e);
    }
    if (p.slot_cap < slot_bytes) {
        if (p.buf) (void)hipFree(p.buf);
        p.buf = nullptr;
        p.slot_cap = 0;
        if ((e = F(F_MALLOC, hipMalloc(&p.buf, slot_bytes * kPipeSlots))) != hipSuccess) return fail(e);
        p.slot_cap = slot_bytes;
    }
    out = p;
    return HADES252_OK;
}

int hades252_trim(void) {
    std::vector<HostPipe> victims;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        victims.swap(g_pool);
    }
    int cur = -1;
    if (!victims.empty() && hipGetDevice(&cur) != hipSuccess) cur = -1;
    for (HostPipe &p : victims) {
        (void)hipSetDevice(p.device);
        destroy_pipe(p);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    return HADES252_OK;
}

size_t hades252_pool_bytes(void) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t total = 0;
    for (const HostPipe &q : g_pool) total += pipe_bytes(q);
    return total;
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

// A big batch in ORDINARY memory.  Page-locking it costs more than moving it (tools/pin_probe.hip on this pool: a first
// hipHostRegister runs at 18 GB/s, the link moves 47 GB/s each way; hipHostUnregister waits for the device to go idle), while
// a CPU core copies into page-locked memory at 30 GB/s and four cores at 64 GB/s.  So the caller's pages are never locked:
// helper threads copy chunk after chunk into page-locked staging buffers the pipe owns, the chunk pipeline of the
// page-locked path runs on those, and as many threads copy the results back behind the device -> host copies.  Six
// slots per direction; a chunk is 2^16 states (10 MiB); thread t of a direction takes chunks t, t + T, ...
constexpr int kStageSlots = kPipeSlots;              // one staging slot per device chunk buffer and direction
constexpr size_t kStageChunkStates = (size_t)1 << 16;
// copy threads per direction (HADES252_STAGE_THREADS, 1 .. 6).  Beside each other the threads get ~15 GB/s apiece
// (tools/pin_probe.hip: 4 threads 64 GB/s, 8 threads 120 GB/s); the link wants 44 GB/s each way: three per direction.
static int stage_threads() {
    static const int v = []() {
        const char *e = getenv("HADES252_STAGE_THREADS");
        int t = e ? atoi(e) : 3;
        return t < 1 ? 1 : (t > kStageSlots ? kStageSlots : t);
    }();
    return v;
}

static bool host_pin_enabled() {
    static const bool v = []() {
        const char *e = getenv("HADES252_HOST_PIN");
        return !(e && e[0] == '0');
    }();
    return v;
}
static int pipe_ensure_stage(HostPipe &p) {
    if (p.stage != nullptr) return HADES252_OK;
    HIP_TRY(F(F_HOSTMALLOC, hipHostMalloc(&p.stage, 2 * kStageSlots * kStageChunkStates * 160, hipHostMallocDefault)));
    return HADES252_OK;
}

// what a chunk goes through on the device: the permutation, between the two wire-format conversions for canonical bytes
static int host_run_kernels(void *d, size_t n, hipStream_t st, bool bytes_format) {
    if (!bytes_format) return hades252_perm_batch_dev(d, n, st);
    int r = hades252_from_bytes_dev(d, d, n * 5, nullptr, st);
    if (r == HADES252_OK) r = hades252_perm_batch_dev(d, n, st);
    if (r == HADES252_OK) r = hades252_to_bytes_dev(d, d, n * 5, st);
    return r;
}

static int perm_batch_host_staged(uint8_t *h, size_t n_perms, HostPipe &pipe, bool bytes_format) {
    const size_t chunk = kStageChunkStates, cb = chunk * 160;
    const size_t n_chunks = (n_perms + chunk - 1) / chunk;
    uint8_t *st_in = (uint8_t *)pipe.stage, *st_out = st_in + (size_t)kStageSlots * cb;
    struct Shared {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<char> filled, drained;           // chunk c is in its staging slot / has been copied back to the caller
        size_t h2d_enq = 0, d2h_enq = 0;             // chunks whose copy (and its event) has been enqueued by the main thread
        bool failed = false;
        int hip_err = 0;
    } sh;
    sh.filled.assign(n_chunks, 0);
    sh.drained.assign(n_chunks, 0);
    auto fail = [&](hipError_t e) {
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.failed = true;
            if (sh.hip_err == 0) sh.hip_err = (int)e;
        }
        sh.cv.notify_all();
    };
    const int device = pipe.device, kStageThreads = stage_threads();
    std::vector<std::thread> threads;
    bool spawned = true;
    for (int t = 0; t < kStageThreads && spawned; t++) {
        spawned = spawn(threads, [&, t]() {                               // caller -> staging
            (void)hipSetDevice(device);
            for (size_t c = t; c < n_chunks; c += kStageThreads) {
                if (c >= (size_t)kStageSlots) {                           // the slot's previous chunk has left for the device
                    {
                        std::unique_lock<std::mutex> lk(sh.mu);
                        sh.cv.wait(lk, [&]() { return sh.failed || sh.h2d_enq > c - kStageSlots; });
                        if (sh.failed) return;
                    }
                    const hipError_t e = F(F_SYNC, hipEventSynchronize(pipe.in_done[c % kStageSlots]));
                    if (e != hipSuccess) return fail(e);
                }
                const size_t off = c * chunk, n = n_perms - off < chunk ? n_perms - off : chunk;
                memcpy(st_in + (c % kStageSlots) * cb, h + off * 160, n * 160);
                {
                    std::lock_guard<std::mutex> lk(sh.mu);
                    sh.filled[c] = 1;
                }
                sh.cv.notify_all();
            }
        });
        spawned = spawned && spawn(threads, [&, t]() {                    // staging -> caller
            (void)hipSetDevice(device);
            for (size_t c = t; c < n_chunks; c += kStageThreads) {
                {
                    std::unique_lock<std::mutex> lk(sh.mu);
                    sh.cv.wait(lk, [&]() { return sh.failed || sh.d2h_enq > c; });
                    if (sh.failed) return;
                }
                const hipError_t e = F(F_SYNC, hipEventSynchronize(pipe.out_done[c % kStageSlots]));
                if (e != hipSuccess) return fail(e);
                const size_t off = c * chunk, n = n_perms - off < chunk ? n_perms - off : chunk;
                memcpy(h + off * 160, st_out + (c % kStageSlots) * cb, n * 160);
                {
                    std::lock_guard<std::mutex> lk(sh.mu);
                    sh.drained[c] = 1;
                }
                sh.cv.notify_all();
            }
        });
    }
    int rc = HADES252_OK;
    hipError_t e = spawned ? hipSuccess : hipErrorOutOfMemory;           // a missing helper would leave chunks unstaged
    for (size_t c = 0; c < n_chunks && rc == HADES252_OK && spawned; c++) {
        const int k = (int)(c % kStageSlots);
        const size_t off = c * chunk, n = n_perms - off < chunk ? n_perms - off : chunk;
        void *d = (uint8_t *)pipe.buf + (size_t)k * pipe.slot_cap;
        {
            std::unique_lock<std::mutex> lk(sh.mu);                       // the chunk is staged; its output slot is free again
            sh.cv.wait(lk, [&]() { return sh.failed || (sh.filled[c] && (c < (size_t)kStageSlots || sh.drained[c - kStageSlots])); });
            if (sh.failed) break;
        }
        // the device buffer of slot k is free: chunk c - kStageSlots has been copied out of it (drained => out_done passed)
        if ((e = F(F_MEMCPY, hipMemcpyAsync(d, st_in + (size_t)k * cb, n * 160, hipMemcpyHostToDevice, pipe.s_in))) != hipSuccess) break;
        if ((e = hipEventRecord(pipe.in_done[k], pipe.s_in)) != hipSuccess) break;
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.h2d_enq = c + 1;
        }
        sh.cv.notify_all();
        if ((e = hipStreamWaitEvent(pipe.s_k, pipe.in_done[k], 0)) != hipSuccess) break;
        rc = host_run_kernels(d, n, pipe.s_k, bytes_format);
        if (rc != HADES252_OK) break;
        if ((e = hipEventRecord(pipe.k_done[k], pipe.s_k)) != hipSuccess) break;
        if ((e = hipStreamWaitEvent(pipe.s_out, pipe.k_done[k], 0)) != hipSuccess) break;
        if ((e = F(F_MEMCPY, hipMemcpyAsync(st_out + (size_t)k * cb, d, n * 160, hipMemcpyDeviceToHost, pipe.s_out))) != hipSuccess) break;
        if ((e = hipEventRecord(pipe.out_done[k], pipe.s_out)) != hipSuccess) break;
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.d2h_enq = c + 1;
        }
        sh.cv.notify_all();
    }
    if (e != hipSuccess) fail(e);
    if (rc != HADES252_OK) fail(hipSuccess);
    for (auto &t : threads) t.join();                                     // the last chunk is back in the caller's buffer
    if (sh.failed) {
        if (rc == HADES252_OK) {
            tl_last_hip_error = sh.hip_err;
            (void)hipGetLastError();
            rc = HADES252_ERR_HIP;
        }
    }
    return rc;
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
                                             bool never_register = false) {
    if (n_perms == 0) return HADES252_OK;
    if (states == nullptr) return HADES252_ERR_INVALID_ARG;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    auto run_kernels = [&](void *d, size_t n, hipStream_t st) { return host_run_kernels(d, n, st, bytes_format); };
    HostPipe pipe;
    if (n_perms <= kPinnedStates) {
        rc = acquire_pipe(0, pipe);
        if (rc != HADES252_OK) return rc;
        memcpy(pipe.pinned, states, n_perms * 160);
        rc = run_kernels(pipe.pinned_dev, n_perms, pipe.s_k);
        hipError_t e = hipStreamSynchronize(pipe.s_k);      // always really drained, whatever the hook says
        if (e == hipSuccess) e = F(F_SYNC, hipSuccess);
        if (rc == HADES252_OK && e == hipSuccess) memcpy(states, pipe.pinned, n_perms * 160);
        release_pipe(pipe, rc != HADES252_OK || e != hipSuccess);   // only now: the staging buffer belongs to the pipe
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
    // the route is decided before the pipe is taken: the staging-thread path moves kStageChunkStates per chunk whatever
    // HADES252_HOST_CHUNK says, so its device slots are sized for that, and it wants a pipe that owns a staging buffer
    const bool unpinned_big = host_pin_enabled() && n_perms * 160 >= ((size_t)8 << 20) && !host_range_pinned(h, n_perms * 160);
    const bool staged = unpinned_big && n_perms > 2 * kStageChunkStates;
    rc = acquire_pipe((staged && chunk < kStageChunkStates ? kStageChunkStates : chunk) * 160, pipe, staged);
    if (rc != HADES252_OK) return rc;
    bool registered = false;
    auto finish = [&](int code) {
        (void)hipStreamSynchronize(pipe.s_in);
        (void)hipStreamSynchronize(pipe.s_k);
        (void)hipStreamSynchronize(pipe.s_out);
        (void)hipGetLastError();
        release_pipe(pipe, code != HADES252_OK);
        if (registered) (void)hipHostUnregister(h);
        return code;
    };
    // Memory the caller has not pinned.  Several chunks: the batch travels through page-locked staging buffers filled and
    // drained by helper threads (perm_batch_host_staged) -- the caller's pages are never locked.  One chunk (8 .. 40 MiB):
    // page-locked in place for the duration of the call, so its two copies are true DMA; if that is refused, or below
    // 8 MiB, the runtime's own pageable copies.  HADES252_HOST_PIN=0 disables both (plain pageable copies).
    if (unpinned_big) {
        if (staged) {
            rc = pipe_ensure_stage(pipe);
            if (rc != HADES252_OK) return finish(rc);
            return finish(perm_batch_host_staged(h, n_perms, pipe, bytes_format));
        }
        if (!never_register) {
            if (F(F_HOSTREGISTER, hipHostRegister(h, n_perms * 160, hipHostRegisterDefault)) == hipSuccess)
                registered = true;
            else
                (void)hipGetLastError();
        }
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
        TRY_FIN(F(F_MEMCPY, hipMemcpyAsync(pipe.buf, h, n_perms * 160, hipMemcpyHostToDevice, pipe.s_k)));
        rc = run_kernels(pipe.buf, n_perms, pipe.s_k);
        if (rc != HADES252_OK) return finish(rc);
        TRY_FIN(F(F_MEMCPY, hipMemcpyAsync(h, pipe.buf, n_perms * 160, hipMemcpyDeviceToHost, pipe.s_k)));
        TRY_FIN(F(F_SYNC, hipStreamSynchronize(pipe.s_k)));
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
        if (c >= (size_t)kPipeSlots) TRY_FIN(F(F_SYNC, hipEventSynchronize(pipe.out_done[k])));   // chunk c - kPipeSlots left slot k
        TRY_FIN(F(F_MEMCPY, hipMemcpyAsync(d, h + off * 160, n * 160, hipMemcpyHostToDevice, pipe.s_in)));
        TRY_FIN(hipEventRecord(pipe.in_done[k], pipe.s_in));
        TRY_FIN(hipStreamWaitEvent(pipe.s_k, pipe.in_done[k], 0));
        rc = run_kernels(d, n, pipe.s_k);
        if (rc != HADES252_OK) return finish(rc);
        TRY_FIN(hipEventRecord(pipe.k_done[k], pipe.s_k));
        TRY_FIN(hipStreamWaitEvent(pipe.s_out, pipe.k_done[k], 0));
        TRY_FIN(F(F_MEMCPY, hipMemcpyAsync(h + off * 160, d, n * 160, hipMemcpyDeviceToHost, pipe.s_out)));
        TRY_FIN(hipEventRecord(pipe.out_done[k], pipe.s_out));
    }
    TRY_FIN(F(F_SYNC, hipStreamSynchronize(pipe.s_out)));   // the last copy-out is behind everything else
    TRY_FIN(hipStreamSynchronize(pipe.s_k));
    TRY_FIN(hipStreamSynchronize(pipe.s_in));
#undef TRY_FIN
    return finish(HADES252_OK);
}

int hades252_perm_batch(uint64_t *states, size_t n_perms) {
    return perm_batch_host_on_current_device(states, n_perms, false);
}

// Pays the one-time costs now instead of inside the first real call: the code object is loaded by a one-state permutation
// on an internal buffer (~35 ms in a fresh process), and -- for a hint above 256 states -- the pipe such a batch would
// take (streams, events, chunk buffers; the page-locked staging buffers too when the hint is big enough for the
// staging-thread path) is created and put into the pool.
int hades252_warm_up(size_t n_perms_hint) {
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    uint64_t one[20] = {0};
    rc = perm_batch_host_on_current_device(one, 1, false);
    if (rc != HADES252_OK || n_perms_hint <= kPinnedStates) return rc;
    const size_t chunk = n_perms_hint < host_chunk_states(n_perms_hint) ? n_perms_hint : host_chunk_states(n_perms_hint);
    HostPipe pipe;
    rc = acquire_pipe(chunk * 160, pipe);
    if (rc != HADES252_OK) return rc;
    if (host_pin_enabled() && n_perms_hint > 2 * kStageChunkStates) rc = pipe_ensure_stage(pipe);
    release_pipe(pipe, rc != HADES252_OK);
    return rc;
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

// ... on several threads for big batches: one thread reads ~10 GB/s, 2^22 states are 671 MB -- 60 ms in front of a 17 ms call
static bool all_canonical_mt(const uint8_t *bytes, size_t n_scalars) {
    unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = n_scalars < ((size_t)1 << 18) ? 1 : (hw >= 8 ? 8 : (hw >= 2 ? hw : 1));
    if (nt == 1) return all_canonical(bytes, n_scalars);
    std::atomic<bool> ok{true};
    std::vector<std::thread> ts;
    for (size_t t = 0; t < nt; t++) {
        auto slice = [&, t]() {
            const size_t b = n_scalars * t / nt, e = n_scalars * (t + 1) / nt;
            if (!all_canonical(bytes + 32 * b, e - b)) ok.store(false);
        };
        if (!spawn(ts, slice)) slice();                                  // no thread to be had: on this one
    }
    for (auto &t : ts) t.join();
    return ok.load();
}

int hades252_perm_batch_bytes(uint8_t *states, size_t n_perms) {
    if (n_perms == 0) return HADES252_OK;
    if (states == nullptr) return HADES252_ERR_INVALID_ARG;
    // reject the whole batch up front, so a failing call leaves the buffer untouched
    if (!all_canonical_mt(states, n_perms * 5)) return HADES252_ERR_NOT_CANONICAL;
    return perm_batch_host_on_current_device((uint64_t *)states, n_perms, true);
}

// Worker threads of the _multi entry points run on the CPUs next to their device when the kernel says which those are
// (/sys/bus/pci/devices/<bus id>/local_cpulist): their staging copies and page-lock calls then stay on the socket the
// GPU hangs off.  Silent no-op when the file is missing, unparsable, or disjoint from the CPUs this process may use.
static void pin_thread_near_device(int dev) {
    char bus[64] = {0}, path[160], line[1024];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, dev) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    for (char *c = bus; *c; c++)
        if (*c >= 'A' && *c <= 'Z') *c = (char)(*c - 'A' + 'a');
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
    FILE *f = fopen(path, "r");
    if (f == nullptr) return;
    const bool got = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!got) return;
    cpu_set_t allowed, want;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    CPU_ZERO(&want);
    int n_want = 0;
    for (char *q = line; *q;) {                                  // "0-15,32-47"
        char *end;
        long a = strtol(q, &end, 10);
        if (end == q) break;
        long b = a;
        if (*end == '-') {
            q = end + 1;
            b = strtol(q, &end, 10);
            if (end == q) return;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) {
                CPU_SET((int)c, &want);
                n_want++;
            }
        q = (*end == ',') ? end + 1 : end;
        if (*end != ',' ) break;
    }
    if (n_want > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof(want), &want);
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
    // Nothing is page-locked here.  A buffer the caller pinned goes straight to DMA on every device; ordinary memory
    // travels through each worker's own staging threads (shards share boundary pages, so a worker must never register
    // its sub-range; one registration of the whole buffer up front -- round 3 -- runs at 18 GB/s against the 47 GB/s each
    // way of EVERY device's link).
    std::vector<int> rcs(n_workers, HADES252_OK);
    std::vector<int> hip_errs(n_workers, 0);
    std::vector<std::thread> threads;
    for (int g = 0; g < n_workers; g++) {
        rcs[g] = HADES252_ERR_HIP;                                       // stands if the thread cannot be started
        hip_errs[g] = (int)hipErrorOutOfMemory;
        spawn(threads, [&, g]() {
            rcs[g] = HADES252_OK;
            size_t b = n_perms * (size_t)g / n_workers, e = n_perms * (size_t)(g + 1) / n_workers;
            hipError_t err = F(F_WORKER, hipSetDevice(virt ? g % avail : g));
            if (err != hipSuccess) {
                rcs[g] = HADES252_ERR_HIP;
                hip_errs[g] = (int)err;
                return;
            }
            pin_thread_near_device(virt ? g % avail : g);
            rcs[g] = perm_batch_host_on_current_device(states + 20 * b, e - b, false, /*never_register=*/true);
            hip_errs[g] = tl_last_hip_error;
        });
    }
    for (auto &t : threads) t.join();
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
    hipLaunchKernelGGL(k_perm_witness, dim3(blocks_for(n_perms)), dim3(kBlock), 0, (hipStream_t)stream,
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
    hipLaunchKernelGGL(k_wire<1>, dim3(blocks_for((n_scalars + kWireU<1> - 1) / kWireU<1>)), dim3(kBlock), 0,
                       (hipStream_t)stream, (const uint8_t *)d_bytes, (uint8_t *)d_limbs, n_scalars, d_bad_count);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_to_bytes_dev(const void *d_limbs, void *d_bytes, size_t n_scalars, void *stream) {
    if (n_scalars == 0) return HADES252_OK;
    if (d_bytes == nullptr || d_limbs == nullptr || n_scalars > kMaxLaunchRecords || misaligned(d_bytes) ||
        misaligned(d_limbs))
        return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_wire<0>, dim3(blocks_for((n_scalars + kWireU<0> - 1) / kWireU<0>)), dim3(kBlock), 0,
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

// ---- device memory for callers without HIP bindings ------------------------------------------------
int hades252_dev_alloc(void **d_ptr, size_t bytes) {
    if (d_ptr == nullptr || bytes == 0) return HADES252_ERR_INVALID_ARG;
    *d_ptr = nullptr;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    HIP_TRY(hipMalloc(d_ptr, bytes));
    return HADES252_OK;
}

int hades252_dev_free(void *d_ptr) {
    if (d_ptr == nullptr) return HADES252_OK;
    HIP_TRY(hipFree(d_ptr));
    return HADES252_OK;
}

int hades252_dev_upload(void *d_dst, const void *h_src, size_t bytes, void *stream) {
    if (bytes == 0) return HADES252_OK;
    if (d_dst == nullptr || h_src == nullptr) return HADES252_ERR_INVALID_ARG;
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return HADES252_OK;
}

int hades252_dev_download(void *h_dst, const void *d_src, size_t bytes, void *stream) {
    if (bytes == 0) return HADES252_OK;
    if (h_dst == nullptr || d_src == nullptr) return HADES252_ERR_INVALID_ARG;
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return HADES252_OK;
}

int hades252_stream_create(void **stream) {
    if (stream == nullptr) return HADES252_ERR_INVALID_ARG;
    *stream = nullptr;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return HADES252_OK;
}

int hades252_stream_destroy(void *stream) {
    if (stream == nullptr) return HADES252_OK;
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return HADES252_OK;
}

int hades252_stream_sync(void *stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return HADES252_OK;
}

// ---- the callers of perm on host memory ----------------------------------------------------------------------
// Input travels host -> device in chunks on the pipe's copy stream while the previous chunk is hashed on its kernel
// stream (the pipe of the host-pointer perm path: same streams, chunk buffers and events); what travels back is 32 bytes
// per tree / message.  Big pageable inputs are page-locked for the duration of the call like hades252_perm_batch does.
static size_t host_chunk_bytes() {
    static const size_t forced = []() -> size_t {
        const char *e = getenv("HADES252_HOST_CHUNK_BYTES");
        return e ? (size_t)strtoull(e, nullptr, 0) : 0;
    }();
    return forced ? forced : (size_t)32 << 20;
}

static int pipe_ensure_aux(HostPipe &p, size_t bytes) {
    if (p.aux_cap >= bytes) return HADES252_OK;
    if (p.aux) (void)hipFree(p.aux);
    p.aux = nullptr;
    p.aux_cap = 0;
    HIP_TRY(F(F_MALLOC, hipMalloc(&p.aux, bytes)));
    p.aux_cap = bytes;
    return HADES252_OK;
}

// Uploads from ORDINARY memory for the one-shot callers (Merkle root, sponge): helper threads copy the input, chunk by
// chunk, into the pipe's page-locked staging slots and the chunk copies to the device start from there -- the caller's
// pages are never locked, for the reasons given at perm_batch_host_staged (a first hipHostRegister runs at 18 GB/s and
// only LOOKS free when a benchmark reuses its buffer: the driver caches the pinning).
class StagedSource {
  public:
    static size_t slot_bytes() { return 2 * kStageChunkStates * 160; }      // kStageSlots of them fill the staging buffer
    // stages [h, h + bytes) in chunks of chunk_bytes <= slot_bytes(); pipe.stage must exist
    StagedSource(const uint8_t *h, size_t bytes, size_t chunk_bytes, HostPipe &pipe)
        : h_(h), bytes_(bytes), cb_(chunk_bytes), pipe_(pipe), n_chunks_((bytes + chunk_bytes - 1) / chunk_bytes) {
        filled_.assign(n_chunks_, 0);
        const int nt = stage_threads();
        for (int t = 0; t < nt; t++)
            if (!spawn(threads_, [this, t, nt]() { run(t, nt); })) {
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    failed_ = true;                        // wait() then returns nullptr: the call fails, nothing hangs
                    hip_err_ = (int)hipErrorOutOfMemory;
                }
                cv_.notify_all();
                break;
            }
    }
    ~StagedSource() { stop(); }
    // staged address of chunk c (blocks until it is there); nullptr if a helper thread failed
    const uint8_t *wait(size_t c) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return failed_ || filled_[c]; });
        return failed_ ? nullptr : (const uint8_t *)pipe_.stage + (c % kStageSlots) * slot_bytes();
    }
    // the copy of chunk c out of its slot has been enqueued on pipe.s_in and pipe.in_done[c % kStageSlots] recorded behind it
    void enqueued(size_t c) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            enq_ = c + 1;
        }
        cv_.notify_all();
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            failed_ = failed_ || enq_ < n_chunks_;       // an early exit of the caller releases the helpers
        }
        cv_.notify_all();
        for (auto &t : threads_)
            if (t.joinable()) t.join();
    }
    int hip_error() const { return hip_err_; }

  private:
    void run(int t, int nt) {
        (void)hipSetDevice(pipe_.device);
        for (size_t c = t; c < n_chunks_; c += nt) {
            if (c >= (size_t)kStageSlots) {               // the slot's previous chunk has left for the device
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&]() { return failed_ || enq_ > c - kStageSlots; });
                    if (failed_) return;
                }
                const hipError_t e = F(F_SYNC, hipEventSynchronize(pipe_.in_done[c % kStageSlots]));
                if (e != hipSuccess) {
                    {
                        std::lock_guard<std::mutex> lk(mu_);
                        failed_ = true;
                        hip_err_ = (int)e;
                    }
                    cv_.notify_all();
                    return;
                }
            }
            const size_t off = c * cb_, n = bytes_ - off < cb_ ? bytes_ - off : cb_;
            memcpy((uint8_t *)pipe_.stage + (c % kStageSlots) * slot_bytes(), h_ + off, n);
            {
                std::lock_guard<std::mutex> lk(mu_);
                filled_[c] = 1;
            }
            cv_.notify_all();
        }
    }
    const uint8_t *h_;
    size_t bytes_, cb_;
    HostPipe &pipe_;
    size_t n_chunks_, enq_ = 0;
    std::vector<char> filled_;
    bool failed_ = false;
    int hip_err_ = 0;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::thread> threads_;
};

struct HostCall {                 // releases what a one-shot host call holds, whichever way it ends
    HostPipe pipe;
    StagedSource *src = nullptr;  // upload through staging threads (input in ordinary memory)
    bool have_pipe = false;
    // decides how the input travels: through staging threads when it is big, in ordinary memory and HADES252_HOST_PIN
    // allows; else straight from the caller's memory (DMA if page-locked, the runtime's pageable copy otherwise).
    // *chunk_bytes is clamped to a staging slot in the first case.  Call after acquire_pipe.
    static bool will_stage(const void *h, size_t bytes) {
        return host_pin_enabled() && bytes >= ((size_t)8 << 20) && !host_range_pinned(h, bytes);
    }
    int plan_upload(const void *h, size_t bytes, size_t *chunk_bytes, size_t granule) {
        if (!will_stage(h, bytes)) return HADES252_OK;
        int rc = pipe_ensure_stage(pipe);
        if (rc != HADES252_OK) return rc;
        size_t cb = *chunk_bytes < StagedSource::slot_bytes() ? *chunk_bytes : StagedSource::slot_bytes();
        cb -= cb % granule;
        *chunk_bytes = cb;
        src = new StagedSource((const uint8_t *)h, bytes, cb, pipe);
        return HADES252_OK;
    }
    int finish(int code) {
        if (src) src->stop();
        if (have_pipe) {
            (void)hipStreamSynchronize(pipe.s_in);
            (void)hipStreamSynchronize(pipe.s_k);
            (void)hipStreamSynchronize(pipe.s_out);
            (void)hipGetLastError();
            release_pipe(pipe, code != HADES252_OK);
            have_pipe = false;
        }
        if (src) {
            delete src;
            src = nullptr;
        }
        return code;
    }
    // a staging thread failed (StagedSource::wait returned nullptr): the call fails with THAT thread's HIP error,
    // whatever an earlier call left in the thread-local
    int staged_failure() {
        tl_last_hip_error = src ? src->hip_error() : (int)hipErrorUnknown;
        return finish(HADES252_ERR_HIP);
    }
};

#define TRY_CALL(call, expr)                           \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) {                        \
            tl_last_hip_error = (int)e_;               \
            (void)hipGetLastError();                   \
            return (call).finish(HADES252_ERR_HIP);    \
        }                                              \
    } while (0)

static int merkle_root_host(const uint64_t *leaves, size_t n_leaves, int arity, const uint64_t tag_mont[4], int out_idx,
                            const uint64_t *pad, uint64_t root[4]) {
    const int depth = hades252_merkle_depth(n_leaves, arity);
    if (leaves == nullptr || root == nullptr || tag_mont == nullptr || depth < 1 || out_idx < 0 || out_idx >= 5)
        return HADES252_ERR_INVALID_ARG;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    size_t chunk = host_chunk_bytes() / 32;                           // leaves per chunk, a multiple of the arity
    chunk -= chunk % arity;
    if (chunk < (size_t)arity) chunk = arity;
    if (chunk > n_leaves) chunk = n_leaves;
    const size_t n1 = (n_leaves + arity - 1) / arity;
    const size_t scratch = hades252_merkle_scratch_bytes(n1, arity);  // 0 unless the tree over level 1 has >= 2 levels
    const size_t head = (size_t)depth * 32 + 32;                      // padding table, root
    HostCall call;
    rc = acquire_pipe(chunk * 32, call.pipe, HostCall::will_stage(leaves, n_leaves * 32));
    if (rc != HADES252_OK) return rc;
    call.have_pipe = true;
    HostPipe &pp = call.pipe;
    {
        size_t cbytes = chunk * 32;
        rc = call.plan_upload(leaves, n_leaves * 32, &cbytes, (size_t)32 * arity);
        if (rc != HADES252_OK) return call.finish(rc);
        chunk = cbytes / 32;
    }
    rc = pipe_ensure_aux(pp, head + n1 * 32 + scratch);
    if (rc != HADES252_OK) return call.finish(rc);
    uint8_t *d_pad = (uint8_t *)pp.aux, *d_root = d_pad + (size_t)depth * 32, *d_l1 = d_pad + head;
    uint8_t *buf_a = d_l1 + n1 * 32, *buf_b = buf_a + ((n1 + arity - 1) / arity) * 32;
    const Fr tag = fr_from_u64(tag_mont);
    if (pad != nullptr) TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(d_pad, pad, (size_t)depth * 32, hipMemcpyHostToDevice, pp.s_k)));
    const uint8_t *dp = pad != nullptr ? d_pad : nullptr;
    const uint8_t *h = (const uint8_t *)leaves;
    const size_t n_chunks = (n_leaves + chunk - 1) / chunk;
    for (size_t c = 0; c < n_chunks; c++) {                           // level 1, chunk by chunk behind the copies
        const int k = (int)(c % kPipeSlots);
        const size_t off = c * chunk, n = n_leaves - off < chunk ? n_leaves - off : chunk;
        uint8_t *d = (uint8_t *)pp.buf + (size_t)k * pp.slot_cap;
        if (c >= (size_t)kPipeSlots) TRY_CALL(call, F(F_SYNC, hipEventSynchronize(pp.k_done[k])));   // chunk c - kPipeSlots is hashed
        const uint8_t *from = call.src ? call.src->wait(c) : h + off * 32;
        if (from == nullptr) return call.staged_failure();
        TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(d, from, n * 32, hipMemcpyHostToDevice, pp.s_in)));
        TRY_CALL(call, hipEventRecord(pp.in_done[k], pp.s_in));
        if (call.src) call.src->enqueued(c);
        TRY_CALL(call, hipStreamWaitEvent(pp.s_k, pp.in_done[k], 0));
        launch_merkle_any(arity, d, n, n1 == 1 ? d_root : d_l1 + (off / arity) * 32, tag, out_idx, dp, pp.s_k);
        TRY_CALL(call, hipGetLastError());
        TRY_CALL(call, hipEventRecord(pp.k_done[k], pp.s_k));
    }
    if (n1 > 1) {
        rc = merkle_run(d_l1, n1, arity, nullptr, buf_a, buf_b, d_root, tag, out_idx, dp != nullptr ? dp + 32 : nullptr,
                        pp.s_k);
        if (rc != HADES252_OK) return call.finish(rc);
    }
    uint64_t got[4];                                                   // the caller's root is written on success only
    TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(got, d_root, 32, hipMemcpyDeviceToHost, pp.s_k)));
    TRY_CALL(call, F(F_SYNC, hipStreamSynchronize(pp.s_k)));
    memcpy(root, got, 32);
    return call.finish(HADES252_OK);
}

int hades252_merkle_root(const uint64_t *leaves, size_t n_leaves, int arity, const uint64_t tag_mont[4], int out_idx,
                         const uint64_t *pad, uint64_t root[4]) {
    return merkle_root_host(leaves, n_leaves, arity, tag_mont, out_idx, pad, root);
}

int hades252_sponge_hash(const uint64_t *msgs, size_t n_msgs, size_t msg_len, const uint64_t capacity_mont[4],
                         int pad_mode, uint64_t *digests) {
    if (n_msgs == 0) return HADES252_OK;
    if (digests == nullptr || capacity_mont == nullptr || (msgs == nullptr && msg_len > 0) ||
        (pad_mode != 0 && pad_mode != 1) || (msg_len > 0 && n_msgs > (SIZE_MAX / 32) / msg_len))
        return HADES252_ERR_INVALID_ARG;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    const size_t msg_bytes = msg_len * 32;
    size_t chunk = msg_bytes ? host_chunk_bytes() / msg_bytes : n_msgs;                  // messages per chunk
    if (chunk == 0) chunk = 1;
    if (chunk > n_msgs) chunk = n_msgs;
    if (chunk > kMaxLaunchRecords) chunk = kMaxLaunchRecords;
    HostCall call;
    rc = acquire_pipe(chunk * msg_bytes > 16 ? chunk * msg_bytes : 16, call.pipe,
                      msg_bytes && msg_bytes <= StagedSource::slot_bytes() && HostCall::will_stage(msgs, n_msgs * msg_bytes));
    if (rc != HADES252_OK) return rc;
    call.have_pipe = true;
    HostPipe &pp = call.pipe;
    if (msg_bytes && msg_bytes <= StagedSource::slot_bytes()) {
        size_t cbytes = chunk * msg_bytes;
        rc = call.plan_upload(msgs, n_msgs * msg_bytes, &cbytes, msg_bytes);
        if (rc != HADES252_OK) return call.finish(rc);
        chunk = cbytes / msg_bytes;
    }
    rc = pipe_ensure_aux(pp, (size_t)kPipeSlots * chunk * 32);                           // digests of the chunks in flight
    if (rc != HADES252_OK) return call.finish(rc);
    const uint8_t *h = (const uint8_t *)msgs;
    uint8_t *out = (uint8_t *)digests;
    const size_t n_chunks = (n_msgs + chunk - 1) / chunk;
    for (size_t c = 0; c < n_chunks; c++) {
        const int k = (int)(c % kPipeSlots);
        const size_t off = c * chunk, n = n_msgs - off < chunk ? n_msgs - off : chunk;
        uint8_t *d = (uint8_t *)pp.buf + (size_t)k * pp.slot_cap, *dd = (uint8_t *)pp.aux + (size_t)k * chunk * 32;
        if (c >= (size_t)kPipeSlots) TRY_CALL(call, F(F_SYNC, hipEventSynchronize(pp.out_done[k])));
        if (msg_bytes) {
            const uint8_t *from = call.src ? call.src->wait(c) : h + off * msg_bytes;
            if (from == nullptr) return call.staged_failure();
            TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(d, from, n * msg_bytes, hipMemcpyHostToDevice, pp.s_in)));
        }
        TRY_CALL(call, hipEventRecord(pp.in_done[k], pp.s_in));
        if (call.src) call.src->enqueued(c);
        TRY_CALL(call, hipStreamWaitEvent(pp.s_k, pp.in_done[k], 0));
        rc = sponge_launch(d, nullptr, nullptr, n, msg_len, capacity_mont, pad_mode, dd, pp.s_k, n * msg_len, nullptr, nullptr);
        if (rc != HADES252_OK) return call.finish(rc);
        TRY_CALL(call, hipEventRecord(pp.k_done[k], pp.s_k));
        TRY_CALL(call, hipStreamWaitEvent(pp.s_out, pp.k_done[k], 0));
        TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(out + off * 32, dd, n * 32, hipMemcpyDeviceToHost, pp.s_out)));
        TRY_CALL(call, hipEventRecord(pp.out_done[k], pp.s_out));
    }
    TRY_CALL(call, F(F_SYNC, hipStreamSynchronize(pp.s_out)));
    return call.finish(HADES252_OK);
}
// The tree sharded over several devices (every GPU builds complete sub-trees, the sub-roots are hashed
// by one more small tree; no collective, the only exchange is 32 bytes per sub-tree through host memory).  Full trees
// only (n_leaves = arity^k): the sub-trees are the S = arity^j nodes of one level, S the smallest power of the arity that
// is >= n_workers; worker g takes sub-trees [S g / W, S (g + 1) / W) on device g (or g % devices with HADES252_MULTI_VIRTUAL).
int hades252_merkle_root_multi(const uint64_t *leaves, size_t n_leaves, int arity, const uint64_t tag_mont[4], int out_idx,
                               int n_workers, unsigned flags, uint64_t root[4]) {
    if (flags & ~(unsigned)HADES252_MULTI_VIRTUAL) return HADES252_ERR_INVALID_ARG;
    const int k = log_arity(n_leaves, arity);
    if (leaves == nullptr || root == nullptr || tag_mont == nullptr || k < 1 || out_idx < 0 || out_idx >= 5)
        return HADES252_ERR_INVALID_ARG;
    const int avail = hades252_device_count();
    if (avail <= 0) return HADES252_ERR_NO_DEVICE;
    const bool virt = (flags & HADES252_MULTI_VIRTUAL) != 0;
    if (n_workers <= 0) n_workers = avail;
    if (n_workers > (virt ? 64 : avail)) return HADES252_ERR_INVALID_ARG;
    size_t n_sub = 1;                                                // sub-trees: a power of the arity, each >= arity leaves
    while (n_sub < (size_t)n_workers && n_sub * arity * arity <= n_leaves) n_sub *= arity;
    if ((size_t)n_workers > n_sub) n_workers = (int)n_sub;
    if (n_sub == 1) return hades252_merkle_root(leaves, n_leaves, arity, tag_mont, out_idx, nullptr, root);
    const size_t per = n_leaves / n_sub;
    // (nothing is page-locked here: leaves in ordinary memory travel through each worker's staging threads)
    std::vector<uint64_t> sub(n_sub * 4);
    std::vector<int> rcs(n_workers, HADES252_OK), hip_errs(n_workers, 0);
    std::vector<std::thread> threads;
    for (int g = 0; g < n_workers; g++) {
        rcs[g] = HADES252_ERR_HIP;                                       // stands if the thread cannot be started
        hip_errs[g] = (int)hipErrorOutOfMemory;
        spawn(threads, [&, g]() {
            rcs[g] = HADES252_OK;
            hipError_t err = F(F_WORKER, hipSetDevice(virt ? g % avail : g));
            if (err != hipSuccess) {
                rcs[g] = HADES252_ERR_HIP;
                hip_errs[g] = (int)err;
                return;
            }
            pin_thread_near_device(virt ? g % avail : g);
            const size_t b = n_sub * (size_t)g / n_workers, e = n_sub * (size_t)(g + 1) / n_workers;
            for (size_t t = b; t < e && rcs[g] == HADES252_OK; t++)
                rcs[g] = merkle_root_host(leaves + t * per * 4, per, arity, tag_mont, out_idx, nullptr, &sub[t * 4]);
            hip_errs[g] = tl_last_hip_error;
        });
    }
    for (auto &t : threads) t.join();
    for (int g = 0; g < n_workers; g++)
        if (rcs[g] != HADES252_OK) {
            tl_last_hip_error = hip_errs[g];
            return rcs[g];
        }
    return hades252_merkle_root(sub.data(), n_sub, arity, tag_mont, out_idx, nullptr, root);
}

// Variable-length form: the whole pool is uploaded (messages may overlap and lie anywhere in it), offsets / lengths with
// it; ragged batches are sorted by block count on the device as hades252_sponge_hash_var_ex_dev does with scratch.
int hades252_sponge_hash_var(const uint64_t *scalars, size_t n_scalars, const uint64_t *offsets, const uint64_t *lengths,
                             size_t n_msgs, const uint64_t capacity_mont[4], int pad_mode, uint64_t *digests,
                             size_t *n_bad) {
    if (n_bad != nullptr) *n_bad = 0;
    if (n_msgs == 0) return HADES252_OK;
    if (digests == nullptr || capacity_mont == nullptr || offsets == nullptr || lengths == nullptr ||
        (scalars == nullptr && n_scalars > 0) || (pad_mode != 0 && pad_mode != 1) || n_msgs > kMaxLaunchRecords ||
        n_scalars > SIZE_MAX / 64)
        return HADES252_ERR_INVALID_ARG;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t pool_b = up16(n_scalars * 32 + 16), idx_b = up16(n_msgs * 8), dig_b = n_msgs * 32;
    const size_t scr_b = up16(hades252_sponge_sort_scratch_bytes(n_msgs));
    HostCall call;
    rc = acquire_pipe(16, call.pipe, n_scalars && HostCall::will_stage(scalars, n_scalars * 32));
    if (rc != HADES252_OK) return rc;
    call.have_pipe = true;
    HostPipe &pp = call.pipe;
    rc = pipe_ensure_aux(pp, pool_b + 2 * idx_b + dig_b + scr_b + 16);
    if (rc != HADES252_OK) return call.finish(rc);
    uint8_t *d_pool = (uint8_t *)pp.aux, *d_off = d_pool + pool_b, *d_len = d_off + idx_b, *d_dig = d_len + idx_b;
    uint8_t *d_scr = d_dig + dig_b, *d_bad = d_scr + scr_b;
    if (n_scalars) {                                                 // the pool: through the staging threads when it is big
        size_t cbytes = StagedSource::slot_bytes();                  // and in ordinary memory, else one copy
        rc = call.plan_upload(scalars, n_scalars * 32, &cbytes, 32);
        if (rc != HADES252_OK) return call.finish(rc);
        if (call.src) {
            const size_t total = n_scalars * 32, n_chunks = (total + cbytes - 1) / cbytes;
            for (size_t c = 0; c < n_chunks; c++) {
                const size_t off = c * cbytes, n = total - off < cbytes ? total - off : cbytes;
                const uint8_t *from = call.src->wait(c);
                if (from == nullptr) return call.staged_failure();
                TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(d_pool + off, from, n, hipMemcpyHostToDevice, pp.s_in)));
                TRY_CALL(call, hipEventRecord(pp.in_done[c % kPipeSlots], pp.s_in));
                call.src->enqueued(c);
            }
            TRY_CALL(call, hipStreamWaitEvent(pp.s_k, pp.in_done[(n_chunks - 1) % kPipeSlots], 0));
        } else {
            TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(d_pool, scalars, n_scalars * 32, hipMemcpyHostToDevice, pp.s_k)));
        }
    }
    TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(d_off, offsets, n_msgs * 8, hipMemcpyHostToDevice, pp.s_k)));
    TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(d_len, lengths, n_msgs * 8, hipMemcpyHostToDevice, pp.s_k)));
    TRY_CALL(call, hipMemsetAsync(d_bad, 0, 4, pp.s_k));
    rc = hades252_sponge_hash_var_ex_dev(d_pool, n_scalars, (const uint64_t *)d_off, (const uint64_t *)d_len, n_msgs,
                                         capacity_mont, pad_mode, d_dig, (int *)d_bad, d_scr, scr_b, pp.s_k);
    if (rc != HADES252_OK) return call.finish(rc);
    int bad = 0;
    TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(digests, d_dig, dig_b, hipMemcpyDeviceToHost, pp.s_k)));
    TRY_CALL(call, F(F_MEMCPY, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, pp.s_k)));
    TRY_CALL(call, F(F_SYNC, hipStreamSynchronize(pp.s_k)));
    if (n_bad != nullptr) *n_bad = (size_t)bad;
    return call.finish(HADES252_OK);
}
#undef TRY_CALL

}  // extern "C"
