// host_pool.hpp -- the pool of "pipes" the host-pointer entry points work through (streams, events, chunk buffers, scratch,
// staging), bounded and trimmable (hades252_trim / hades252_pool_bytes).
#pragma once

extern "C" {

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

}  // extern "C"
