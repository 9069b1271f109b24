// host_pin.hpp -- C ABI: page-locked host memory (hades252_host_alloc / _register ...) and the registry that lets the
// host-pointer entry points recognise it.
#pragma once

extern "C" {

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

}  // extern "C"
