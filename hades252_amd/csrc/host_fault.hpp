// host_fault.hpp -- what every host-side part of libhades252 shares: the per-thread HIP error behind
// hades252_last_hip_error, HIP_TRY, the fault-injection hook (hades252_fault_inject / HADES252_FAIL_AT) and a
// std::thread spawner that cannot throw across the C boundary.  Host code only; included once by hades252.hip.
#pragma once

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
