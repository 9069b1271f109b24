// host_pipe.hpp -- C ABI, HOST memory in and out: hades252_perm_batch (what a Rust `Strategy::perm` binds), _bytes, _multi,
// warm_up.  Chunks are copied in, permuted and copied out on three streams chained by events; ordinary (pageable) memory
// travels through page-locked staging slots filled and drained by helper threads.
#pragma once

extern "C" {

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
// Never more than the CPUs this process may run on can carry: `workers` concurrent calls (the worker threads of
// hades252_perm_batch_multi, one per device) x 2 directions x threads <= usable CPUs, at least one per direction.
static int usable_cpus() {
    static const int v = []() {
        cpu_set_t set;
        int n = 0;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
        if (n <= 0) n = (int)std::thread::hardware_concurrency();
        return n > 0 ? n : 1;
    }();
    return v;
}
static thread_local int tl_concurrent_workers = 1;     // set by the worker threads of the _multi entry points
static int stage_threads_for(int configured, int cpus, int workers) {
    const int fair = cpus / (2 * (workers < 1 ? 1 : workers));
    const int t = configured < fair ? configured : fair;
    return t < 1 ? 1 : t;
}
static int stage_threads() {
    static const int configured = []() {
        const char *e = getenv("HADES252_STAGE_THREADS");
        int t = e ? atoi(e) : 3;
        return t < 1 ? 1 : (t > kStageSlots ? kStageSlots : t);
    }();
    return stage_threads_for(configured, usable_cpus(), tl_concurrent_workers);
}

// (exported for tuning and tests: the figure a call made by one of `n_workers` concurrent workers would use)
int hades252_stage_threads(int n_workers) {
    const int saved = tl_concurrent_workers;
    tl_concurrent_workers = n_workers < 1 ? 1 : n_workers;
    const int t = stage_threads();
    tl_concurrent_workers = saved;
    return t;
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
            tl_concurrent_workers = n_workers;                          // its staging threads share the CPUs with the others'
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

}  // extern "C"
