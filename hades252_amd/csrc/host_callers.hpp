// host_callers.hpp -- C ABI, HOST memory in: the callers of `perm` as one-shot calls (Merkle root, sharded Merkle root, sponge
// hashes): the input is uploaded in chunks behind the hashing, 32 bytes per tree / message come back.
#pragma once

extern "C" {

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
            tl_concurrent_workers = n_workers;                          // its staging threads share the CPUs with the others'
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
