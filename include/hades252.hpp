// hades252.hpp -- C++ host-side mirror of the reference's operator interface, over the C ABI.
//
// The reference is compiled code (Rust) and this image has no Rust toolchain, so the host side
// above the C ABI is written in C++.  Names, argument meaning and failure behaviour follow
//   pub trait Strategy<T>           reference src/strategies.rs:31-163
//   pub struct ScalarStrategy       reference src/strategies/scalar.rs:11-50
//   WIDTH / TOTAL_FULL_ROUNDS / PARTIAL_ROUNDS   reference src/lib.rs:20-27
// batched: where the reference takes `&mut [BlsScalar]` of exactly WIDTH words, these methods take
// any whole number of WIDTH-word states and apply the operation to each, in place.
// Header-only; link with -lhades252 (and the HIP runtime for device buffers).
//
// Size switch: the Rust binding (rust/src/hip.rs, MIN_GPU_STATES = 2) sends a call that carries ONE state to the
// reference's own CPU `ScalarStrategy` (one GPU call costs ~65 us, one CPU permutation ~50 us).  This mirror has NO
// CPU path, on purpose -- there is no reference CPU code in a C++ caller's crate to delegate to, and a CPU leg inside
// this library would be an oracle in the product: every call here goes to the device or throws.
#ifndef HADES252_HPP
#define HADES252_HPP

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "hades252.h"

namespace dusk_hades {

constexpr std::size_t WIDTH = HADES252_WIDTH;                          // src/lib.rs:27
constexpr std::size_t TOTAL_FULL_ROUNDS = HADES252_TOTAL_FULL_ROUNDS;  // src/lib.rs:21
constexpr std::size_t PARTIAL_ROUNDS = HADES252_PARTIAL_ROUNDS;        // src/lib.rs:25

// In-memory BlsScalar of dusk-bls12_381: 4 x u64 little-endian Montgomery limbs.
struct BlsScalar {
    std::uint64_t limbs[4];
};
static_assert(sizeof(BlsScalar) == 32, "BlsScalar must be 32 bytes");

// The reference panics (and aborts: Cargo.toml:20); the mirror throws.
struct HadesPanic : std::runtime_error {
    int code;
    HadesPanic(const std::string &what, int c) : std::runtime_error(what), code(c) {}
};

inline void check(int rc, const char *where) {
    if (rc != HADES252_OK)
        throw HadesPanic(std::string(where) + ": " + hades252_strerror(rc), rc);
}

// `ROUND_CONSTANTS.iter()` (src/strategies.rs:141): a cursor into the device-resident table.
struct RoundConstantsIter {
    std::size_t pos = 0;
    static constexpr std::size_t CONSTANTS = 960;          // src/round_constants.rs:18
};

// A device-resident batch of states (caller-owned memory on the current HIP device).
struct DeviceStates {
    void *ptr;
    std::size_t n_states;
    void *stream;   // hipStream_t, nullptr = default stream
};

template <typename T>
class Strategy {
public:
    virtual ~Strategy() = default;

    // src/strategies.rs:33-41
    static std::size_t next_c(RoundConstantsIter &constants) {
        if (constants.pos >= RoundConstantsIter::CONSTANTS) throw HadesPanic("Hades252 out of ARK constants", -1);
        return constants.pos++;
    }
    virtual void add_round_key(RoundConstantsIter &constants, T words) = 0;   // src/strategies.rs:50-52
    virtual void quintic_s_box(T value) = 0;                                  // src/strategies.rs:59
    virtual void mul_matrix(RoundConstantsIter &constants, T values) = 0;     // src/strategies.rs:63-65
    virtual void apply_partial_round(RoundConstantsIter &constants, T words) = 0;   // :79-93
    virtual void apply_full_round(RoundConstantsIter &constants, T words) = 0;      // :107-119
    virtual void perm(T data) = 0;                                                  // :140-157
    static std::size_t rounds() { return TOTAL_FULL_ROUNDS + PARTIAL_ROUNDS; }      // :160-162
};

// GPU-backed ScalarStrategy.  Stateless like the reference's zero-sized struct.
class ScalarStrategy : public Strategy<DeviceStates> {
    // any cursor is legal (src/strategies.rs:33-41); past the 960 constants the reference panics
    static int cursor_of(RoundConstantsIter &c) {
        if (c.pos + WIDTH > RoundConstantsIter::CONSTANTS) throw HadesPanic("Hades252 out of ARK constants", -6);
        return static_cast<int>(c.pos);
    }
    static void advance(RoundConstantsIter &c) {
        for (std::size_t i = 0; i < WIDTH; i++) next_c(c);
    }

public:
    static ScalarStrategy new_() { return ScalarStrategy(); }   // src/strategies/scalar.rs:17-19

    void add_round_key(RoundConstantsIter &constants, DeviceStates w) override {   // scalar.rs:23-30
        check(hades252_add_round_key_at_dev(w.ptr, w.n_states, cursor_of(constants), w.stream),
              "add_round_key");
        advance(constants);
    }
    // every 32-byte scalar of the batch (n_states counts scalars here)
    void quintic_s_box(DeviceStates v) override {                                   // scalar.rs:32-34
        check(hades252_quintic_s_box_dev(v.ptr, v.n_states, v.stream), "quintic_s_box");
    }
    void mul_matrix(RoundConstantsIter &, DeviceStates v) override {                // scalar.rs:36-49
        check(hades252_mul_matrix_dev(v.ptr, v.n_states, v.stream), "mul_matrix");
    }
    void apply_partial_round(RoundConstantsIter &constants, DeviceStates w) override {
        check(hades252_apply_partial_round_at_dev(w.ptr, w.n_states, cursor_of(constants), w.stream),
              "apply_partial_round");
        advance(constants);
    }
    void apply_full_round(RoundConstantsIter &constants, DeviceStates w) override {
        check(hades252_apply_full_round_at_dev(w.ptr, w.n_states, cursor_of(constants), w.stream),
              "apply_full_round");
        advance(constants);
    }
    void perm(DeviceStates data) override {
        check(hades252_perm_batch_dev(data.ptr, data.n_states, data.stream), "perm");
    }

    // `strategy.perm(&mut state)` on host memory: len must be a multiple of WIDTH
    // (the reference panics for len != WIDTH, scalar.rs:48).
    void perm(BlsScalar *data, std::size_t len) {
        if (len % WIDTH != 0) throw HadesPanic("perm: slice length is not a multiple of WIDTH", -1);
        check(hades252_perm_batch(reinterpret_cast<std::uint64_t *>(data), len / WIDTH), "perm");
    }
};

// A batch of states in page-locked host memory (hades252_host_alloc): `perm` on it goes straight to DMA instead of
// page-locking the slice inside every call.  Movable, frees on destruction.
class PinnedStates {
    BlsScalar *p_ = nullptr;
    std::size_t len_ = 0;

public:
    explicit PinnedStates(std::size_t n_states) : len_(n_states * WIDTH) {
        void *raw = nullptr;
        check(hades252_host_alloc(&raw, (len_ ? len_ : 1) * sizeof(BlsScalar)), "host_alloc");
        p_ = static_cast<BlsScalar *>(raw);
    }
    PinnedStates(const PinnedStates &) = delete;
    PinnedStates &operator=(const PinnedStates &) = delete;
    PinnedStates(PinnedStates &&o) noexcept : p_(o.p_), len_(o.len_) { o.p_ = nullptr; o.len_ = 0; }
    ~PinnedStates() { if (p_) (void)hades252_host_free(p_); }
    BlsScalar *data() { return p_; }
    std::size_t len() const { return len_; }          // scalars: WIDTH per state
};

// Device memory through the library alone (hades252_dev_alloc / _upload / _download): a caller that does not link HIP
// keeps its data resident and feeds the `DeviceStates` of the strategy above.  Movable, frees on destruction.
class DeviceBuffer {
    void *p_ = nullptr;
    std::size_t bytes_ = 0;

public:
    explicit DeviceBuffer(std::size_t bytes) : bytes_(bytes) { check(hades252_dev_alloc(&p_, bytes ? bytes : 16), "dev_alloc"); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    DeviceBuffer(DeviceBuffer &&o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
    ~DeviceBuffer() { if (p_) (void)hades252_dev_free(p_); }
    void *ptr() const { return p_; }
    std::size_t bytes() const { return bytes_; }
    void upload(const void *host, std::size_t bytes, void *stream = nullptr) { check(hades252_dev_upload(p_, host, bytes, stream), "dev_upload"); }
    void download(void *host, std::size_t bytes, void *stream = nullptr) const {
        check(hades252_dev_download(host, p_, bytes, stream), "dev_download");
        check(hades252_stream_sync(stream), "stream_sync");
    }
};

// The callers of `perm` on host memory (dusk-poseidon's node and sponge shapes, README.md:9; tag / out_idx / capacity /
// padding are that crate's convention and parameters here).
inline BlsScalar merkle_root(const BlsScalar *leaves, std::size_t n_leaves, int arity, const BlsScalar &tag, int out_idx = 1,
                             const BlsScalar *pad = nullptr) {
    BlsScalar root{};
    check(hades252_merkle_root(reinterpret_cast<const std::uint64_t *>(leaves), n_leaves, arity, tag.limbs, out_idx,
                               reinterpret_cast<const std::uint64_t *>(pad), root.limbs), "merkle_root");
    return root;
}

inline void sponge_hash(const BlsScalar *msgs, std::size_t n_msgs, std::size_t msg_len, const BlsScalar &capacity, bool pad_one,
                        BlsScalar *digests) {
    check(hades252_sponge_hash(reinterpret_cast<const std::uint64_t *>(msgs), n_msgs, msg_len, capacity.limbs, pad_one ? 1 : 0,
                               reinterpret_cast<std::uint64_t *>(digests)), "sponge_hash");
}

// What the library caches (pipes: streams, chunk buffers, staging memory) and which kernel a batch size gets.
inline void trim() { check(hades252_trim(), "trim"); }
inline std::size_t pool_bytes() { return hades252_pool_bytes(); }
/// Pay the one-time costs (code object, pipe for a host batch of `n_perms_hint` states) before the first real call.
inline void warm_up(std::size_t n_perms_hint = 0) { check(hades252_warm_up(n_perms_hint), "warm_up"); }
inline int kernel_for(std::size_t n_perms) { return hades252_kernel_for(n_perms); }
inline const char *kernel_name(int kernel, std::size_t n_perms) { return hades252_kernel_name(kernel, n_perms); }

}  // namespace dusk_hades
#endif
