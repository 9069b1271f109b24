/*
 * hades252.h -- C ABI of libhades252 (MI355X / gfx950 batched Hades252 permutation).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference (dusk-hades 0.24.1)
 * has no FFI of its own; every entry point below names the reference interface it stands in
 * for (paths relative to the reference crate root).  A Rust `Strategy<BlsScalar>` implementor
 * binds these with a plain `extern "C"` block -- see INTEGRATION.md.
 *
 * Data formats
 *   "Montgomery limbs": the in-memory `BlsScalar` -- 4 x u64 little-endian limbs of
 *       value * 2^256 mod p, fully reduced to [0, p).  A state is WIDTH = 5 scalars = 20 u64 =
 *       160 bytes; a batch is an array of states (AoS), permutation i at u64 offset 20*i.
 *       This is exactly `&mut [BlsScalar]` with len = 5*n (zero-copy from Rust).
 *   "canonical bytes": `BlsScalar::to_bytes()` -- 32 little-endian bytes of the value in [0, p)
 *       (the format used at src/round_constants.rs:61-62).
 *   Inputs that are not fully reduced are outside the reference's type invariant; the limb entry
 *   points do not check them, the byte entry points reject them (HADES252_ERR_NOT_CANONICAL).
 *
 * Conventions
 *   - All functions return HADES252_OK (0) or a negative HADES252_ERR_* code; nothing throws or
 *     unwinds across the boundary (the reference builds with panic = 'abort', Cargo.toml:20).
 *   - The caller owns every buffer.  Host-pointer functions are synchronous and keep no
 *     pointer after returning.  `_dev` functions take device pointers valid on the CURRENT HIP
 *     device, enqueue on `stream` (a hipStream_t, NULL = default stream) and return without
 *     synchronising; buffers must stay alive until the stream has drained.  Device pointers must
 *     be 16-byte aligned (hipMalloc'ed buffers and whole-state offsets into them are); host
 *     pointers need only their natural 8-byte alignment.
 *   - Thread-safe and re-entrant: any number of host threads may call concurrently (the reference's
 *     strategy is a stateless ZST, src/strategies/scalar.rs:11-20).  The only internal state is a
 *     mutex-protected pool of pipes (three streams + chunk buffers in device memory) used by the
 *     host-pointer entry points and the list of host ranges pinned through this library; device
 *     tables are immutable.
 *   - There is no CPU fallback: without a usable HIP device the calls fail with
 *     HADES252_ERR_NO_DEVICE / HADES252_ERR_HIP.
 */
#ifndef HADES252_H
#define HADES252_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HADES252_WIDTH 5              /* src/lib.rs:27 */
#define HADES252_TOTAL_FULL_ROUNDS 8  /* src/lib.rs:21 */
#define HADES252_PARTIAL_ROUNDS 59    /* src/lib.rs:25 */

#define HADES252_OK 0
#define HADES252_ERR_INVALID_ARG (-1)   /* NULL pointer with n > 0, bad round / index / device count */
#define HADES252_ERR_HIP (-2)           /* a HIP runtime call failed; see hades252_last_hip_error() */
#define HADES252_ERR_NOT_CANONICAL (-3) /* a canonical-bytes input encodes a value >= p */
#define HADES252_ERR_NO_DEVICE (-4)     /* no HIP device visible */
#define HADES252_ERR_SCRATCH (-5)       /* scratch buffer too small */
#define HADES252_ERR_OUT_OF_CONSTANTS (-6) /* cursor + WIDTH > 960: the reference panics with
                                              "Hades252 out of ARK constants" (src/strategies.rs:40) */

/* kernel selectors for hades252_perm_batch_dev_ex (all produce identical bits) */
#define HADES252_KERNEL_DEFAULT 0
#define HADES252_KERNEL_LITERAL 1 /* the reference's round structure, 1972 Montgomery products */
#define HADES252_KERNEL_FAST 2    /* scale-tracked small-integer MDS formulation, one state per lane:
                                     the throughput kernel */
#define HADES252_KERNEL_COOP 3    /* same arithmetic, the five words of a state on five waves: ~104 us per
                                     launch up to 16 384 states, ~45 % of the throughput of FAST */
#define HADES252_KERNEL_LANES 4   /* one state per wave, every field element spread over a 16-lane row: the
                                     lowest latency for ONE permutation -- the reference's own call shape
                                     (README.md:60-61): 50 us up to 768 states (helper wave per three states), 54-57 us up
                                     to 1 024 */
#define HADES252_KERNEL_ROWS 5    /* the same lane arithmetic, one state per 16-lane ROW, four per wave, the throughput
                                     kernel's schedule */
/* The size rule of HADES252_KERNEL_DEFAULT lives in ONE place, the library: hades252_kernel_for(n) is the selector
 * hades252_perm_batch_dev(.., n_perms = n, ..) runs (never DEFAULT, never LITERAL); hades252_chain_form_for(n) is the form
 * the chain entry points (sponge, streaming absorb, path verification, tree update) and the Merkle levels run for n
 * chains / parents (one of LANES, ROWS, COOP, FAST -- the per-state arithmetic of that kernel); hades252_kernel_name gives
 * the name of the __global__ function a profiler shows for a selector and batch size ("k_perm_fast", "k_perm_lanes", ..;
 * NULL for an unknown selector). */
int hades252_kernel_for(size_t n_perms);
int hades252_chain_form_for(size_t n_chains);
const char *hades252_kernel_name(int kernel, size_t n_perms);

/* ---- meta --------------------------------------------------------------------------- */
/* Strategy::rounds() (src/strategies.rs:160-162): TOTAL_FULL_ROUNDS + PARTIAL_ROUNDS = 67 */
int hades252_rounds(void);
int hades252_device_count(void);
const char *hades252_strerror(int code);
/* hipError_t of the most recent failing HIP call on this thread (0 if none) */
int hades252_last_hip_error(void);
const char *hades252_version(void);

/* ---- Strategy::perm, batched (src/strategies.rs:140-157 via ScalarStrategy) --------------
 * DELIBERATELY NOT EXPORTED: `hades252_perm_batch_cpu(uint64_t *, size_t, int n_threads)`, which SURVEY.md section 8(b)
 * lists as "the C++ oracle / baseline".  The product has no CPU path at all: a caller that wants the reference's CPU
 * behaviour keeps calling the reference's `ScalarStrategy` (that is what `HipStrategy` delegates the per-operation trait
 * methods to), and the C port that `bench.py` times as `cpu_baseline` lives under oracle/ as
 * test infrastructure -- linking it into this library would make every parity claim circular (the checker would ship
 * inside the thing it checks) and would let a missing GPU go unnoticed.  Do not "fix" this by linking oracle/ in:
 * tests/test_abi.py::test_no_cpu_fallback_in_product fails if anything under hades252_amd/, include/ or rust/ reaches
 * it.  For ONE permutation per call the CPU is the right tool anyway -- see the crossover table in INTEGRATION.md. */
/* In place on host memory, Montgomery limbs.  n_perms == 1 is exactly
 * `ScalarStrategy::new().perm(&mut state)` (README.md:60-61). */
int hades252_perm_batch(uint64_t *states, size_t n_perms);
/* In place on host memory, canonical bytes (5 x 32 B per state). */
int hades252_perm_batch_bytes(uint8_t *states, size_t n_perms);
/* In place on device memory, Montgomery limbs, asynchronous on `stream`. */
int hades252_perm_batch_dev(void *d_states, size_t n_perms, void *stream);
int hades252_perm_batch_dev_ex(void *d_states, size_t n_perms, void *stream, int kernel);
/* Host batch sharded over the first n_devices GPUs (contiguous ranges, one host thread and
 * pipe per device, no collective).  n_devices <= 0 means all visible devices.  A buffer the caller page-locked
 * (hades252_host_alloc / _register: portable, every device sees it) goes straight to DMA on every link; ordinary memory
 * travels through each worker's staging threads (3 + 3 CPU threads per worker). */
int hades252_perm_batch_multi(uint64_t *states, size_t n_perms, int n_devices);
/* Same with n_workers host threads; flags = HADES252_MULTI_VIRTUAL maps worker g to device g % (visible devices)
 * instead of device g, so that n_workers may exceed the device count (up to 64): a one-GPU box then runs the code
 * path of an 8-GPU node, several workers sharing a device.  Worker g owns states [n g / W, n (g+1) / W). */
#define HADES252_MULTI_VIRTUAL 1u
int hades252_perm_batch_multi_ex(uint64_t *states, size_t n_perms, int n_workers, unsigned flags);

/* ---- page-locked host memory for the host-pointer entry points ---------------------------------------
 * The reference's caller holds its states in ordinary memory (`&mut [BlsScalar]`, src/strategies.rs:140).  DMA needs
 * page-locked memory.  A caller with a long-lived buffer pins it ONCE: either allocate it here, or register an existing
 * allocation; hades252_perm_batch* recognise such memory (and memory the caller pinned through HIP itself) and go
 * straight to DMA (93-98 % of the link's bidirectional ceiling from 2^22 states on).  Anything else is never locked
 * when large: batches of more than 2^17 states travel through page-locked staging buffers of the library, filled and
 * drained by helper threads (HADES252_STAGE_THREADS per direction, default 3; ~88-92 % of the page-locked rate, at the
 * price of those CPU cores for the duration of the call); one-chunk batches of 8 MiB and more are page-locked in place
 * for the call; smaller ones are copied as pageable memory.  HADES252_HOST_PIN=0: plain pageable copies throughout.
 *   hades252_host_alloc      bytes of page-locked host memory, usable from every device; free with hades252_host_free
 *   hades252_host_register   page-lock [p, p + bytes) in place (any alignment); undo with hades252_host_unregister(p)
 *   hades252_host_is_pinned  1 if [p, p + bytes) is page-locked (by either call or through HIP), else 0
 * free / unregister return HADES252_ERR_INVALID_ARG for a pointer that did not come from alloc / register. */
int hades252_host_alloc(void **out, size_t bytes);
int hades252_host_free(void *p);
int hades252_host_register(void *p, size_t bytes);
int hades252_host_unregister(void *p);
int hades252_host_is_pinned(const void *p, size_t bytes);

/* ---- device memory for callers that do not link HIP themselves -----------------------------------------
 * A Rust crate binding only this library can keep its data resident and use every *_dev entry point through these:
 *   hades252_dev_alloc / _free   device memory on the current device (hipMalloc / hipFree)
 *   hades252_dev_upload          host -> device on `stream` (NULL = the default stream); asynchronous when the host memory
 *   hades252_dev_download        is page-locked (hades252_host_alloc / _register), device -> host likewise
 *   hades252_stream_create / _destroy / _sync    a stream handle for the `stream` arguments (sync(NULL) = the default one)
 * Pointers and streams are plain HIP objects: a HIP program may mix them with its own. */
int hades252_dev_alloc(void **d_ptr, size_t bytes);
int hades252_dev_free(void *d_ptr);
int hades252_dev_upload(void *d_dst, const void *h_src, size_t bytes, void *stream);
int hades252_dev_download(void *h_dst, const void *d_src, size_t bytes, void *stream);
int hades252_stream_create(void **stream);
int hades252_stream_destroy(void *stream);
int hades252_stream_sync(void *stream);

/* Optional.  Pays the one-time costs of the host-pointer path now instead of inside the first real call: loads the code
 * object (a one-state permutation on an internal buffer: ~35 ms in a fresh process) and, for n_perms_hint > 256, creates
 * the pipe a batch of that size would take -- streams, events, device chunk buffers, and the page-locked staging buffers
 * when the hint is large enough for the staging-thread path of ordinary memory -- and leaves it in the pool.  Current
 * device; a caller of the _multi entry points calls it once per device. */
int hades252_warm_up(size_t n_perms_hint);

/* ---- what the library caches, and how to give it back ---------------------------------------------------------
 * The host-pointer entry points reuse "pipes" (three streams, events, chunk buffers and a scratch arena in device memory,
 * a small page-locked staging buffer) so that a call pays no allocation.  The pool is bounded: per device it keeps at most
 * 16 pipes and at most HADES252_POOL_MAX_BYTES (environment, default 1 GiB) of device memory -- a pipe coming back from
 * a big one-shot call gives up its arena / chunk buffers first.  hades252_trim() destroys every pooled pipe (pipes in
 * use by concurrent calls are untouched and return to the pool later); hades252_pool_bytes() is the device memory the
 * pool holds right now on all devices.  A long-lived process that shares the GPU with another allocator calls trim after
 * a burst of large calls.  NOT counted by either figure: the page-locked HOST memory of the staging-thread path -- 120
 * MiB per pipe that has served a big call from ordinary memory; at most two such buffers stay cached per device (plus
 * one per call in flight) and a staging call reuses a pipe that owns one before allocating another; trim frees them. */
int hades252_trim(void);
size_t hades252_pool_bytes(void);
/* Helper threads per copy direction a host-pointer call on ORDINARY memory uses when it is one of n_workers concurrent
 * calls (the worker threads of the _multi entry points; 1 = a plain hades252_perm_batch): HADES252_STAGE_THREADS
 * (environment, 1 .. 6, default 3), capped so that n_workers x 2 directions x threads never exceeds the CPUs this process
 * may run on (sched_getaffinity), at least 1. */
int hades252_stage_threads(int n_workers);

/* ---- failure contract of the host-pointer entry points, and the hook that tests it -------------------------------
 * On HADES252_ERR_HIP from hades252_perm_batch* every 160-byte state of the caller's buffer holds EITHER its input OR
 * its permutation, never anything else (after a call of one chunk -- up to 65 536 states -- the buffer is either
 * untouched or completely written); which states were completed is unspecified: the call streams chunks in place, so making the whole batch atomic
 * would cost a second copy of it.  Every stream has drained when the call returns, nothing is leaked, the pipe is
 * destroyed rather than pooled, and the next call works.  hades252_merkle_root[_multi] write the root only on
 * success; a failing sponge call may already have delivered the digests of its first chunks.
 * Test hook: hades252_fault_inject("<site>:<nth>") makes the nth (1-based) HIP call of that class made by the
 * library from now on fail as if the runtime had refused it; sites: malloc, hostmalloc, hostregister, memcpy,
 * streamcreate, eventcreate, sync, worker (the device selection of one worker thread of the _multi entry points),
 * thread (the start of a helper thread: staging copies, _multi workers; reported as hipErrorOutOfMemory).
 * NULL or "" disarms.  The environment variable HADES252_FAIL_AT holds the same spec for processes that cannot call
 * the hook (read once, when the library is loaded: a static initialiser, so the variable must be set before dlopen /
 * process start).  Disarmed cost: one relaxed load per wrapped call.
 * HADES252_TEST_MAX_LAUNCH (tests only; read at the first call of hades252_perm_batch_dev*) lowers the number of states
 * one kernel launch of that entry point takes from 2^30, so that its multi-launch loop can be exercised on a few
 * thousand states; values below 256 are ignored. */
int hades252_fault_inject(const char *spec);

/* ---- the callers of perm, host memory in, host memory out -------------------------------------------------
 * One-shot forms of the Merkle root and the fixed-length sponge for data that lives in host memory (page-locked or not,
 * as for hades252_perm_batch: big inputs in ordinary memory go through the library's staging threads, nothing of the
 * caller's is page-locked): the leaves / messages travel to the device in chunks while the previous chunk is being
 * hashed (the first tree level / the sponge itself), only 32 bytes per tree / message travel back.
 *   hades252_merkle_root   leaves: n_leaves x 4 u64 (Montgomery limbs); pad: NULL or depth x 4 u64 (the padding table of
 *                          hades252_merkle_root_pad_dev, host memory); semantics of hades252_merkle_root_pad_dev
 *   hades252_sponge_hash   msgs: n_msgs x msg_len x 4 u64; digests: n_msgs x 4 u64; semantics of hades252_sponge_hash_dev */
int hades252_merkle_root(const uint64_t *leaves, size_t n_leaves, int arity, const uint64_t tag_mont[4], int out_idx,
                         const uint64_t *pad, uint64_t root[4]);
int hades252_sponge_hash(const uint64_t *msgs, size_t n_msgs, size_t msg_len, const uint64_t capacity_mont[4],
                         int pad_mode, uint64_t *digests);
/* The same root with the sub-trees of a FULL tree (n_leaves = arity^k) sharded over n_workers devices (0 = all), no
 * collective: worker g builds its complete sub-trees on device g, the sub-roots (32 B each) meet in host memory and one
 * more small tree finishes.  flags as hades252_perm_batch_multi_ex. */
int hades252_merkle_root_multi(const uint64_t *leaves, size_t n_leaves, int arity, const uint64_t tag_mont[4], int out_idx,
                               int n_workers, unsigned flags, uint64_t root[4]);
/* ... and the variable-length sponge (semantics of hades252_sponge_hash_var_ex_dev with sorting): scalars is the pool
 * (n_scalars x 4 u64), message i = scalars[offsets[i] .. offsets[i] + lengths[i]); *n_bad (may be NULL) receives the
 * number of messages that did not lie inside the pool (hashed as empty messages, never read). */
int hades252_sponge_hash_var(const uint64_t *scalars, size_t n_scalars, const uint64_t *offsets, const uint64_t *lengths,
                             size_t n_msgs, const uint64_t capacity_mont[4], int pad_mode, uint64_t *digests,
                             size_t *n_bad);

/* Per-round trace (witness pre-computation for GadgetStrategy, src/strategies/gadget.rs:41-133):
 * d_trace receives 67 batches, round-major: trace[r] (n_perms x 160 B, same AoS format) is the
 * state of every permutation after round r's mul_matrix; trace[66] equals the perm output.
 * d_states is not modified.  Needs 67 * 160 * n_perms bytes. */
int hades252_perm_trace_dev(const void *d_states, void *d_trace, size_t n_perms, void *stream);
/* kernel: HADES252_KERNEL_FAST (default; radix-2^29 rounds that hold every value in true form, so a round's five words
 * leave through an exact division by 32: ~200 M permutations/s) or HADES252_KERNEL_LITERAL (the reference's schedule,
 * ~35 M/s); identical bits. */
int hades252_perm_trace_dev_ex(const void *d_states, void *d_trace, size_t n_perms, void *stream, int kernel);

/* The same trace in SCALED form (opt-in, ~1.5x the rate: 313 against 215 M permutations/s at 2^20 states): trace[r] holds, fully reduced, the state the throughput kernel
 * carries after round r -- the true state times the running scale of the schedule, without the constants the partial rounds
 * defer -- so a word leaves the kernel without a multiplication.  The consumer recovers
 *     true[r][w] = scaled[r][w] * mul[r] + add[r][w]        (BlsScalar multiplication and addition, in-memory values)
 * with the 67 multipliers and 67 x 5 addends of hades252_perm_trace_scale_table (mul: 67 x 4 u64, add: 67 x 5 x 4 u64, host
 * memory, Montgomery limbs like every BlsScalar here; add[r] is zero in the full rounds) -- lazily, fused into whatever
 * reads the trace next.  Same layout and size as hades252_perm_trace_dev; d_states is not modified. */
int hades252_perm_trace_scaled_dev(const void *d_states, void *d_trace, size_t n_perms, void *stream);
int hades252_perm_trace_scale_table(uint64_t *mul, uint64_t *add);

/* Full gadget witness: every gate output GadgetStrategy assigns for a permutation (src/strategies/gadget.rs:41-133),
 * hades252_witness_wires() = 972 values per state in gate order:
 *   round 0: 5 x (w + c);  every round: per S-boxed word v^2, v^4, v^5 (5 words in a full round, the last word in a
 *   partial round), then for j = 0..4: r1[j] = M[j][0] v0 + M[j][1] v1 + M[j][2] v2 and
 *   r2[j] = M[j][3] v3 + M[j][4] v4 + r1[j] + (next round's constant j, 0 after the last round).
 * d_wires receives 972 batches, wire-major: wires[g] is n_perms scalars of 32 B (Montgomery limbs).  r2 of the last
 * round is the permutation's output.  Needs 972 * 32 * n_perms bytes; d_states is not modified.  ~125 M permutations/s at
 * 2^20 states (3.9 TB/s of wires written). */
int hades252_witness_wires(void);
int hades252_perm_witness_dev(const void *d_states, void *d_wires, size_t n_perms, void *stream);

/* ---- the trait's per-operation methods, batched on device ------------------------------- */
/* Strategy::add_round_key (src/strategies/scalar.rs:23-30).  The trait method takes the constants
 * ITERATOR (src/strategies.rs:33-41, :50-52); `cursor` is its position: word w of every state +=
 * ROUND_CONSTANTS[cursor + w].  Any 0 <= cursor <= 955 of the 960 constants (src/round_constants.rs:18)
 * is accepted; beyond that the reference panics "Hades252 out of ARK constants" and this returns
 * HADES252_ERR_OUT_OF_CONSTANTS.  The `round` forms are cursor = 5*round (what perm() does). */
int hades252_add_round_key_at_dev(void *d_states, size_t n_states, int cursor, void *stream);
int hades252_add_round_key_dev(void *d_states, size_t n_states, int round, void *stream);
/* Strategy::quintic_s_box (src/strategies/scalar.rs:32-34) on n_scalars independent scalars. */
int hades252_quintic_s_box_dev(void *d_scalars, size_t n_scalars, void *stream);
/* Strategy::mul_matrix (src/strategies/scalar.rs:36-49) on every state. */
int hades252_mul_matrix_dev(void *d_states, size_t n_states, void *stream);
/* Strategy::apply_full_round / apply_partial_round (src/strategies.rs:107-119, :79-93). */
int hades252_apply_full_round_dev(void *d_states, size_t n_states, int round, void *stream);
int hades252_apply_partial_round_dev(void *d_states, size_t n_states, int round, void *stream);
int hades252_apply_full_round_at_dev(void *d_states, size_t n_states, int cursor, void *stream);
int hades252_apply_partial_round_at_dev(void *d_states, size_t n_states, int cursor, void *stream);

/* ---- BlsScalar arithmetic, batched (external crate dusk-bls12_381; call sites src/strategies/scalar.rs:28,
 * :33, :44, src/round_constants.rs:41) ------------------------------------------------------------------
 * out[i] = a[i] op b[i] on Montgomery limbs (32 B each, fully reduced in and out; out may alias a or b).
 * op: 0 add, 1 mul, 2 square (b ignored), 3 from_raw (a = canonical limbs, b ignored), 4 reduce_signed (impl 1 only; a =
 *     the 256-bit two's-complement image of a signed integer x in (-p - 2^250, 2^250], b ignored; out = x mod p as plain
 *     limbs: the exit routine of the scaled per-round trace, exposed so that its rare branches can be driven directly).
 * impl: 0 = saturated 8 x u32 arithmetic of the literal kernels, 1 = radix-2^29 signed-limb arithmetic
 * of the shipped kernel.  Both give identical bits; tests regenerate the reference's constant blobs
 * through each of them. */
#define HADES252_FR_ADD 0
#define HADES252_FR_MUL 1
#define HADES252_FR_SQUARE 2
#define HADES252_FR_FROM_RAW 3
#define HADES252_FR_REDUCE_SIGNED 4
int hades252_fr_op_dev(int op, int impl, const void *d_a, const void *d_b, void *d_out, size_t n, void *stream);

/* ---- wire format on device: BlsScalar::from_bytes / to_bytes ------------------------------ */
/* d_bad_count (device int, may be NULL) is incremented once per input >= p; such inputs
 * produce an all-zero scalar. */
int hades252_from_bytes_dev(const void *d_bytes, void *d_limbs, size_t n_scalars, int *d_bad_count, void *stream);
int hades252_to_bytes_dev(const void *d_limbs, void *d_bytes, size_t n_scalars, void *stream);

/* ---- Poseidon Merkle trees over the permutation (caller shape of dusk-poseidon, README.md:9) -------
 * parent = perm([tag, child_0 .. child_{arity-1}, 0 ..])[out_idx]; tag in Montgomery limbs.  The external convention
 * for arity 4 is tag = 2^4 - 1 = 15, out_idx = 1 (NOT pinned by the reference: both are parameters).  Digests are 32 B
 * Montgomery limbs.
 * Shapes: arity 2, 3 or 4 for trees (1 .. 4 for single levels and path verification); ANY number of leaves >= 2.
 *   Level l (l = 0: the leaves) has n_l nodes, n_{l+1} = ceil(n_l / arity), down to one root: hades252_merkle_depth
 *   levels above the leaves.  Padding rule: a child position past the end of level l holds pad[l], a table of `depth`
 *   digests in DEVICE memory the caller supplies to the _pad entry points (NULL = the zero scalar at every level);
 *   hades252_merkle_empty_digests_dev fills it with the roots of empty subtrees (pad[0] = e0, pad[l+1] = the parent of
 *   `arity` copies of pad[l]) -- the table an append-only tree uses.  Trees whose leaf count is a power of the arity
 *   never touch the table. */
int hades252_merkle_depth(size_t n_leaves, int arity);      /* -1: invalid (arity not 2..4, or fewer than 2 leaves) */
/* One level.  _dev: n_parents full parents from arity * n_parents children (arity 1 .. 4);
 * _pad_dev: n_children children -> ceil(n_children / arity) parents, missing children = the digest at d_pad (32 B). */
int hades252_merkle_level_dev(const void *d_children, void *d_parents, size_t n_parents, int arity,
                              const uint64_t tag_mont[4], int out_idx, void *stream);
int hades252_merkle_level_pad_dev(const void *d_children, size_t n_children, void *d_parents, int arity,
                                  const uint64_t tag_mont[4], int out_idx, const void *d_pad, void *stream);
/* Root only.  d_scratch needs hades252_merkle_scratch_bytes(n_leaves, arity) bytes (0 for a one-level tree;
 * d_scratch may then be NULL); the root (32 B) is written to d_root.  Levels of more than 16 384 parents run one
 * parent per lane; full levels of 1 025 .. 16 384 parents five waves per parent (arity 2 / 4 and power-of-arity levels:
 * 64 parents per workgroup through several levels in LDS); levels of at most 1 024 parents one parent per wave. */
size_t hades252_merkle_scratch_bytes(size_t n_leaves, int arity);
int hades252_merkle_root_dev(const void *d_leaves, size_t n_leaves, int arity, void *d_scratch, size_t scratch_bytes,
                             const uint64_t tag_mont[4], int out_idx, void *d_root, void *stream);
int hades252_merkle_root_pad_dev(const void *d_leaves, size_t n_leaves, int arity, void *d_scratch, size_t scratch_bytes,
                                 const uint64_t tag_mont[4], int out_idx, const void *d_pad, void *d_root, void *stream);
/* Whole tree, every level kept: d_tree (hades252_merkle_tree_bytes = 32 * (n_1 + n_2 + ... + 1) bytes; for
 * n_leaves = arity^k that is 32 * (n_leaves - 1) / (arity - 1)) receives level 1, then level 2, ... ; the root is its
 * last 32 bytes.  0 = invalid shape. */
size_t hades252_merkle_tree_bytes(size_t n_leaves, int arity);
int hades252_merkle_build_dev(const void *d_leaves, size_t n_leaves, int arity, const uint64_t tag_mont[4], int out_idx,
                              void *d_tree, void *stream);
int hades252_merkle_build_pad_dev(const void *d_leaves, size_t n_leaves, int arity, const uint64_t tag_mont[4], int out_idx,
                                  const void *d_pad, void *d_tree, void *stream);
/* The padding table of empty subtrees: d_pad[0] = e0, d_pad[l+1] = parent of `arity` copies of d_pad[l], l < depth - 1. */
int hades252_merkle_empty_digests_dev(int arity, int depth, const uint64_t e0_mont[4], const uint64_t tag_mont[4],
                                      int out_idx, void *d_pad, void *stream);
/* Openings (authentication paths) from a built tree: for query t with leaf index d_indices[t] (device u64) and
 * level l = 0 .. depth-1, the arity-1 siblings of the path node, in child order with the node's own position
 * (index / arity^l) % arity skipped: d_paths[t][l][s], 32 B each, depth * (arity-1) * 32 bytes per query; a sibling
 * position past the end of its level reads pad[l].  An index >= n_leaves yields an all-zero path (nothing outside the
 * tree is read). */
int hades252_merkle_open_dev(const void *d_leaves, const void *d_tree, size_t n_leaves, int arity,
                             const uint64_t *d_indices, size_t n_queries, void *d_paths, void *stream);
int hades252_merkle_open_pad_dev(const void *d_leaves, const void *d_tree, size_t n_leaves, int arity,
                                 const uint64_t *d_indices, size_t n_queries, const void *d_pad, void *d_paths,
                                 void *stream);
/* Batched path verification: d_roots[t] (32 B) = the root recomputed from leaf value d_leaves[t] (32 B each, query
 * order), its index d_indices[t] and its opening d_paths[t] (the layout above); the caller compares with the root it
 * trusts.  `depth` dependent permutations per query; arity 1 .. 4.  One query per lane; up to 1 024 queries one query per
 * WAVE (the low-latency form: 12 levels in 0.6 ms instead of 1.9), up to 4 096 four queries per wave (1.0 ms), up to 16 384 five
 * waves per query (1.3 ms). */
int hades252_merkle_verify_dev(const void *d_leaves, const uint64_t *d_indices, const void *d_paths, size_t n_queries,
                               int depth, int arity, const uint64_t tag_mont[4], int out_idx, void *d_roots, void *stream);
/* Incremental update (the append / overwrite path of a Merkle-tree caller: README.md:9 names the tree, the update is the
 * operation it serves): the caller has overwritten the leaves d_leaves[d_indices[q]], q < n_updates (device u64; sorted
 * lists do each ancestor once, any order is correct, an index >= n_leaves is ignored); their ancestors in d_tree -- built
 * by hades252_merkle_build[_pad]_dev with the same arity, tag, out_idx and pad -- are recomputed bottom-up in place:
 * at most depth * n_updates permutations, one launch per level, the form per level by hades252_chain_form_for (one
 * ancestor per wave for up to 1 024 updates: ~51 us a level).  A level with no more parents than updates is recomputed whole. */
int hades252_merkle_update_dev(const void *d_leaves, void *d_tree, size_t n_leaves, int arity, const uint64_t tag_mont[4],
                               int out_idx, const void *d_pad, const uint64_t *d_indices, size_t n_updates, void *stream);
/* Forest: n_trees independent trees of leaves_per_tree = arity^k leaves each (leaves contiguous, tree after tree); level l
 * of all trees is one launch; d_roots receives n_trees roots.  Scratch: hades252_merkle_forest_scratch_bytes. */
size_t hades252_merkle_forest_scratch_bytes(size_t n_trees, size_t leaves_per_tree, int arity);
int hades252_merkle_forest_dev(const void *d_leaves, size_t n_trees, size_t leaves_per_tree, int arity, void *d_scratch,
                               size_t scratch_bytes, const uint64_t tag_mont[4], int out_idx, void *d_roots, void *stream);
/* arity-4 forms (BASELINE config 4).  hades252_merkle4_scratch_bytes is 0 both for a one-level tree (4 leaves need no
 * scratch) and for an invalid shape: hades252_merkle_depth(n, 4) < 1 tells the second from the first. */
int hades252_merkle4_level_dev(const void *d_children, void *d_parents, size_t n_parents,
                               const uint64_t tag_mont[4], int out_idx, void *stream);
size_t hades252_merkle4_scratch_bytes(size_t n_leaves);
int hades252_merkle4_root_dev(const void *d_leaves, size_t n_leaves, void *d_scratch, size_t scratch_bytes,
                              const uint64_t tag_mont[4], int out_idx, void *d_root, void *stream);

/* ---- batched sponge (caller shape of dusk-poseidon's sponge hash, README.md:9) ----
 * d_msgs: n_msgs messages of msg_len scalars each (Montgomery limbs, contiguous: message i at scalar
 * offset i*msg_len).  state = [capacity, 0, 0, 0, 0]; each block of 4 scalars is added to words 1..4,
 * then permuted; pad_mode 1 first appends a single scalar 1 (then zeros), pad_mode 0 zero-fills;
 * at least one permutation is always applied.  Digest i = word 1 (32 B) -> d_digests[i].
 * dusk-poseidon is not part of the reference tree: capacity and padding are parameters and this
 * entry point's parity is pinned to this repo's oracle only.
 * A sponge is a chain of dependent permutations per message: batches of up to 1 024 messages (all sponge entry points,
 * hades252_sponge_absorb_dev included) run one message per WAVE on the low-latency arithmetic, ~50 us per block -- ONE
 * long message hashes at the speed of a CPU core (52 us per block) instead of 160 us per block; up to 4 096 four messages per
 * wave (~80 us per block), up to 16 384 five waves per message (~105 us); larger batches run one message per lane (throughput). */
int hades252_sponge_hash_dev(const void *d_msgs, size_t n_msgs, size_t msg_len, const uint64_t capacity_mont[4],
                             int pad_mode, void *d_digests, void *stream);
/* Variable-length batch: message i = d_scalars[d_offsets[i] .. d_offsets[i] + d_lengths[i]) (offsets and
 * lengths in scalars, device arrays of n_msgs u64; messages may overlap or leave gaps; length 0 allowed).
 * n_scalars = size of the pool: a message that does not lie inside it is never read -- it is hashed as the empty
 * message and d_bad_count (device int, may be NULL) is incremented.
 * Same absorption / padding rule per message as above (CONVENTION UNPINNED: parameters, see above).
 * Every lane of a wave runs to the longest message among the wave's 64.  _ex with d_scratch != NULL
 * (hades252_sponge_sort_scratch_bytes(n_msgs) bytes of device memory) first sorts the message indices by block count
 * on the device (three small launches), so that a wave's messages are alike: ragged batches keep > 90 % of the lanes
 * on useful permutations instead of ~50 %.  The digests are the same and land in message order either way. */
int hades252_sponge_hash_var_dev(const void *d_scalars, size_t n_scalars, const uint64_t *d_offsets,
                                 const uint64_t *d_lengths, size_t n_msgs, const uint64_t capacity_mont[4], int pad_mode,
                                 void *d_digests, int *d_bad_count, void *stream);
size_t hades252_sponge_sort_scratch_bytes(size_t n_msgs);
int hades252_sponge_hash_var_ex_dev(const void *d_scalars, size_t n_scalars, const uint64_t *d_offsets,
                                    const uint64_t *d_lengths, size_t n_msgs, const uint64_t capacity_mont[4], int pad_mode,
                                    void *d_digests, int *d_bad_count, void *d_scratch, size_t scratch_bytes, void *stream);
/* Streaming sponge: the states (160 B each, the AoS format of the perm entry points) live in device memory between
 * calls, so a caller can keep absorbing.  init: state = [capacity, 0, 0, 0, 0]; absorb: for t < blocks_each, words
 * 1..4 += d_blocks[i][t][0..3] (4 scalars = 128 B per block, state-major), then the permutation; squeeze: d_digests[i]
 * = word `word` of state i (dusk-poseidon's sponge returns word 1).  Padding is the caller's: the final block carries
 * whatever padding scalars the convention wants.  init + absorb of ceil(len / 4) zero-filled blocks + squeeze(1)
 * equals hades252_sponge_hash_dev with pad_mode 0. */
int hades252_sponge_init_dev(void *d_states, size_t n_states, const uint64_t capacity_mont[4], void *stream);
int hades252_sponge_absorb_dev(void *d_states, const void *d_blocks, size_t n_states, int blocks_each, void *stream);
int hades252_sponge_squeeze_dev(const void *d_states, void *d_digests, size_t n_states, int word, void *stream);
/* Named parameter sets a dusk-poseidon caller would pass -- NAMED, NOT PINNED: that crate is outside the reference
 * tree (README.md:9 only names it); the values below are the commonly described conventions, to be checked against
 * the crate version in use:
 *   "sponge/pad10"   capacity = 2^64 (Montgomery form), pad_mode 1 (a single 1, then zeros), digest = word 1
 *   "merkle/arity4"  tag = 2^4 - 1 = 15 in word 0, children in words 1..4, digest = word 1 */

/* ---- synthetic inputs and digests (benchmark / verification plumbing) --------------------- */
/* Generator B: scalar e (global element index first_elem + k) gets 4 splitmix64 limbs, top limb
 * masked to 62 bits (always < p); see DESIGN.md.  Stateless, so shards generate independently. */
int hades252_gen_b_dev(void *d_scalars, uint64_t first_elem, size_t n_elems, uint64_t seed, void *stream);
/* Generator A: scalar e has value first_elem + k (Montgomery form). */
int hades252_gen_a_dev(void *d_scalars, uint64_t first_elem, size_t n_elems, void *stream);
/* Position-sensitive 256-bit digest of n_u64 words: with g = first_index + i the global index
 * of word i, out[g % 4] += mix(word_i, g) (mod 2^64).  Additive over disjoint ranges, so
 * shards combine by limb-wise wrapping addition.  d_out4 = 4 device u64, zeroed by the call. */
int hades252_digest_dev(const void *d_words, uint64_t first_index, size_t n_u64, void *d_out4, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HADES252_H */
