/*
 * hades_oracle.c -- CPU restatement of the reference's ScalarStrategy::perm.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hades252_amd/ may link, load or call this file;
 * it is used by tests/, by __graft_entry__.smoke() as the checker, and by bench.py's
 * `cpu_baseline` leg ("kind": "port").
 *
 * PARITY STATUS: unpinned by the reference.  The reference cannot be built here (no
 * cargo/rustc; its field arithmetic is the un-vendored crate dusk-bls12_381 = "0.13",
 * Cargo.toml:12) and its own tests hold no known-answer vectors.  This file restates the
 * published algorithm of that crate's `Scalar` (4 x u64 little-endian limbs, Montgomery
 * form with R = 2^256, schoolbook multiply + word-by-word Montgomery reduction, results
 * always fully reduced to [0, p)) and is pinned by (1) sha256 equality of the regenerated
 * constant blobs with assets/ark.bin / assets/mds.bin, (2) bit-for-bit agreement with the
 * big-integer specification oracle oracle/hades_spec.py, (3) the anchors of SURVEY.md 8(a).
 *
 * What each function follows (paths relative to /root/reference):
 *   fr_add            BlsScalar `+=`            call site src/strategies/scalar.rs:28,44
 *   fr_mul            BlsScalar `*`             call site src/strategies/scalar.rs:33,44
 *   fr_square         BlsScalar::square         call site src/strategies/scalar.rs:33
 *   fr_from_raw       BlsScalar::from_raw       call site src/round_constants.rs:41, src/mds_matrix.rs:33
 *   load_constants    src/round_constants.rs:29-48, src/mds_matrix.rs:18-40, src/lib.rs:33-44
 *   add_round_key     src/strategies/scalar.rs:23-30   (cursor: src/strategies.rs:33-41)
 *   quintic_s_box     src/strategies/scalar.rs:32-34
 *   mul_matrix        src/strategies/scalar.rs:36-49
 *   apply_full_round  src/strategies.rs:107-119
 *   apply_partial_round src/strategies.rs:79-93
 *   hades_oracle_perm src/strategies.rs:140-157
 */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "hades_oracle_constants.h"

#define WIDTH 5              /* src/lib.rs:27 */
#define TOTAL_FULL_ROUNDS 8  /* src/lib.rs:21 */
#define PARTIAL_ROUNDS 59    /* src/lib.rs:25 */

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr_t;

/* p, little-endian u64 limbs (src/strategies.rs:14) */
static const uint64_t MODULUS[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL,
                                    0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
/* -p^{-1} mod 2^64 */
static const uint64_t INV = 0xfffffffeffffffffULL;
/* R^2 mod p, used by from_raw */
static const fr_t R2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL,
                         0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}};

static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t *carry) {
    u128 t = (u128)a + b + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t *borrow) {
    u128 t = (u128)a - b - (*borrow >> 63);
    *borrow = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
static inline uint64_t mac(uint64_t a, uint64_t b, uint64_t c, uint64_t *carry) {
    u128 t = (u128)a + (u128)b * c + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}

/* a - p if a >= p else a  (constant-time style: subtract, add back masked modulus) */
static inline fr_t fr_sub_modulus(const uint64_t a[4], uint64_t top) {
    uint64_t borrow = 0;
    fr_t d;
    d.l[0] = sbb(a[0], MODULUS[0], &borrow);
    d.l[1] = sbb(a[1], MODULUS[1], &borrow);
    d.l[2] = sbb(a[2], MODULUS[2], &borrow);
    d.l[3] = sbb(a[3], MODULUS[3], &borrow);
    (void)sbb(top, 0, &borrow);
    uint64_t mask = borrow, carry = 0; /* borrow is 0 or all-ones */
    d.l[0] = adc(d.l[0], MODULUS[0] & mask, &carry);
    d.l[1] = adc(d.l[1], MODULUS[1] & mask, &carry);
    d.l[2] = adc(d.l[2], MODULUS[2] & mask, &carry);
    d.l[3] = adc(d.l[3], MODULUS[3] & mask, &carry);
    return d;
}

fr_t fr_add(fr_t a, fr_t b) {
    uint64_t c = 0, s[4];
    s[0] = adc(a.l[0], b.l[0], &c);
    s[1] = adc(a.l[1], b.l[1], &c);
    s[2] = adc(a.l[2], b.l[2], &c);
    s[3] = adc(a.l[3], b.l[3], &c);
    return fr_sub_modulus(s, c); /* 2p < 2^256, so c == 0 for reduced inputs */
}

/* 512-bit -> Montgomery reduction, one 64-bit word at a time */
static inline fr_t montgomery_reduce(uint64_t r[8]) {
    uint64_t carry2 = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t k = r[i] * INV, carry = 0;
        (void)mac(r[i], k, MODULUS[0], &carry);
        r[i + 1] = mac(r[i + 1], k, MODULUS[1], &carry);
        r[i + 2] = mac(r[i + 2], k, MODULUS[2], &carry);
        r[i + 3] = mac(r[i + 3], k, MODULUS[3], &carry);
        r[i + 4] = adc(r[i + 4], carry2, &carry);
        carry2 = carry;
    }
    return fr_sub_modulus(&r[4], carry2);
}

fr_t fr_mul(fr_t a, fr_t b) {
    uint64_t r[8] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) r[i + j] = mac(r[i + j], a.l[i], b.l[j], &carry);
        r[i + 4] = carry;
    }
    return montgomery_reduce(r);
}

fr_t fr_square(fr_t a) {
    /* off-diagonal products once, doubled, plus the diagonal */
    uint64_t r[8] = {0}, carry;
    carry = 0;
    r[1] = mac(0, a.l[0], a.l[1], &carry);
    r[2] = mac(0, a.l[0], a.l[2], &carry);
    r[3] = mac(0, a.l[0], a.l[3], &carry);
    r[4] = carry;
    carry = 0;
    r[3] = mac(r[3], a.l[1], a.l[2], &carry);
    r[4] = mac(r[4], a.l[1], a.l[3], &carry);
    r[5] = carry;
    carry = 0;
    r[5] = mac(r[5], a.l[2], a.l[3], &carry);
    r[6] = carry;
    r[7] = r[6] >> 63;
    r[6] = (r[6] << 1) | (r[5] >> 63);
    r[5] = (r[5] << 1) | (r[4] >> 63);
    r[4] = (r[4] << 1) | (r[3] >> 63);
    r[3] = (r[3] << 1) | (r[2] >> 63);
    r[2] = (r[2] << 1) | (r[1] >> 63);
    r[1] = r[1] << 1;
    carry = 0;
    r[0] = mac(0, a.l[0], a.l[0], &carry);
    r[1] = adc(r[1], 0, &carry);
    r[2] = mac(r[2], a.l[1], a.l[1], &carry);
    r[3] = adc(r[3], 0, &carry);
    r[4] = mac(r[4], a.l[2], a.l[2], &carry);
    r[5] = adc(r[5], 0, &carry);
    r[6] = mac(r[6], a.l[3], a.l[3], &carry);
    r[7] = adc(r[7], 0, &carry);
    return montgomery_reduce(r);
}

/* from_raw: canonical integer -> Montgomery form = mont_mul(v, R^2) */
fr_t fr_from_raw(const uint64_t v[4]) {
    fr_t a;
    memcpy(a.l, v, sizeof a.l);
    return fr_mul(a, R2);
}

/* Montgomery form -> canonical integer (what to_bytes() serialises, little-endian) */
fr_t fr_to_canonical(fr_t a) {
    uint64_t r[8] = {a.l[0], a.l[1], a.l[2], a.l[3], 0, 0, 0, 0};
    return montgomery_reduce(r);
}

static int fr_is_canonical(const uint64_t v[4]) { /* v < p ? */
    uint64_t borrow = 0;
    (void)sbb(v[0], MODULUS[0], &borrow);
    (void)sbb(v[1], MODULUS[1], &borrow);
    (void)sbb(v[2], MODULUS[2], &borrow);
    (void)sbb(v[3], MODULUS[3], &borrow);
    return (borrow >> 63) != 0;
}

/* ---- constants: the loaders of src/round_constants.rs and src/mds_matrix.rs ---------- */
static fr_t ROUND_CONSTANTS[HADES_ORACLE_N_CONSTANTS];
static fr_t MDS_MATRIX[WIDTH][WIDTH];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void load_constants(void) {
    for (int i = 0; i < HADES_ORACLE_N_CONSTANTS; i++) ROUND_CONSTANTS[i] = fr_from_raw(HADES_ARK_RAW[i]);
    for (int i = 0; i < WIDTH; i++)
        for (int j = 0; j < WIDTH; j++) MDS_MATRIX[i][j] = fr_from_raw(HADES_MDS_RAW[i][j]);
}
void hades_oracle_init(void) { pthread_once(&g_once, load_constants); }

/* ---- the strategy ------------------------------------------------------------------- */
static inline void add_round_key(const fr_t **cursor, fr_t *words) {
    for (int w = 0; w < WIDTH; w++) words[w] = fr_add(words[w], *(*cursor)++);
}
static inline fr_t quintic_s_box(fr_t v) { return fr_mul(fr_square(fr_square(v)), v); }

static inline void mul_matrix(fr_t *values) {
    fr_t result[WIDTH];
    memset(result, 0, sizeof result);
    for (int j = 0; j < WIDTH; j++)
        for (int k = 0; k < WIDTH; k++) result[k] = fr_add(result[k], fr_mul(MDS_MATRIX[k][j], values[j]));
    memcpy(values, result, sizeof result);
}
static inline void apply_partial_round(const fr_t **cursor, fr_t *words) {
    add_round_key(cursor, words);
    words[WIDTH - 1] = quintic_s_box(words[WIDTH - 1]);
    mul_matrix(words);
}
static inline void apply_full_round(const fr_t **cursor, fr_t *words) {
    add_round_key(cursor, words);
    for (int w = 0; w < WIDTH; w++) words[w] = quintic_s_box(words[w]);
    mul_matrix(words);
}

/* One permutation, in place, on 20 u64 Montgomery limbs.  `trace` (may be NULL) receives the
 * state after every round: 67 x 20 u64. */
void hades_oracle_perm_trace(uint64_t *state, uint64_t *trace) {
    hades_oracle_init();
    fr_t words[WIDTH];
    memcpy(words, state, sizeof words);
    const fr_t *cursor = ROUND_CONSTANTS;
    int r = 0;
    for (int i = 0; i < TOTAL_FULL_ROUNDS / 2; i++, r++) {
        apply_full_round(&cursor, words);
        if (trace) memcpy(trace + 20 * r, words, sizeof words);
    }
    for (int i = 0; i < PARTIAL_ROUNDS; i++, r++) {
        apply_partial_round(&cursor, words);
        if (trace) memcpy(trace + 20 * r, words, sizeof words);
    }
    for (int i = 0; i < TOTAL_FULL_ROUNDS / 2; i++, r++) {
        apply_full_round(&cursor, words);
        if (trace) memcpy(trace + 20 * r, words, sizeof words);
    }
    memcpy(state, words, sizeof words);
}
void hades_oracle_perm(uint64_t *state) { hades_oracle_perm_trace(state, NULL); }

/* ---- per-operation entry points (batched, for the per-op kernels' parity tests) ------ */
/* add_round_key with the constants iterator standing at `cursor` (src/strategies.rs:33-41) */
void hades_oracle_add_round_key_at(uint64_t *states, size_t n, int cursor_pos) {
    hades_oracle_init();
    for (size_t i = 0; i < n; i++) {
        const fr_t *cursor = ROUND_CONSTANTS + cursor_pos;
        add_round_key(&cursor, (fr_t *)(states + 20 * i));
    }
}
void hades_oracle_add_round_key(uint64_t *states, size_t n, int round) {
    hades_oracle_add_round_key_at(states, n, WIDTH * round);
}
void hades_oracle_quintic_s_box(uint64_t *scalars, size_t n) {
    for (size_t i = 0; i < n; i++) {
        fr_t *v = (fr_t *)(scalars + 4 * i);
        *v = quintic_s_box(*v);
    }
}
void hades_oracle_mul_matrix(uint64_t *states, size_t n) {
    hades_oracle_init();
    for (size_t i = 0; i < n; i++) mul_matrix((fr_t *)(states + 20 * i));
}

/* ---- batch driver ------------------------------------------------------------------- */
typedef struct { uint64_t *states; size_t begin, end; } span_t;
static void *perm_span(void *arg) {
    span_t *s = (span_t *)arg;
    for (size_t i = s->begin; i < s->end; i++) hades_oracle_perm(s->states + 20 * i);
    return NULL;
}
/* AoS batch: permutation i occupies states[20*i .. 20*i+20).  Static range split. */
void hades_oracle_perm_batch(uint64_t *states, size_t n, int n_threads) {
    hades_oracle_init();
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if ((size_t)n_threads > n) n_threads = n ? (int)n : 1;
    pthread_t th[256];
    span_t sp[256];
    for (int t = 0; t < n_threads; t++) {
        sp[t].states = states;
        sp[t].begin = n * t / n_threads;
        sp[t].end = n * (t + 1) / n_threads;
        if (t) pthread_create(&th[t], NULL, perm_span, &sp[t]);
    }
    perm_span(&sp[0]);
    for (int t = 1; t < n_threads; t++) pthread_join(th[t], NULL);
}

/* ---- byte format (to_bytes / from_bytes, src/round_constants.rs:61-62) --------------- */
/* 32 LE bytes of the canonical integer -> Montgomery limbs; returns 0, or -1 if >= p */
int hades_oracle_from_bytes(const uint8_t *bytes, uint64_t *limbs) {
    uint64_t v[4];
    for (int k = 0; k < 4; k++) {
        v[k] = 0;
        for (int b = 7; b >= 0; b--) v[k] = (v[k] << 8) | bytes[8 * k + b];
    }
    if (!fr_is_canonical(v)) return -1;
    fr_t m = fr_from_raw(v);
    memcpy(limbs, m.l, sizeof m.l);
    return 0;
}
void hades_oracle_to_bytes(const uint64_t *limbs, uint8_t *bytes) {
    fr_t a;
    memcpy(a.l, limbs, sizeof a.l);
    a = fr_to_canonical(a);
    for (int k = 0; k < 4; k++)
        for (int b = 0; b < 8; b++) bytes[8 * k + b] = (uint8_t)(a.l[k] >> (8 * b));
}

/* ---- synthetic generators ------------------------------------------ */
static inline uint64_t splitmix_limb(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* generator B: scalars first_elem .. first_elem+n_elems, 4 limbs each, top limb 62 bits */
void hades_oracle_gen_b(uint64_t *out, uint64_t first_elem, size_t n_elems, uint64_t seed) {
    for (size_t e = 0; e < n_elems; e++) {
        for (int k = 0; k < 4; k++) out[4 * e + k] = splitmix_limb(seed, 4 * (first_elem + e) + k);
        out[4 * e + 3] &= 0x3fffffffffffffffULL;
    }
}
/* generator A: scalar e has VALUE (first_elem + e), stored in Montgomery form */
void hades_oracle_gen_a(uint64_t *out, uint64_t first_elem, size_t n_elems) {
    for (size_t e = 0; e < n_elems; e++) {
        uint64_t v[4] = {first_elem + e, 0, 0, 0};
        fr_t m = fr_from_raw(v);
        memcpy(out + 4 * e, m.l, sizeof m.l);
    }
}

/* ---- Merkle level: parent = perm([tag, c_0 .. c_{arity-1}, 0 ..])[out_idx], arity 1..4 (caller shape of
 * dusk-poseidon, README.md:9; tag / out_idx / arity are PARAMETERS, unpinned by the reference) ---------- */
typedef struct { const uint64_t *children; uint64_t *parents; const uint64_t *tag; int arity, out_idx; size_t begin, end; } mspan_t;
static void *merkle_span(void *arg) {
    mspan_t *s = (mspan_t *)arg;
    for (size_t i = s->begin; i < s->end; i++) {
        uint64_t st[20];
        memset(st, 0, sizeof st);
        memcpy(st, s->tag, 32);
        memcpy(st + 4, s->children + 4 * (size_t)s->arity * i, 32 * (size_t)s->arity);
        hades_oracle_perm(st);
        memcpy(s->parents + 4 * i, st + 4 * s->out_idx, 32);
    }
    return NULL;
}
void hades_oracle_merkle_level(const uint64_t *children, uint64_t *parents, size_t n_parents, int arity,
                               const uint64_t *tag_mont, int out_idx, int n_threads) {
    hades_oracle_init();
    if (arity < 1 || arity > 4) return;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if ((size_t)n_threads > n_parents) n_threads = n_parents ? (int)n_parents : 1;
    pthread_t th[256];
    mspan_t sp[256];
    for (int t = 0; t < n_threads; t++) {
        sp[t] = (mspan_t){children, parents, tag_mont, arity, out_idx, n_parents * t / n_threads,
                          n_parents * (t + 1) / n_threads};
        if (t) pthread_create(&th[t], NULL, merkle_span, &sp[t]);
    }
    merkle_span(&sp[0]);
    for (int t = 1; t < n_threads; t++) pthread_join(th[t], NULL);
}
void hades_oracle_merkle4_level(const uint64_t *children, uint64_t *parents, size_t n_parents,
                                const uint64_t *tag_mont, int out_idx, int n_threads) {
    hades_oracle_merkle_level(children, parents, n_parents, 4, tag_mont, out_idx, n_threads);
}

/* ---- sponge: state = [cap,0,0,0,0]; add 4 scalars to words 1..4, perm; pad_mode 1 appends a single 1
 * first; at least one perm.  Digest = word 1.  (Caller shape of dusk-poseidon, README.md:9, which is
 * not part of the reference tree: the convention is a set of PARAMETERS, unpinned.) ---------------- */
static void sponge_one(const uint64_t *msg, size_t msg_len, const uint64_t *cap_mont, int pad_mode, uint64_t *digest) {
    const uint64_t one_raw[4] = {1, 0, 0, 0};
    const fr_t one = fr_from_raw(one_raw);
    size_t padded = msg_len + (pad_mode == 1 ? 1 : 0);
    size_t blocks = (padded + 3) / 4;
    if (blocks == 0) blocks = 1;
    fr_t st[WIDTH];
    memset(st, 0, sizeof st);
    memcpy(st[0].l, cap_mont, 32);
    for (size_t t = 0; t < blocks; t++) {
        for (int k = 0; k < 4; k++) {
            size_t idx = 4 * t + k;
            fr_t v;
            memset(&v, 0, sizeof v);
            if (idx < msg_len) memcpy(v.l, msg + 4 * idx, 32);
            else if (idx == msg_len && pad_mode == 1) v = one;
            st[1 + k] = fr_add(st[1 + k], v);
        }
        hades_oracle_perm((uint64_t *)st);
    }
    memcpy(digest, st[1].l, 32);
}
/* fixed length: message i = msgs[i*msg_len .. (i+1)*msg_len) */
void hades_oracle_sponge(const uint64_t *msgs, size_t n_msgs, size_t msg_len, const uint64_t *cap_mont,
                         int pad_mode, uint64_t *digests) {
    hades_oracle_init();
    for (size_t i = 0; i < n_msgs; i++) sponge_one(msgs + 4 * i * msg_len, msg_len, cap_mont, pad_mode, digests + 4 * i);
}
/* variable length: message i = scalars[offsets[i] .. offsets[i] + lengths[i]) */
void hades_oracle_sponge_var(const uint64_t *scalars, const uint64_t *offsets, const uint64_t *lengths, size_t n_msgs,
                             const uint64_t *cap_mont, int pad_mode, uint64_t *digests) {
    hades_oracle_init();
    for (size_t i = 0; i < n_msgs; i++)
        sponge_one(scalars + 4 * offsets[i], (size_t)lengths[i], cap_mont, pad_mode, digests + 4 * i);
}

/* ---- field-op exports for unit tests -------------------------------------------------- */
void hades_oracle_fr_add(const uint64_t *a, const uint64_t *b, uint64_t *out) {
    fr_t x, y; memcpy(x.l, a, 32); memcpy(y.l, b, 32); x = fr_add(x, y); memcpy(out, x.l, 32);
}
void hades_oracle_fr_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) {
    fr_t x, y; memcpy(x.l, a, 32); memcpy(y.l, b, 32); x = fr_mul(x, y); memcpy(out, x.l, 32);
}
void hades_oracle_fr_square(const uint64_t *a, uint64_t *out) {
    fr_t x; memcpy(x.l, a, 32); x = fr_square(x); memcpy(out, x.l, 32);
}
void hades_oracle_fr_from_raw(const uint64_t *a, uint64_t *out) {
    fr_t x = fr_from_raw(a); memcpy(out, x.l, 32);
}
void hades_oracle_fr_to_canonical(const uint64_t *a, uint64_t *out) {
    fr_t x; memcpy(x.l, a, 32); x = fr_to_canonical(x); memcpy(out, x.l, 32);
}
/* table access for tests: Montgomery limbs of ROUND_CONSTANTS[i], MDS_MATRIX[i][j] */
void hades_oracle_round_constant(int i, uint64_t *out) { hades_oracle_init(); memcpy(out, ROUND_CONSTANTS[i].l, 32); }
void hades_oracle_mds(int i, int j, uint64_t *out) { hades_oracle_init(); memcpy(out, MDS_MATRIX[i][j].l, 32); }
