// kernels_perm.hpp -- permutation kernels other than k_perm_fast: the literal schedule, per-round trace, gadget witness, per-operation kernels, field ops, wire format, five waves per state, one state per wave
// Part of the single translation unit hades252.hip (included there after device_tables.hpp); not a stand-alone header.
#pragma once

template <int OP>
__global__ void __launch_bounds__(kBlock) k_states_literal(uint8_t *states, size_t n, int cursor) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr st[5];
    wave_load_records<5>(states, rec0, n, slab, st);
    LiteralView V{d_ark_mont, d_mds_mont};
    if constexpr (OP == OP_PERM) lit_perm(V, st);
    if constexpr (OP == OP_ARK) lit_add_round_key(V, cursor, st);
    if constexpr (OP == OP_MDS) lit_mul_matrix(V, st);
    if constexpr (OP == OP_FULL) lit_full_round(V, cursor, st);
    if constexpr (OP == OP_PARTIAL) lit_partial_round(V, cursor, st);
    wave_store_records<5>(states, rec0, n, slab, st);
}

// Per-round trace: the state after every round (what the PLONK gadget needs as witnesses,
// reference src/strategies/gadget.rs:41-133), round-major: trace[r] is a whole AoS batch.
// Literal variant (the reference's schedule; parity anchor for the fast one).
__global__ void __launch_bounds__(kBlock) k_perm_trace_literal(const uint8_t *__restrict__ states,
                                                               uint8_t *__restrict__ trace, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr st[5];
    wave_load_records<5>(states, rec0, n, slab, st);
    LiteralView V{d_ark_mont, d_mds_mont};
#pragma unroll 1
    for (int r = 0; r < 67; r++) {
        if (r < 4 || r >= 63)
            lit_full_round(V, 5 * r, st);
        else
            lit_partial_round(V, 5 * r, st);
        wave_store_records<5>(trace + (size_t)r * n * 160, rec0, n, slab, st);
    }
}

// Full gadget witness: EVERY gate output of the reference's GadgetStrategy for every state -- the 972 values a
// PLONK prover assigns per permutation (src/strategies/gadget.rs:41-133: round-0 key additions, v^2 / v^4 / v^5 of
// each S-box, and per linear layer the 3-term partial sums r1[j] and the rows r2[j] with the NEXT round's constant
// appended).  Wire-major output: wires[g] is a batch of n scalars (32 B, in-memory BlsScalar), g in gate order.
//
// TRUE-FORM schedule (round 4; hades252_amd/_derive.py::witness_schedule, limb-exact replay
// tests/test_fast_model.py::witness_model): 972 values leave per permutation, so what a value costs on its way OUT
// decides the kernel, not the 400-odd products of the permutation itself.  The throughput kernel's running scale would
// make every gate output pay a 153 multiply-add constant product (round 2-3: 481 k VALU instructions per permutation,
// issue-bound at 2.3 TB/s of stores).  Here every held value is x * Rp exactly:
//   * Montgomery products (the S-boxes) are closed in that form, so v^2, v^4, v^5 are gate outputs as they stand;
//   * the linear layer M = lam C is ONE constant product per word (U = Y lam 2^29, a linear map with the same table every
//     round) + the small-integer sums over C + the one-limb Montgomery step, which divides the 2^29 out again: rows in
//     true form, r1 (three columns) and r2 (five) alike;
//   * a gate output is finalize32 of the held value: the exact division by 32 (Rp / 2^256) with nine multiply-adds and
//     one conditional subtraction.
// Round constants enter as Rp-form addends minus p (range discipline of finalize32).  Loops over words rotate the state so
// that every piece of code exists once (I-cache).
__device__ __forceinline__ void store_wire(uint8_t *wires, size_t n, int wire, size_t rec, bool live, const Fr &v) {
    if (live) {
        uint4 *q = reinterpret_cast<uint4 *>(wires + ((size_t)wire * n + rec) * 32);
        q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
        q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    }
}

// x / 32 mod p, fully reduced, for x in (-30 p, p / 8): the step from the Rp form (value * 2^261) every true-form
// kernel holds to the in-memory BlsScalar (value * 2^256).  Exact division Montgomery-style: p == 1 (mod 32), so with
// m = (-x mod 32) + 32 in [32, 63] the sum x + m p is a multiple of 32; it lies in (2 p, 64 p) -- any x in (-32 p, p)
// would do -- and its 32nd part in (0, 2 p): ONE conditional subtraction.  What the schedule hands it: products in
// (-p - 2^253, 2^253); rows (Y - m p) / 2^29 with |Y| < 2^18.1 (p + 2^228) in (-p - 2^245, 2^245), with a round constant
// in (-p, 0] appended in (-2 p - 2^245, 2^245) -- far inside the window (tests/test_fast_model.py asserts (-30 p, p / 8)
// on every output and drives the row path to its extremes).  Nine multiply-adds (m p, limb by limb, on a 32-bit carry chain) and the 9 x 29 -> 8 x 32
// packing shifted down by five bits -- against a 153 multiply-add constant product + finalize when the held value
// carries a running scale.  Lazy limbs welcome (|limb| < 2^31 - 2^29).  tests/test_fast_model.py::finalize32_model.
__device__ __forceinline__ Fr finalize32(const F29 &x) {
    const int32_t m = ((0 - x.l[0]) & 31) + 32;
    F29 y;
    int32_t carry = 0;
#pragma unroll
    for (int k = 0; k < kNL; k++) {
        const int64_t t = (int64_t)m * P29[k] + (int64_t)(x.l[k] + carry);
        y.l[k] = (int32_t)((uint32_t)t & kMask29);
        carry = (int32_t)(t >> kLB);
        if (k == kNL - 1) y.l[k] = (int32_t)t;                   // below 2^29: the sum is below 64 p < 2^261
    }
    Fr r;
#pragma unroll
    for (int w = 0; w < 8; w++) {                                  // word w = bits [32 w + 5, 32 w + 37) of the sum
        const int bit = 32 * w + 5, k = bit / kLB, sh = bit - kLB * k;
        uint64_t acc = (uint64_t)(uint32_t)y.l[k] >> sh;
        int have = kLB - sh;
        if (k + 1 < kNL) acc |= (uint64_t)(uint32_t)y.l[k + 1] << have;
        have += kLB;
        if (have < 32 && k + 2 < kNL) acc |= (uint64_t)(uint32_t)y.l[k + 2] << have;
        r.l[w] = (uint32_t)acc;
    }
    return fr_cond_sub_p(r);
}

// (sum_{k < NCOL} c[k] U_k - m p) / 2^29, normalised: NCOL columns of one row of small_mds (lazy limbs allowed)
template <int NCOL>
__device__ __forceinline__ F29 mds_row_cols(const int32_t *crow, const F29 (&u)[5]) {
    int32_t c[NCOL];
#pragma unroll
    for (int k = 0; k < NCOL; k++) c[k] = crow[k];
    F29 y;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NCOL; k++) mac(acc, u[k].l[0], c[k]);
    const int32_t m = (int32_t)((uint32_t)acc & kMask29);
    acc >>= kLB;
#pragma unroll
    for (int i = 1; i < kNL; i++) {
#pragma unroll
        for (int k = 0; k < NCOL; k++) mac(acc, u[k].l[i], c[k]);
        mac(acc, m, NEGP29[i]);
        y.l[i - 1] = (int32_t)((uint32_t)acc & kMask29);
        acc >>= kLB;
    }
    y.l[kNL - 1] = (int32_t)acc;
    return y;
}

// mont_lin (hades_fast.hpp) on NW words of a state with ONE walk over the table: the nine multipliers of a column arrive
// once (scalar loads, a column ahead) and serve NW accumulators -- 1/NW of the scalar-load traffic of separate maps, and NW
// independent multiply-add chains for the scheduler to interleave.  Word for word the same operations in the same order
// as mont_lin, hence the same limbs.  (All five words at once need 2 x 45 registers for operands and results: the callers
// go 3 + 2.)
template <int NW>
__device__ __forceinline__ void mont_lin_words(F29 *a, const int32_t *e) {
    int32_t m0[NW], m1[NW];
    F29 r[NW];
    int64_t acc[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) acc[w] = 0;
    int32_t cur[kNL], nxt[kNL], nx2[kNL];               // the multipliers travel TWO columns ahead of their use
#pragma unroll
    for (int k = 0; k < kNL; k++) {
        cur[k] = e[k];
        nxt[k] = e[kNL + k];
    }
#pragma unroll
    for (int j = 0; j < kNL; j++) {
        if (j + 2 < kNL) {
#pragma unroll
            for (int k = 0; k < kNL; k++) nx2[k] = e[kNL * (j + 2) + k];
        }
#pragma unroll
        for (int w = 0; w < NW; w++) {
#pragma unroll
            for (int k = 0; k < kNL; k++) mac(acc[w], a[w].l[k], cur[k]);
            if (j >= 1) mac(acc[w], m0[w], NEGP29[j]);
            if (j >= 2) mac(acc[w], m1[w], NEGP29[j - 1]);
            const int32_t low = (int32_t)((uint32_t)acc[w] & kMask29);
            if (j == 0)
                m0[w] = low;
            else if (j == 1)
                m1[w] = low;
            else
                r[w].l[j - 2] = low;
            acc[w] >>= kLB;
            asm volatile("" : "+v"(acc[w]));            // ties the `pin` statements of this column to the data flow (see
        }                                               // mont_mul_small: left floating they keep every partial sum alive)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < kNL; k++) {
            cur[k] = nxt[k];
            nxt[k] = nx2[k];
        }
    }
#pragma unroll
    for (int w = 0; w < NW; w++) {
        mac(acc[w], m1[w], NEGP29[kNL - 1]);
        r[w].l[kNL - 2] = (int32_t)((uint32_t)acc[w] & kMask29);
        acc[w] >>= kLB;
        r[w].l[kNL - 1] = (int32_t)acc[w];
        a[w] = r[w];
    }
}

__global__ void __launch_bounds__(kBlock, 3) k_perm_witness(const uint8_t *__restrict__ states,
                                                            uint8_t *__restrict__ wires, size_t n) {
    const size_t rec = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = rec < n;
    F29 y[5];                                           // the state WITHOUT the coming round's constants, normalised
    {
        // 160 bytes in against 31 104 out: the lanes fetch their own records, no LDS slab.  Residency is three waves per
        // SIMD either way: the kernel holds 135 VGPRs (launch bounds (kBlock, 3) = at most 168; the codegen guard allows
        // 152), and three or five waves measured the same when an earlier form of it fitted five
        Fr in[5];
#pragma unroll
        for (int w = 0; w < 5; w++) in[w] = live ? load_word(states + rec * 160 + w * 32) : zero_word();
#pragma unroll
        for (int w = 0; w < 5; w++) y[w] = to_f29(in[w]);
#pragma unroll 1
        for (int i = 0; i < 5; i++) {                   // in-memory limbs (x 2^256) -> x Rp
            int off = 0;                                // an offset the compiler cannot see through, per iteration: the 81
            asm volatile("" : "+s"(off));               // multipliers are NOT to be hoisted out of the loop (as 64-bit values,
            y[4] = mont_lin(y[4], d_wit.in_lin + off);  // spilling SGPRs and widening every product)
            rotate_right(y);
#pragma unroll
            for (int w = 0; w < 5; w++)
#pragma unroll
                for (int k = 0; k < kNL; k++) limb_fence(y[w].l[k]);
        }
    }
    int wire = 0;
#pragma unroll 1
    for (int r = 0; r < 67; r++) {
        const int32_t *c = d_wit.c[r], *ck = d_wit.ck[r];
        const bool full = r < 4 || r >= 63;
        if (r == 0) {
#pragma unroll 1
            for (int i = 0; i < 5; i++) {               // state after the first round key: word 4 - i sits at y[4]
                F29 s = y[4];
                add_lazy(s, c + (4 - i) * kNL);
                store_wire(wires, n, wire + 4 - i, rec, live, finalize32(s));
                rotate_right(y);
            }
            wire += 5;
        }
        // S-boxes: every word (full round: word 4 - i rotates through y[4]) or word 4 alone
        const int cnt = full ? 5 : 1;
#pragma unroll 1
        for (int i = 0; i < cnt; i++) {
            const int w = 4 - i;
            const int g = wire + (full ? 3 * w : 0);        // gate order: word 0 first (a partial round has one S-box)
            F29 z = y[4];
            add_lazy(z, c + w * kNL);
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(z.l[k]);
            F29 v2 = mont_sqr(z);
            store_wire(wires, n, g, rec, live, finalize32(v2));
            // (a store is a branch on `live`; instruction selection works block by block, and a limb whose 64-bit
            // extension was computed in an earlier block is multiplied as a 64-bit value -- three instructions per
            // product instead of one.  The fences make the operands 32-bit values of THIS block.)
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(v2.l[k]);
            F29 v4 = mont_sqr(v2);
            store_wire(wires, n, g + 1, rec, live, finalize32(v4));
#pragma unroll
            for (int k = 0; k < kNL; k++) {
                limb_fence(v4.l[k]);
                limb_fence(z.l[k]);
            }
            z = mont_mul(v4, z);
            store_wire(wires, n, g + 2, rec, live, finalize32(z));
            y[4] = z;
            if (full) rotate_right(y);
#pragma unroll
            for (int w2 = 0; w2 < 5; w2++)
#pragma unroll
                for (int k = 0; k < kNL; k++) limb_fence(y[w2].l[k]);
        }
        wire += 3 * cnt;
        // the constant product of the linear layer, U_w = Y_w lam 2^29; in a partial round words 0..3 carry their round
        // constant through the map as an addend
        {
            int off = 0;                                // an offset the compiler cannot see through: the 81 multipliers are
            asm volatile("" : "+s"(off));               // NOT to be hoisted out of the round loop (as 64-bit values: SGPR
            mont_lin_words<3>(y, d_wit.k_lin + off);    // spills, every product widened)
            int off2 = 0;
            asm volatile("" : "+s"(off2));
            mont_lin_words<2>(y + 3, d_wit.k_lin + off2);
            if (!full) {
#pragma unroll
                for (int w = 0; w < 4; w++) add_lazy(y[w], ck + w * kNL);
            }
        }
        // the linear layer over U = y.  r1[j]: columns 0..2 of row j, a gate output only; then the rows themselves in
        // place (small_mds, limb-major: no second copy of the state); r2[j] = row j + the next round's constant
#pragma unroll 1
        for (int j = 0; j < 5; j++) {
#pragma unroll
            for (int w = 0; w < 3; w++)
#pragma unroll
                for (int k = 0; k < kNL; k++) limb_fence(y[w].l[k]);
            store_wire(wires, n, wire + 2 * j, rec, live, finalize32(mds_row_cols<3>(d_coop.mds[j], y)));
        }
        small_mds(y);
        const int32_t *cn = d_wit.c[r + 1];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            F29 s = y[j];
            add_lazy(s, cn + j * kNL);
            store_wire(wires, n, wire + 2 * j + 1, rec, live, finalize32(s));
        }
#pragma unroll
        for (int w = 0; w < 5; w++)
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(y[w].l[k]);
        wire += 10;
    }
}

// slab_flush (staging.hpp) for a kernel that flushes EVERY ROUND: the slab is private to the wave, so what orders the
// per-lane ds_write of slab_put against the transposed ds_read here is the wave's own program order (the LDS unit serves a
// wave's requests in order) -- not a block barrier.  Two s_barrier per round would march the four waves of a block in
// lockstep: all of them store together, none of them issues arithmetic meanwhile (measured on the scaled trace, 2^20 states:
// see profiles/r6/).  The wavefront-scope fences only keep the compiler from moving LDS accesses across the hand-over.
template <int NW>
__device__ __forceinline__ void slab_flush_wave(uint8_t *base, size_t rec0, size_t n_recs, uint8_t *slab) {
    constexpr int kLdsRecBytes = lds_rec_bytes(NW);
    const int lane = threadIdx.x & (kWave - 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const size_t rec_bytes = (size_t)NW * 32;
    const size_t total_chunks = n_recs * (size_t)(2 * NW);
    uint4 *g = reinterpret_cast<uint4 *>(base + rec0 * rec_bytes);
    const size_t chunk0 = rec0 * (size_t)(2 * NW);
#pragma unroll
    for (int k = 0; k < 2 * NW; k++) {
        int c = k * kWave + lane;
        int rec = c / (2 * NW), part = c - rec * (2 * NW);
        uint4 v = *reinterpret_cast<const uint4 *>(slab + rec * kLdsRecBytes + part * 16);
        if (chunk0 + c < total_chunks) g[c] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-round trace, shipped form: the state after every round (reference src/strategies.rs:140-157 observed round by round;
// what the gadget's rows are before the next round key), round-major: trace[r] is a whole AoS batch.  The rounds of
// k_perm_witness -- every value held as x Rp, the linear layer as one constant linear map per word + the small-integer
// rows -- with the five words of the state as the only outputs, each through finalize32, coalesced through the wave's
// staging slab.  Rounds 2-3 ran the throughput kernel's scale-tracked rounds and un-scaled five words per round with a
// per-round constant map + full reduction + (partial rounds) a field addition of the deferred constants: 188 k VALU
// instructions per permutation against 154 k here.
__global__ void __launch_bounds__(kBlock, 3) k_perm_trace_fast(const uint8_t *__restrict__ states,
                                                               uint8_t *__restrict__ trace, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    const size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    F29 y[5];
    {
        Fr in[5];
        wave_load_records<5>(states, rec0, n, slab, in);
#pragma unroll
        for (int w = 0; w < 5; w++) y[w] = to_f29(in[w]);
#pragma unroll 1
        for (int i = 0; i < 5; i++) {                   // in-memory limbs (x 2^256) -> x Rp
            int off = 0;                                // (opaque offset: see k_perm_witness)
            asm volatile("" : "+s"(off));
            y[4] = mont_lin(y[4], d_wit.in_lin + off);
            rotate_right(y);
#pragma unroll
            for (int w = 0; w < 5; w++)
#pragma unroll
                for (int k = 0; k < kNL; k++) limb_fence(y[w].l[k]);
        }
    }
#pragma unroll 1
    for (int r = 0; r < 67; r++) {
        const int32_t *c = d_wit.c[r], *ck = d_wit.ck[r];
        const bool full = r < 4 || r >= 63;
        // touch the lines of the NEXT round's addends now (26 KB of them do not stay in the scalar cache while the waves of
        // a CU sit in different rounds); consumed after the S-boxes, so the misses travel behind ~2 000 cycles of products
        const int32_t *cq = d_wit.c[r + 1], *ckq = d_wit.ck[r < 66 ? r + 1 : 66];
        const int32_t warm = cq[0] | cq[16] | cq[32] | ckq[0] | ckq[16] | ckq[32];
        const int cnt = full ? 5 : 1;
#pragma unroll 1
        for (int i = 0; i < cnt; i++) {                 // S-boxes: word 4 - i rotates through y[4] in a full round
            F29 z = y[4];
            add_lazy(z, c + (4 - i) * kNL);
            y[4] = sbox29(z);
            if (full) rotate_right(y);
#pragma unroll
            for (int w = 0; w < 5; w++)
#pragma unroll
                for (int k = 0; k < kNL; k++) limb_fence(y[w].l[k]);
        }
        asm volatile("" ::"s"(warm));
        {                                               // U_w = Y_w lam 2^29 (+ the round constants seen through the map)
            int off = 0;
            asm volatile("" : "+s"(off));
            mont_lin_words<3>(y, d_wit.k_lin + off);
            int off2 = 0;
            asm volatile("" : "+s"(off2));
            mont_lin_words<2>(y + 3, d_wit.k_lin + off2);
            if (!full) {
#pragma unroll
                for (int w = 0; w < 4; w++) add_lazy(y[w], ck + w * kNL);
            }
        }
        small_mds(y);
#pragma unroll
        for (int w = 0; w < 5; w++) slab_put<5>(slab, w, finalize32(y[w]));
        slab_flush_wave<5>(trace + (size_t)r * n * 160, rec0, n, slab);
#pragma unroll
        for (int w = 0; w < 5; w++)
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(y[w].l[k]);
    }
}

// x mod p, fully reduced, for the values the throughput kernel's linear layer leaves: NORMALISED limbs (0..7 in [0, 2^29),
// the signed top limb carries the sign) and x in (-p - 2^250, 2^250].  Almost all of that window is [-p, 0) -- the layer's
// one-limb Montgomery step subtracts m p with m uniform in [0, 2^29) from a sum of magnitude below 2^-8 p 2^29 -- where the
// answer is x + p and nothing else.  So: pack the limbs straight into the 256-bit two's-complement image of x (no carry
// pass: they are normalised), add p once, and take the two rare sides -- x >= 0: x itself; x < -p: one more p -- on a
// wave-uniform branch that a wave skips unless one of its lanes needs it (a few per thousand words).  36 + 4 instructions on
// the common path against the 87 of `finalize` (carry pass with 2 p, packing, two conditional subtractions).
// tests/test_fast_model.py::finalize_window_model replays it on the window's edges.
__device__ __forceinline__ Fr finalize_window(const F29 &x) {
    const Fr t = from_f29(x);                           // x mod 2^256
    Fr u;
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {                       // (add-with-carry builtin: v_add_co_u32 + 7 v_addc_co_u32, see fr32.hpp)
        unsigned co;
        u.l[i] = __builtin_addc(t.l[i], FR_P[i], c, &co);
        c = co;
    }
    const bool pos = (int32_t)t.l[7] >= 0;             // x >= 0 (below 2^250 < p): canonical as it stands
    const bool low = (int32_t)u.l[7] < 0;              // x + p < 0: one more p
    if (__any(pos || low)) {
        Fr v;
        c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            unsigned co;
            v.l[i] = __builtin_addc(u.l[i], FR_P[i], c, &co);
            c = co;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) u.l[i] = pos ? t.l[i] : (low ? v.l[i] : u.l[i]);
    }
    return u;
}

// Per-round trace, SCALED form (opt-in: hades252_perm_trace_scaled_dev).  The true-form kernel above pays 485 multiply-adds
// per round to keep every word at scale Rp so that a word can leave through finalize32; here the rounds are the throughput
// kernel's own (fast_round: the state carries the running scale s_r and, in the partial rounds, lacks the constants still
// deferred), and a word leaves through `finalize_window` ALONE -- pack, add p, a rarely taken fix-up: no product.
// What is stored is a fully reduced field element all the same: scaled[r][w] = s_after(r) (true[r][w] - d_r[w]) mod p, and
// the consumer recovers true[r][w] = scaled[r][w] * MUL[r] + ADD[r][w] (BlsScalar operations, 67 multipliers + 67 x 5
// addends from hades252_perm_trace_scale_table) lazily -- fused into whatever reads the trace next -- or never, where a
// relation is homogeneous.  The window (-p - 2^250, 2^250] holds for every word the linear layer produces
// (tests/test_fast_model.py::test_scaled_trace_model_matches_spec_oracle asserts it round by round).
__global__ void __launch_bounds__(kBlock, 4) k_perm_trace_scaled(const uint8_t *__restrict__ states,
                                                                 uint8_t *__restrict__ trace, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    const size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    F29 st[5];
    {
        Fr in[5];
        wave_load_records<5>(states, rec0, n, slab, in);
#pragma unroll
        for (int w = 0; w < 5; w++) st[w] = to_f29(in[w]);
    }
#pragma unroll 1
    for (int r = 0; r < 67; r++) {
        fast_round(d_fast.round[r], d_fast.lin[r], r < 4 || r >= 63, st);
#pragma unroll
        for (int w = 0; w < 5; w++) slab_put<5>(slab, w, finalize_window(st[w]));
        slab_flush_wave<5>(trace + (size_t)r * n * 160, rec0, n, slab);
    }
}

// Generic batched BlsScalar operations (reference call sites src/strategies/scalar.rs:28,33,44;
// src/round_constants.rs:41): out[i] = a[i] (op) b[i] on Montgomery limbs, fully reduced.
// IMPL 0: the saturated 8 x u32 CIOS arithmetic of fr32.hpp (what the literal kernels use);
// IMPL 1: the radix-2^29 signed-limb arithmetic of the shipped kernel (to_f29, mont_fips, finalize).
// These exist so that tests can drive BOTH device arithmetics through the computations that produced
// the reference's constant blobs (tests/test_gpu_blob_kat.py), and as a13's batched surface.
enum FrOp { FR_ADD = 0, FR_MUL = 1, FR_SQUARE = 2, FR_FROM_RAW = 3, FR_REDUCE_SIGNED = 4 };
template <int IMPL>
__global__ void __launch_bounds__(kBlock) k_fr_op(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n, int op) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<1>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr x[1], y[1];
    wave_load_records<1>(a, rec0, n, slab, x);
    if (op == FR_ADD || op == FR_MUL) {
        wave_load_records<1>(b, rec0, n, slab, y);
    } else if (op == FR_SQUARE) {
        y[0] = x[0];
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) y[0].l[i] = d_r2[i];
    }
    Fr r[1];
    if constexpr (IMPL == 0) {
        r[0] = (op == FR_ADD) ? fr_add(x[0], y[0]) : fr_mul(x[0], y[0]);
    } else {
        F29 xa = to_f29(x[0]), yb = to_f29(y[0]);
        if (op == FR_REDUCE_SIGNED) {
            // a = two's-complement image of a signed x in (-p - 2^250, 2^250]: the exit of the scaled trace kernel
            // (finalize_window) on values a test chooses -- both of its rare sides included
            xa.l[kNL - 1] = (int32_t)x[0].l[7] >> 8;               // bits 232 .. 255, sign-extended
            r[0] = finalize_window(xa);
        } else if (op == FR_ADD) {
            add_lazy(xa, yb.l);                                    // limbs < 2^30
            r[0] = finalize(mont_mul_const(xa, d_rp_mod_p));       // (a + b) * Rp / Rp
        } else {
            F29 t = (op == FR_SQUARE) ? mont_sqr(xa) : mont_mul(xa, yb);   // a b / Rp
            r[0] = finalize(mont_mul_const(t, d_rp2_over_r));      // * (Rp^2 / 2^256) / Rp = a b / 2^256
        }
    }
    wave_store_records<1>(out, rec0, n, slab, r);
}

// The trait's per-operation methods on the radix-2^29 path: same field elements as the literal forms above (kept
// for add_round_key, which is five additions), a sixth to a tenth of the instructions -- mul_matrix is the
// small-integer layer + one un-scaling product per word instead of 25 full products.  Round keys are added in the
// memory format first (any cursor over all 960 constants), so the result of every method is the unique reduced
// BlsScalar, bit-identical to the literal kernels and the oracle.
template <int OP>
__global__ void __launch_bounds__(kBlock, 3) k_states_fast(uint8_t *states, size_t n, int cursor) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<5>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr in[5];
    wave_load_records<5>(states, rec0, n, slab, in);
    if constexpr (OP != OP_MDS) {
        LiteralView V{d_ark_mont, d_mds_mont};
        lit_add_round_key(V, cursor, in);
    }
    F29 st[5];
#pragma unroll
    for (int w = 0; w < 5; w++) st[w] = to_f29(in[w]);
    if constexpr (OP == OP_FULL) {
#pragma unroll 1
        for (int i = 0; i < 5; i++) {                 // one S-box body, the state rotates through it
            st[4] = sbox29(st[4]);
            rotate_right(st);
#pragma unroll
            for (int k = 0; k < kNL; k++) limb_fence(st[4].l[k]);
        }
    }
    if constexpr (OP == OP_PARTIAL) st[4] = mont_lin(sbox29(st[4]), d_op_k_lin);
    small_mds(st);
    const int32_t *u = OP == OP_FULL ? d_op_w_full_lin : d_op_w_lin;
    Fr out[5];
#pragma unroll
    for (int w = 0; w < 5; w++) out[w] = finalize(mont_lin(st[w], u));
    wave_store_records<5>(states, rec0, n, slab, out);
}

__global__ void __launch_bounds__(kBlock) k_sbox(uint8_t *scalars, size_t n) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *slab = wave_slab<1>(lds);
    size_t rec0 = (size_t)blockIdx.x * kBlock + (threadIdx.x / kWave) * kWave;
    Fr st[1];
    wave_load_records<1>(scalars, rec0, n, slab, st);
    st[0] = finalize(mont_lin(sbox29(to_f29(st[0])), d_op_k_lin));
    wave_store_records<1>(scalars, rec0, n, slab, st);
}

// canonical bytes <-> Montgomery limbs (BlsScalar::from_bytes / to_bytes): 64 B of HBM traffic and ONE constant
// multiplication per scalar, on the radix-2^29 path:
//   to_bytes    x / 2^256 = mont_mul_small(x, Rp / 2^256 = 32): the factor is ONE limb, 81 multiply-adds; the result lies
//               in (-p, 0] for a reduced input, so + p and ONE conditional subtraction finish it (finalize1)
//   from_bytes  a * 2^256 as a LINEAR MAP over the limbs of a (mont_lin: 97 multiply-adds instead of the 153 of a
//               Montgomery product; result in (-p, 2^-25 p): finalize1 again); inputs >= p are rejected
// Memory shape (tools/wire_proto.hip, tools/copy_proto.hip; profiles/r4/wire_proto.txt): no LDS, every lane reads and
// writes its own 32 bytes with two 16-byte accesses (a wave's two instructions together cover 2 KiB contiguous); ONE
// trip per thread and no loop -- a plain copy on this part runs 6.19 TB/s with one trip per thread, 5.9 with four,
// 5.5 with a persistent grid: short-lived waves pace HBM best; kWireU scalars per thread, all loads issued before
// the first conversion (from_bytes: 2, its longer arithmetic wants more bytes in flight per wave; to_bytes: 1).
// Non-temporal accesses were slower in every shape.  `out` may be `in` (lane-private in-place update).
constexpr int32_t kRpOverR = 1 << (kLB * kNL - 256);        // 2^261 / 2^256
template <int MODE>
constexpr int kWireU = MODE == 1 ? 2 : 1;
// Launch bounds: 8 (to_bytes) / 6 (from_bytes) waves per SIMD -- without them hipcc schedules the straight-line body for
// instruction-level parallelism, takes 197 VGPRs and leaves two waves per SIMD (measured: 4.0 TB/s instead of 5.9).
template <int MODE>   // 0 = to_bytes (x / 2^256), 1 = from_bytes (a * 2^256, inputs >= p rejected)
__global__ void __launch_bounds__(kBlock, MODE == 1 ? 6 : 8) k_wire(const uint8_t *in, uint8_t *out, size_t n, int *bad_count) {
    constexpr int U = kWireU<MODE>;
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    uint4 lo[U], hi[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t idx = i + (size_t)u * stride;
        if (idx < n) {
            const uint4 *p = reinterpret_cast<const uint4 *>(in + idx * 32);
            lo[u] = p[0];
            hi[u] = p[1];
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t idx = i + (size_t)u * stride;
        if (idx < n) {
            Fr a;
            a.l[0] = lo[u].x; a.l[1] = lo[u].y; a.l[2] = lo[u].z; a.l[3] = lo[u].w;
            a.l[4] = hi[u].x; a.l[5] = hi[u].y; a.l[6] = hi[u].z; a.l[7] = hi[u].w;
            Fr m = finalize1(MODE == 1 ? mont_lin(to_f29(a), d_wire_from_lin) : mont_mul_small(to_f29(a), kRpOverR));
            if (MODE == 1 && !fr_is_canonical(a)) {
#pragma unroll
                for (int k = 0; k < 8; k++) m.l[k] = 0;
                if (bad_count != nullptr) atomicAdd(bad_count, 1);
            }
            uint4 *q = reinterpret_cast<uint4 *>(out + idx * 32);
            q[0] = make_uint4(m.l[0], m.l[1], m.l[2], m.l[3]);
            q[1] = make_uint4(m.l[4], m.l[5], m.l[6], m.l[7]);
        }
    }
}

// ---- low-latency kernels: five waves per state (hades_coop.hpp) --------------------------------------
// In-place permutation of up to 64 states per 320-thread block.
__global__ void __launch_bounds__(kCoopThreads) k_perm_coop(uint8_t *states, size_t n) {
    __shared__ CoopLds L;
    const int wv = coop_word_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));   // the word this wave owns
    const int lane = threadIdx.x & (kWave - 1);
    const size_t rec0 = (size_t)blockIdx.x * kCoopStates;
    const size_t total = n * 10, chunk0 = rec0 * 10;
    uint4 *g = reinterpret_cast<uint4 *>(states + rec0 * 160);
    coop_load_constants(&d_coop, L);
#pragma unroll
    for (int c = threadIdx.x; c < kCoopStates * 10; c += kCoopThreads) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (chunk0 + c < total) v = g[c];
        const int rec = c / 10, part = c - rec * 10;
        *reinterpret_cast<uint4 *>(L.stage + rec * 176 + part * 16) = v;
    }
    __syncthreads();
    Fr w;
    {
        const uint4 *p = reinterpret_cast<const uint4 *>(L.stage + lane * 176 + wv * 32);
        uint4 lo = p[0], hi = p[1];
        w.l[0] = lo.x; w.l[1] = lo.y; w.l[2] = lo.z; w.l[3] = lo.w;
        w.l[4] = hi.x; w.l[5] = hi.y; w.l[6] = hi.z; w.l[7] = hi.w;
    }
    const F29 fin = coop_rounds(&d_coop, L, wv, to_f29(w));     // 67 barriers: everyone has read `stage` by now
    const Fr o = coop_finish(&d_coop, fin);
    {
        uint4 *p = reinterpret_cast<uint4 *>(L.stage + lane * 176 + wv * 32);
        p[0] = make_uint4(o.l[0], o.l[1], o.l[2], o.l[3]);
        p[1] = make_uint4(o.l[4], o.l[5], o.l[6], o.l[7]);
    }
    __syncthreads();
#pragma unroll
    for (int c = threadIdx.x; c < kCoopStates * 10; c += kCoopThreads) {
        const int rec = c / 10, part = c - rec * 10;
        uint4 v = *reinterpret_cast<const uint4 *>(L.stage + rec * 176 + part * 16);
        if (chunk0 + c < total) g[c] = v;
    }
}

// ---- lowest-latency kernels: one state per WAVE, every field element spread over a 16-lane row (hades_lanes.hpp) ---
// Two forms, four waves per block (one per SIMD) either way:
//   HELPED   three states per block + a helper wave that owns word 3 of all three during the full rounds (its S-box then
//            runs beside the main waves' instead of doubling their instruction stream): 50 us -- up to 768 states, one
//            block per CU;
//   plain    four states per block, every wave does everything itself: 54 us -- for 769 .. 1 024 states, where the helped
//            form would put a second block on some CUs.
constexpr int kLanesWaves = 4;
struct LanesAlways {
    __device__ __forceinline__ bool operator()(size_t) const { return true; }
};
// `wanted(rec)` (wave-uniform) lets a kernel drop records it does not need; such a wave idles like one past the end
template <bool HELPED, class Wanted = LanesAlways>
__device__ __forceinline__ bool lanes_role(LanesLds *L, size_t n, size_t &rec, Wanted wanted = Wanted()) {
    const int wave = threadIdx.x >> 6;                                                   // false: this wave is done
    if constexpr (HELPED) {
        if (wave == kLanesWaves - 1) {
            lanes_helper<kLanesWaves - 1>(&d_lanes, *reinterpret_cast<LanesLds(*)[kLanesWaves - 1]>(L));
            return false;
        }
        rec = (size_t)blockIdx.x * (kLanesWaves - 1) + wave;
        if (rec >= n || !wanted(rec)) {
            lanes_idle();
            return false;
        }
        return true;
    } else {
        rec = (size_t)blockIdx.x * kLanesWaves + wave;
        return rec < n && wanted(rec);                               // no block-wide barrier anywhere: idle waves leave
    }
}

// ---- four states per wave, one per 16-lane row (hades_lanes.hpp::rows_perm): batches of 1 025 .. 4 096 states --------
// One wave per SIMD up to 4 096 states (256 blocks x 4 waves x 4 states): ~71 us, between one state per wave (which
// would queue several waves per SIMD here) and five waves per state (104 us).
constexpr int kRowsWaves = 4, kRowsPerWave = 4;
__device__ __forceinline__ bool rows_role(size_t n, size_t &rec, int &word) {          // false: nothing for this lane
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    const int s = lane / 5;
    word = lane - 5 * s;
    rec = ((size_t)blockIdx.x * kRowsWaves + wave) * kRowsPerWave + s;
    return lane < 5 * kRowsPerWave && rec < n;
}

__global__ void __launch_bounds__(kRowsWaves *kWave) k_perm_rows(uint8_t *states, size_t n) {
    __shared__ RowsLds L[kRowsWaves];
    const int wave = threadIdx.x >> 6;
    if (((size_t)blockIdx.x * kRowsWaves + wave) * kRowsPerWave >= n) return;             // (no block-wide barrier anywhere)
    size_t rec;
    int word;
    const bool mine = rows_role(n, rec, word);
    uint8_t *p = states + (mine ? rec : 0) * 160 + word * 32;
    const Fr in = mine ? load_word(p) : zero_word();
    const Fr out = rows_perm(&d_rows, d_rows_klin, L[wave], in);
    if (mine) store_word(p, out);
}

// In-place permutation, one state per wave.
template <bool HELPED>
__global__ void __launch_bounds__(kLanesWaves *kWave) k_perm_lanes(uint8_t *states, size_t n) {
    __shared__ LanesLds L[kLanesWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    size_t rec;
    if (!lanes_role<HELPED>(L, n, rec)) return;
    uint8_t *mine = states + rec * 160 + (lane < 5 ? lane : 0) * 32;
    const Fr in = lane < 5 ? load_word(mine) : zero_word();
    const Fr out = lanes_perm<HELPED>(&d_lanes, L[wave], in);
    if (lane < 5) store_word(mine, out);
}
