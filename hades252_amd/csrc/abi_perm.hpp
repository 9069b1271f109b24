// abi_perm.hpp -- C ABI, device-resident data: library basics, `perm`, the per-round trace and
// the gadget witness (f4), the trait's per-operation methods, bulk BlsScalar arithmetic (a13) and the wire format (f3).
// Every entry point enqueues on the caller's stream and returns; include/hades252.h holds the contracts.
#pragma once

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

int hades252_perm_trace_scaled_dev(const void *d_states, void *d_trace, size_t n_perms, void *stream) {
    if (n_perms == 0) return HADES252_OK;
    if (d_states == nullptr || d_trace == nullptr || misaligned(d_states) || misaligned(d_trace)) return HADES252_ERR_INVALID_ARG;
    if (n_perms > kMaxLaunchRecords) return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_perm_trace_scaled, dim3(blocks_for(n_perms)), dim3(kBlock), lds_for(5), (hipStream_t)stream,
                       (const uint8_t *)d_states, (uint8_t *)d_trace, n_perms);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_perm_trace_scale_table(uint64_t *mul, uint64_t *add) {
    static const uint64_t kMul[67][4] = HADES_TRACE_SCALED_MUL;
    static const uint64_t kAdd[67][5][4] = HADES_TRACE_SCALED_ADD;
    if (mul == nullptr || add == nullptr) return HADES252_ERR_INVALID_ARG;
    memcpy(mul, kMul, sizeof(kMul));
    memcpy(add, kAdd, sizeof(kAdd));
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
    if (op < FR_ADD || op > FR_REDUCE_SIGNED || (impl != 0 && impl != 1)) return HADES252_ERR_INVALID_ARG;
    if (op == FR_REDUCE_SIGNED && impl != 1) return HADES252_ERR_INVALID_ARG;          // a radix-2^29 routine
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

}  // extern "C"
