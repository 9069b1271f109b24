// abi_util.hpp -- C ABI: the synthetic generators and the digest the benchmark and the tests use, and plain device memory /
// stream handles for callers that link nothing but this library.
#pragma once

extern "C" {

// ---- synthetic / digest ------------------------------------------------------------------------
int hades252_gen_b_dev(void *d_scalars, uint64_t first_elem, size_t n_elems, uint64_t seed, void *stream) {
    if (n_elems == 0) return HADES252_OK;
    if (d_scalars == nullptr) return HADES252_ERR_INVALID_ARG;
    size_t n_limbs = n_elems * 4;
    size_t want = (n_limbs + kBlock - 1) / kBlock;
    unsigned grid = (unsigned)(want < 65536 ? want : 65536);
    hipLaunchKernelGGL(k_gen_b, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (uint64_t *)d_scalars, first_elem,
                       n_limbs, seed);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_gen_a_dev(void *d_scalars, uint64_t first_elem, size_t n_elems, void *stream) {
    if (n_elems == 0) return HADES252_OK;
    if (d_scalars == nullptr || n_elems > kMaxLaunchRecords || misaligned(d_scalars)) return HADES252_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_gen_a, dim3(blocks_for(n_elems)), dim3(kBlock), lds_for(1), (hipStream_t)stream,
                       (uint8_t *)d_scalars, first_elem, n_elems);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

int hades252_digest_dev(const void *d_words, uint64_t first_index, size_t n_u64, void *d_out4, void *stream) {
    if (d_out4 == nullptr || (d_words == nullptr && n_u64 > 0)) return HADES252_ERR_INVALID_ARG;
    HIP_TRY(hipMemsetAsync(d_out4, 0, 32, (hipStream_t)stream));
    if (n_u64 == 0) return HADES252_OK;
    size_t want = (n_u64 + kBlock - 1) / kBlock;
    unsigned grid = (unsigned)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(k_digest, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (const uint64_t *)d_words,
                       first_index, n_u64, (unsigned long long *)d_out4);
    HIP_TRY(hipGetLastError());
    return HADES252_OK;
}

// ---- device memory for callers without HIP bindings ------------------------------------------------
int hades252_dev_alloc(void **d_ptr, size_t bytes) {
    if (d_ptr == nullptr || bytes == 0) return HADES252_ERR_INVALID_ARG;
    *d_ptr = nullptr;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    HIP_TRY(hipMalloc(d_ptr, bytes));
    return HADES252_OK;
}

int hades252_dev_free(void *d_ptr) {
    if (d_ptr == nullptr) return HADES252_OK;
    HIP_TRY(hipFree(d_ptr));
    return HADES252_OK;
}

int hades252_dev_upload(void *d_dst, const void *h_src, size_t bytes, void *stream) {
    if (bytes == 0) return HADES252_OK;
    if (d_dst == nullptr || h_src == nullptr) return HADES252_ERR_INVALID_ARG;
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return HADES252_OK;
}

int hades252_dev_download(void *h_dst, const void *d_src, size_t bytes, void *stream) {
    if (bytes == 0) return HADES252_OK;
    if (h_dst == nullptr || d_src == nullptr) return HADES252_ERR_INVALID_ARG;
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return HADES252_OK;
}

int hades252_stream_create(void **stream) {
    if (stream == nullptr) return HADES252_ERR_INVALID_ARG;
    *stream = nullptr;
    int rc = check_device();
    if (rc != HADES252_OK) return rc;
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return HADES252_OK;
}

int hades252_stream_destroy(void *stream) {
    if (stream == nullptr) return HADES252_OK;
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return HADES252_OK;
}

int hades252_stream_sync(void *stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return HADES252_OK;
}

}  // extern "C"
