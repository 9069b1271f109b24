// wire_proto.hip -- variants of the wire-format kernels (BlsScalar::to_bytes / from_bytes on device), round 4.
//
// The shipped k_wire of round 3 ran at 5.3-5.4 TB/s (to_bytes) and 3.9-4.1 TB/s (from_bytes) on 2^26 scalars against a
// float4-copy ceiling of ~6.3 TB/s, with HBM traffic = algorithmic bytes: neither the VALU (0.45 / 0.66 ms of issue) nor
// HBM (0.68 ms) is saturated -- the two do not overlap well.  This tool measures, on the same data:
//   arithmetic   OLD = mont_mul_const / mont_mul_small + finalize (two conditional subtractions)
//                NEW = from_bytes as a linear map (mont_lin, 97 multiply-adds) and finalize1 (one subtraction)
//   pipeline     U scalars per lane per trip (all loads of a trip issued up front), with or without the NEXT trip's loads
//                issued before this trip's arithmetic (PF), plain or non-temporal accesses (NT), grid size
// Every variant's output is compared with variant 0 (64-bit additive digest).
//
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I hades252_amd/csrc -o build_tools/wire_proto tools/wire_proto.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "hades_constants.inc"
#include "hades_fast.hpp"

using namespace hades;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int kBlk = 256;
__device__ const int32_t d_rp_times_r[16] = HADES_RP_TIMES_R29;
__device__ const int32_t d_from_lin[96] = HADES_WIRE_FROM_LIN;
constexpr int32_t kRpOverR = 1 << (kLB * kNL - 256);

template <bool NT>
__device__ __forceinline__ uint4 ld16(const uint4 *p) {
    if constexpr (NT) {
        uint4 v;
        v.x = __builtin_nontemporal_load(&p->x);
        v.y = __builtin_nontemporal_load(&p->y);
        v.z = __builtin_nontemporal_load(&p->z);
        v.w = __builtin_nontemporal_load(&p->w);
        return v;
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ void st16(uint4 *p, uint4 v) {
    if constexpr (NT) {
        __builtin_nontemporal_store(v.x, &p->x);
        __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z);
        __builtin_nontemporal_store(v.w, &p->w);
    } else {
        *p = v;
    }
}

template <int MODE, bool NEWA>
__device__ __forceinline__ void convert(uint4 lo, uint4 hi, uint4 &olo, uint4 &ohi, int *bad_count) {
    Fr a;
    a.l[0] = lo.x; a.l[1] = lo.y; a.l[2] = lo.z; a.l[3] = lo.w;
    a.l[4] = hi.x; a.l[5] = hi.y; a.l[6] = hi.z; a.l[7] = hi.w;
    Fr m;
    if constexpr (MODE == 1) {
        if constexpr (NEWA)
            m = finalize1(mont_lin(to_f29(a), d_from_lin));
        else
            m = finalize(mont_mul_const(to_f29(a), d_rp_times_r));
        if (!fr_is_canonical(a)) {
#pragma unroll
            for (int k = 0; k < 8; k++) m.l[k] = 0;
            if (bad_count != nullptr) atomicAdd(bad_count, 1);
        }
    } else {
        if constexpr (NEWA)
            m = finalize1(mont_mul_small(to_f29(a), kRpOverR));
        else
            m = finalize(mont_mul_small(to_f29(a), kRpOverR));
    }
    olo = make_uint4(m.l[0], m.l[1], m.l[2], m.l[3]);
    ohi = make_uint4(m.l[4], m.l[5], m.l[6], m.l[7]);
}

// U scalars per lane per trip, lane-private 32-byte accesses; PF: the next trip's loads are issued before this trip's
// arithmetic.  Scalar index of (trip t, slot u, thread g) = (t * U + u) * stride + g: every access instruction of a wave
// covers 2 KiB contiguous.
template <int MODE, bool NEWA, int U, bool PF, bool NT>
__global__ void __launch_bounds__(kBlk) k_wire_v(const uint8_t *in, uint8_t *out, size_t n, int *bad_count) {
    const size_t stride = (size_t)gridDim.x * kBlk;
    size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x;
    uint4 lo[U], hi[U];
    auto issue = [&](size_t base, uint4 (&l)[U], uint4 (&h)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t idx = base + (size_t)u * stride;
            if (idx < n) {
                const uint4 *p = reinterpret_cast<const uint4 *>(in + idx * 32);
                l[u] = ld16<NT>(p);
                h[u] = ld16<NT>(p + 1);
            }
        }
    };
    if constexpr (PF) issue(i, lo, hi);
    for (; i < n; i += (size_t)U * stride) {
        uint4 clo[U], chi[U];
        if constexpr (PF) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                clo[u] = lo[u];
                chi[u] = hi[u];
            }
            const size_t nxt = i + (size_t)U * stride;
            if (nxt < n) issue(nxt, lo, hi);
        } else {
            issue(i, clo, chi);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t idx = i + (size_t)u * stride;
            if (idx < n) {
                uint4 olo, ohi;
                convert<MODE, NEWA>(clo[u], chi[u], olo, ohi, bad_count);
                uint4 *q = reinterpret_cast<uint4 *>(out + idx * 32);
                st16<NT>(q, olo);
                st16<NT>(q + 1, ohi);
            }
        }
    }
}

// plain float4 copy with the same access shape (the ceiling for this access pattern on this box)
__global__ void __launch_bounds__(kBlk) k_copy(const uint8_t *in, uint8_t *out, size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlk;
    for (size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += stride) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + i * 32);
        uint4 a = p[0], b = p[1];
        uint4 *q = reinterpret_cast<uint4 *>(out + i * 32);
        q[0] = a;
        q[1] = b;
    }
}
__global__ void __launch_bounds__(kBlk) k_copy16(const uint4 *in, uint4 *out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * kBlk;
    for (size_t i = (size_t)blockIdx.x * kBlk + threadIdx.x; i < n16; i += stride) out[i] = in[i];
}

__global__ void k_gen(uint64_t *out, size_t n_limbs, int canonical_mode) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_limbs; i += stride) {
        uint64_t z = 0x4861646573323532ull + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        if ((i & 3) == 3) z &= 0x3fffffffffffffffull;
        out[i] = z;
    }
}
__global__ void k_digest(const uint64_t *w, size_t n, unsigned long long *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n; i += stride) {
        uint64_t z = w[i] ^ (i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull);
        z = (z ^ (z >> 32)) * 0xD6E8FEB86659FD93ull;
        acc += z ^ (z >> 29);
    }
    atomicAdd(out, acc);
}

struct Variant {
    const char *name;
    void (*fn)(const uint8_t *, uint8_t *, size_t, int *);
    int per_thread;   // scalars per thread (sets the grid); 0 = persistent grid
    int mode;
};

static uint64_t digest(const void *d, size_t n_u64, unsigned long long *d_acc) {
    CHECK(hipMemset(d_acc, 0, 8));
    hipLaunchKernelGGL(k_digest, dim3(4096), dim3(256), 0, 0, (const uint64_t *)d, n_u64, d_acc);
    unsigned long long h = 0;
    CHECK(hipMemcpy(&h, d_acc, 8, hipMemcpyDeviceToHost));
    return h;
}

#define V(MODE, NEWA, U, PF, NT, PER) {#MODE "/" #NEWA "/U" #U "/PF" #PF "/NT" #NT "/per" #PER, k_wire_v<MODE, NEWA, U, PF, NT>, PER, MODE}

int main(int argc, char **argv) {
    const int logn = argc > 1 ? atoi(argv[1]) : 26;
    const size_t n = (size_t)1 << logn;
    uint8_t *d_in, *d_out, *d_canon;
    unsigned long long *d_acc;
    CHECK(hipMalloc(&d_in, n * 32));
    CHECK(hipMalloc(&d_out, n * 32));
    CHECK(hipMalloc(&d_canon, n * 32));
    CHECK(hipMalloc(&d_acc, 8));
    hipLaunchKernelGGL(k_gen, dim3(8192), dim3(256), 0, 0, (uint64_t *)d_in, n * 4, 0);
    CHECK(hipDeviceSynchronize());
    std::vector<Variant> vs = {
        V(0, false, 1, false, false, 4),   // the shipped round-3 kernel
        V(0, true, 1, false, false, 4),
        V(0, true, 1, false, false, 1),    // one trip per thread: no loop at all
        V(0, true, 1, false, true, 1),
        V(0, true, 2, false, false, 2),
        V(0, true, 2, false, true, 2),
        V(0, true, 4, false, false, 4),
        V(0, true, 4, false, true, 4),
        V(0, true, 2, true, false, 4),
        V(0, true, 2, true, true, 4),
        V(0, true, 2, true, false, 8),
        V(1, false, 1, false, false, 4),   // the shipped round-3 kernel
        V(1, true, 1, false, false, 4),
        V(1, true, 1, false, false, 1),
        V(1, true, 1, false, true, 1),
        V(1, true, 2, false, false, 2),
        V(1, true, 2, false, true, 2),
        V(1, true, 4, false, false, 4),
        V(1, true, 4, false, true, 4),
        V(1, true, 2, true, false, 4),
        V(1, true, 2, true, true, 4),
        V(1, true, 2, true, false, 8),
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // the copy ceiling, two shapes
    for (int shape = 0; shape < 2; shape++) {
        std::vector<float> ts;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipEventRecord(e0));
            if (shape == 0)
                hipLaunchKernelGGL(k_copy, dim3((unsigned)(n / 4 / kBlk)), dim3(kBlk), 0, 0, d_in, d_out, n);
            else
                hipLaunchKernelGGL(k_copy16, dim3((unsigned)(2 * n / 4 / kBlk)), dim3(kBlk), 0, 0, (const uint4 *)d_in, (uint4 *)d_out, 2 * n);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-34s median %7.3f ms  %7.1f GB/s\n", shape == 0 ? "copy, 2 x 16 B per lane (stride 32)" : "copy, 16 B per lane (contiguous)", ts[2],
               64.0 * n / (ts[2] * 1e-3) / 1e9);
    }
    // canonical input for from_bytes = to_bytes of the generator's limbs (variant 0)
    hipLaunchKernelGGL(vs[0].fn, dim3((unsigned)(n / 4 / kBlk)), dim3(kBlk), 0, 0, d_in, d_canon, n, (int *)nullptr);
    CHECK(hipDeviceSynchronize());
    uint64_t ref[2] = {0, 0};
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    // a copy of each input in the OTHER buffer position (d_x holds the limbs, d_y the canonical bytes), to tell data /
    // placement effects from the kernels' own: pass 1 reads them from there
    uint8_t *d_x, *d_y;
    CHECK(hipMalloc(&d_y, n * 32));
    CHECK(hipMalloc(&d_x, n * 32));
    CHECK(hipMemcpy(d_x, d_in, n * 32, hipMemcpyDeviceToDevice));
    CHECK(hipMemcpy(d_y, d_canon, n * 32, hipMemcpyDeviceToDevice));
    for (int pass = 0; pass < 2; pass++)
    for (size_t v = 0; v < vs.size(); v++) {
        const Variant &va = vs[v];
        if (pass == 1 && !(v % 11 == 2 || v % 11 == 4 || v % 11 == 5)) continue;
        const uint8_t *src = pass == 0 ? (va.mode == 0 ? d_in : d_canon) : (va.mode == 0 ? d_x : d_y);
        if (pass == 1 && v % 11 == 2) printf("-- inputs from the other buffer positions:\n");
        const unsigned grid = va.per_thread ? (unsigned)(n / va.per_thread / kBlk) : (unsigned)(cus * 8);
        int vg = 0;
        hipFuncAttributes at;
        CHECK(hipFuncGetAttributes(&at, (const void *)va.fn));
        vg = at.numRegs;
        CHECK(hipMemset(d_out, 0xEE, n * 32));
        std::vector<float> ts;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(va.fn, dim3(grid), dim3(kBlk), 0, 0, src, d_out, n, (int *)nullptr);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        const uint64_t h = digest(d_out, n * 4, d_acc);
        if (ref[va.mode] == 0) ref[va.mode] = h;
        printf("%-30s vgpr %3d grid %7u  median %7.3f ms  min %7.3f  %7.1f GB/s  %s\n", va.name, vg, grid, ts[2], ts[0],
               64.0 * n / (ts[2] * 1e-3) / 1e9, h == ref[va.mode] ? "same bits" : "DIFFERENT BITS");
    }
    // from_bytes(to_bytes(x)) == x
    hipLaunchKernelGGL(vs[12].fn, dim3((unsigned)(n / 4 / kBlk)), dim3(kBlk), 0, 0, d_canon, d_out, n, (int *)nullptr);
    CHECK(hipDeviceSynchronize());
    printf("from_bytes(to_bytes(x)) == x: %s\n", digest(d_out, n * 4, d_acc) == digest(d_in, n * 4, d_acc) ? "yes" : "NO");
    return 0;
}
